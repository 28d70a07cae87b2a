// csgn_circuit.hip -- csgn_circuit_*: a circuit of adds, multiplies, permutations, encrypts, compactions and decrypts over
// batches, DESCRIBED by the caller, COMPILED by csgn_circuit_build and replayed as one hipGraph ("keep
// intermediates device-resident across a depth-d add/mul circuit; fuse mul->decrypt / add->decrypt").
// Reference callers: tests/basic_operations.cpp:34-40; operations src/Ciphertext.cpp:204-247, src/SecretKey.cpp:104-147.
//
// Two modes.
//   TAPE (default, csgn_circuit_optimize never called): every value the caller described is written to a region of its
//     own in the circuit's block and stays addressable (csgn_circuit_value) after every run -- one kernel per node.
//   COMPILED (csgn_circuit_optimize): only inputs, values named by csgn_circuit_output and decrypt bits survive a run;
//     everything else is the compiler's to arrange.  Passes, each behind a flag:
//       FUSE_DECRYPT  Dec is a ring homomorphism (Dec(a*b) = Dec(a) & Dec(b), Dec(a+b) = Dec(a) ^ Dec(b)): a product or
//                     sum whose only consumer is a decrypt is never computed -- its operands are decrypted and the bits
//                     combined.  One level by default; PUSHDOWN repeats it as far as single-consumer values reach.
//       dead code     a node whose value nothing reads any more (its consumer was fused away, or never existed) is dropped.
//       PLACE         add is concatenation (src/Ciphertext.cpp:107-122): when a product's or a sum's only consumer is an
//                     add, the SUM's region is allocated and the producer writes straight into its slice of every
//                     element (a per-element output pitch in k_mul_flat / k_mul_tiled / k_add_flat) -- that operand's half
//                     of the add disappears; operands that cannot be placed (inputs, shared values) are copied into
//                     their slice by a strided one-operand copy.
//       HOIST         copies whose source is a circuit INPUT (an input added to a product, a sum of two inputs) do not depend
//                     on anything the graph computes: all of them go into ONE strided-copy launch in front of the first
//                     node (k_copy_list) instead of one small launch each -- config 5 is 16 of its 26 kernels lighter.
//                     (The sums they write into are live from the start of the run: the block grows a little.)
//       REUSE         liveness: the graph is a chain of kernel nodes, so a region is free once the last node that reads it
//                     has been emitted; regions are dealt out of a free list and the block is the PEAK live set, not the
//                     sum of all values (csgn_circuit_block_bytes).
// The words of every retained value and every bit are those of the tape (and of the one-by-one C-ABI calls, and of the
// oracle): the passes move and skip work, they never change a result.
#include "csgn_capi_util.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace csgn::capi;

struct csgn_circuit {
    struct Value {
        uint64_t terms = 0;          // uniform: terms per element; ragged: 0
        // ragged values (static shapes: the per-element term counts are fixed when the circuit is
        // described, so every size downstream is known on the host and the graph needs no plan step)
        std::vector<uint64_t> per;   // batch entries, empty for a uniform value
        uint64_t total = 0;          // terms over the whole batch
        uint64_t max_terms = 0;
        // a value behind a compaction: `per`, `total`, `max_terms` are static UPPER BOUNDS (they size the
        // buffers and the launches), the real CSR offsets are written by the device in every run
        bool dynamic = false;
        bool needs_csr = false;      // someone reads CSR offsets of this value (ragged values always; uniform ones that meet a ragged one)
        bool is_input = false;
        bool output = false;         // csgn_circuit_output: stays materialised and addressable in compiled mode
        int producer = -1;           // op index, -1 for an input
        // ---- filled by csgn_circuit_build
        bool addressable = false;    // csgn_circuit_value answers
        bool has_region = false;
        size_t offset = 0;           // bytes into the block: element 0 of the value (inside its sum's region when placed)
        uint64_t pitch = 0;          // uniform: words from one element to the next where the value is WRITTEN (terms*dL when dense)
        size_t csr_offset = 0;       // bytes into the block of the batch+1 CSR term offsets
        int parent = -1;             // placed: the sum value whose region holds this one
        uint64_t parent_off = 0;     // ... at this word offset inside every element of the parent
    };
    struct Op {
        int kind;             // 0 add, 1 mul, 2 decrypt, 3 permute, 4 encrypt (keyed generator), 5 fused Enc*Enc (+Dec), 6 compact
        uint32_t a, b, out;
        const void *mask;     // decrypt / encrypt: key mask (u64 words); permute: permutation (u32 entries)
        size_t scratch, bits; // byte offsets, assigned by build (decrypt / compact / dynamic multiply)
        int bits_id;          // decrypt, fused Enc*Enc: index into bits_offsets (-1: none)
        // encrypt only
        const uint8_t *plain;
        const uint64_t *key;
        uint64_t d, first;
        csgn_rng rng;
        // fused Enc*Enc only
        const uint8_t *plain_b;
        csgn_rng rng_b;
        bool want_bits;
        // ---- filled by csgn_circuit_build
        bool elided;          // fused into a decrypt, or dead: no kernel
        bool placed_a, placed_b;   // add: that operand was written into the sum's slice by its producer
        bool hoist_a, hoist_b;     // add: that operand is an input, copied into the slice by the prologue launch
        int expr;             // decrypt: root of its expression (index into exprs), -1 = plain decrypt of a
    };
    // a decrypt's expression: leaves are materialised values, inner nodes AND / XOR of bit vectors
    struct Expr {
        int kind;             // -1 leaf, 0 xor (sum), 1 and (product)
        uint32_t value;       // leaf
        int l, r;
        size_t bits;          // byte offset of this node's batch bytes (the root's = the decrypt's result buffer)
        size_t scratch;       // leaf: decrypt scratch
    };
    uint64_t n_bits = 0, batch = 0;
    std::vector<Value> values;
    std::vector<Op> ops;
    std::vector<Expr> exprs;
    std::vector<size_t> bits_offsets;
    uint32_t flags = 0;       // CSGN_CIRCUIT_* passes; 0 = tape
    size_t bytes = 0;         // block size
    size_t epoch_offset = 0;  // u64 run counter in the block, bumped by the graph's first node
    size_t hoist_offset = 0;  // the prologue's copy table in the block: [CopyEntry x n][u32 first workgroup x (n + 1)]
    uint32_t n_hoist = 0;
    bool has_encrypt = false;
    uint64_t runs = 0;
    uint64_t stats[8] = {0};
    void *block = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

namespace {

typedef csgn_circuit::Value Value;
typedef csgn_circuit::Op Op;
typedef csgn_circuit::Expr Expr;

uint64_t terms_of(const Value &v, uint64_t i) { return v.per.empty() ? v.terms : v.per[i]; }

Value make_uniform(csgn_circuit *c, uint64_t terms, int producer)
{
    Value v;
    v.terms = terms;
    v.total = c->batch * terms;
    v.max_terms = terms;
    v.producer = producer;
    v.is_input = producer < 0;
    return v;
}

Op make_op(int kind)
{
    Op op;
    memset(&op, 0, sizeof(op));
    op.kind = kind;
    op.bits_id = -1;
    op.expr = -1;
    return op;
}

// ---- the block's allocator: persistent items from the bottom, transient ones out of a free list above them
struct Arena {
    static constexpr size_t kAlign = 256;
    struct Free {
        size_t at, n;
    };
    std::vector<Free> free;      // sorted by address, coalesced
    size_t top = 0;
    bool reuse = false;
    static size_t up(size_t n) { return (n + kAlign - 1) & ~(kAlign - 1); }
    size_t take(size_t n)
    {
        n = up(n ? n : 1);
        if (reuse) {
            int best = -1;
            for (size_t i = 0; i < free.size(); ++i)
                if (free[i].n >= n && (best < 0 || free[i].n < free[(size_t)best].n))
                    best = (int)i;
            if (best >= 0) {
                const size_t at = free[(size_t)best].at;
                free[(size_t)best].at += n;
                free[(size_t)best].n -= n;
                if (!free[(size_t)best].n)
                    free.erase(free.begin() + best);
                return at;
            }
            // grow the last hole when it touches the top: a larger value reuses a smaller one's bytes
            if (!free.empty() && free.back().at + free.back().n == top) {
                const size_t at = free.back().at;
                top = at + n;
                free.pop_back();
                return at;
            }
        }
        const size_t at = top;
        top += n;
        return at;
    }
    void give(size_t at, size_t n)
    {
        if (!reuse)
            return;
        n = up(n ? n : 1);
        size_t i = 0;
        while (i < free.size() && free[i].at < at)
            ++i;
        free.insert(free.begin() + (long)i, Free{at, n});
        if (i + 1 < free.size() && free[i].at + free[i].n == free[i + 1].at) {
            free[i].n += free[i + 1].n;
            free.erase(free.begin() + (long)i + 1);
        }
        if (i > 0 && free[i - 1].at + free[i - 1].n == free[i].at) {
            free[i - 1].n += free[i].n;
            free.erase(free.begin() + (long)i);
        }
    }
};

constexpr int kForever = 0x7fffffff;

struct Compiler {
    csgn_circuit *c;
    uint64_t dl;
    std::vector<int> uses;            // operand slots of live nodes (and decrypt leaves) that read the value
    std::vector<char> retained;
    explicit Compiler(csgn_circuit *cc) : c(cc), dl(csgn_default_len(cc->n_bits)), uses(cc->values.size(), 0), retained(cc->values.size(), 0) {}

    bool tape() const { return c->flags == 0; }
    size_t value_bytes(const Value &v) const { return (size_t)((v.total ? v.total : 1) * dl * 8); }

    static bool reads_b(const Op &op) { return op.kind == 0 || op.kind == 1; }
    static bool reads_a(const Op &op) { return op.kind == 0 || op.kind == 1 || op.kind == 2 || op.kind == 3 || op.kind == 6; }

    void count_uses()
    {
        std::fill(uses.begin(), uses.end(), 0);
        for (const Op &op : c->ops) {
            if (op.elided)
                continue;
            if (op.kind == 2 && op.expr >= 0)
                continue;                                      // its leaves are counted below
            if (reads_a(op))
                uses[op.a] += 1;
            if (reads_b(op))
                uses[op.b] += 1;
        }
        for (const Expr &e : c->exprs)
            if (e.kind < 0)
                uses[e.value] += 1;
    }

    // Dec(v) as an expression over materialised values.  `depth` levels of producers may be dissolved.
    int expr_of(uint32_t v, int depth, std::map<uint32_t, int> &leaves)
    {
        const Value &val = c->values[v];
        const int p = val.producer;
        if (depth > 0 && !retained[v] && p >= 0 && (c->ops[(size_t)p].kind == 0 || c->ops[(size_t)p].kind == 1) && uses[v] == 1) {
            Op &prod = c->ops[(size_t)p];
            prod.elided = true;                                // v is never written: uses[v] stays 1, but nobody will look
            uses[v] = 0;
            Expr e = {};
            e.kind = prod.kind;
            const uint32_t a = prod.a, b = prod.b;
            // the producer's reads become this expression's: an operand read by the producer alone can dissolve further
            const int l = expr_of(a, depth - 1, leaves);
            const int r = expr_of(b, depth - 1, leaves);
            e.l = l;
            e.r = r;
            c->exprs.push_back(e);
            return (int)c->exprs.size() - 1;
        }
        std::map<uint32_t, int>::iterator it = leaves.find(v);
        if (it != leaves.end()) {                              // a * a: one decrypt of a, read twice by the combine
            uses[v] -= 1;
            return it->second;
        }
        Expr e = {};
        e.kind = -1;
        e.value = v;
        e.l = e.r = -1;
        c->exprs.push_back(e);
        leaves[v] = (int)c->exprs.size() - 1;
        return (int)c->exprs.size() - 1;
    }

    void fuse_decrypts()
    {
        const int depth = (c->flags & CSGN_CIRCUIT_PUSHDOWN) ? kForever : 1;
        for (Op &op : c->ops) {
            if (op.kind != 2)
                continue;
            std::map<uint32_t, int> leaves;
            const int root = expr_of(op.a, depth, leaves);
            if (c->exprs[(size_t)root].kind < 0) {             // nothing dissolved: the plain decrypt
                c->exprs.pop_back();
                continue;
            }
            op.expr = root;
        }
    }

    // nodes whose value nothing reads (any more)
    void drop_dead()
    {
        for (size_t i = c->ops.size(); i-- > 0;) {
            Op &op = c->ops[i];
            if (op.elided || op.kind == 2 || (op.kind == 5 && op.want_bits))
                continue;
            if (retained[op.out] || uses[op.out] > 0)
                continue;
            op.elided = true;
            if (reads_a(op))
                uses[op.a] -= 1;
            if (reads_b(op))
                uses[op.b] -= 1;
        }
    }

    bool placeable(uint32_t v) const
    {
        const Value &val = c->values[v];
        if (retained[v] || uses[v] != 1 || !val.per.empty() || val.producer < 0)
            return false;
        const Op &p = c->ops[(size_t)val.producer];
        return !p.elided && (p.kind == 0 || p.kind == 1);      // kernels with a per-element output pitch
    }

    void place()
    {
        for (size_t i = c->ops.size(); i-- > 0;) {
            Op &op = c->ops[i];
            if (op.kind != 0 || op.elided || !c->values[op.out].per.empty())
                continue;
            if (op.a == op.b)
                continue;                                       // a + a: read twice, written once -- copied
            if (placeable(op.a)) {
                c->values[op.a].parent = (int)op.out;
                c->values[op.a].parent_off = 0;
                op.placed_a = true;
            }
            if (placeable(op.b)) {
                c->values[op.b].parent = (int)op.out;
                c->values[op.b].parent_off = c->values[op.a].terms * dl;
                op.placed_b = true;
            }
        }
    }

    // operands of uniform adds that are circuit inputs: copied by the prologue launch
    void hoist()
    {
        c->n_hoist = 0;
        for (Op &op : c->ops) {
            if (op.kind != 0 || op.elided || !c->values[op.out].per.empty())
                continue;
            if (!op.placed_a && c->values[op.a].is_input && c->values[op.a].per.empty()) {
                op.hoist_a = true;
                c->n_hoist += 1;
            }
            if (!op.placed_b && c->values[op.b].is_input && c->values[op.b].per.empty()) {
                op.hoist_b = true;
                c->n_hoist += 1;
            }
        }
    }

    uint32_t root_of(uint32_t v, uint64_t *off) const
    {
        uint64_t o = 0;
        while (c->values[v].parent >= 0) {
            o += c->values[v].parent_off;
            v = (uint32_t)c->values[v].parent;
        }
        *off = o;
        return v;
    }

    // the nodes of one decrypt's expression, each once (a leaf may hang under several inner nodes)
    std::vector<int> nodes_of(int root) const
    {
        std::vector<int> out, stack(1, root);
        std::vector<char> seen(c->exprs.size(), 0);
        while (!stack.empty()) {
            const int ei = stack.back();
            stack.pop_back();
            if (seen[(size_t)ei])
                continue;
            seen[(size_t)ei] = 1;
            out.push_back(ei);
            if (c->exprs[(size_t)ei].kind >= 0) {
                stack.push_back(c->exprs[(size_t)ei].l);
                stack.push_back(c->exprs[(size_t)ei].r);
            }
        }
        return out;
    }

    // every region of the block: [first node that writes it, last node that reads it]
    struct Item {
        size_t bytes;
        int from, to;
        size_t *where;                 // receives the byte offset
        size_t at;                     // ... and a copy for the dump (`where` may point at a local)
    };
    std::vector<Item> laid;            // every region as laid out (csgn_circuit_plan_json)

    void layout()
    {
        const int nops = (int)c->ops.size();
        std::vector<Item> items;
        std::vector<int> from(c->values.size(), kForever), to(c->values.size(), -1);
        std::vector<size_t> at(c->values.size(), 0);
        // when a value's region is written and read, seen from its root
        for (int i = 0; i < nops; ++i) {
            const Op &op = c->ops[(size_t)i];
            if (op.elided)
                continue;
            auto read = [&](uint32_t v) { to[v] = std::max(to[v], i); };
            if (op.kind == 2 && op.expr >= 0) {
                for (int ei : nodes_of(op.expr))
                    if (c->exprs[(size_t)ei].kind < 0)
                        read(c->exprs[(size_t)ei].value);
            } else {
                if (reads_a(op))
                    read(op.a);
                if (reads_b(op))
                    read(op.b);
            }
            if (op.kind != 2) {
                uint64_t off;
                const uint32_t root = root_of(op.out, &off);
                const int when = (op.hoist_a || op.hoist_b) ? -1 : i;     // the prologue writes its slice before node 0
                from[root] = std::min(from[root], when);
                from[op.out] = std::min(from[op.out], when);
            }
        }
        for (size_t v = 0; v < c->values.size(); ++v) {
            Value &val = c->values[v];
            val.has_region = false;
            val.addressable = false;
            if (val.parent >= 0)
                continue;
            if (from[v] != kForever && to[v] < from[v])
                to[v] = from[v];                                // written by a live node, read by none (a fused Enc*Enc kept for its bits)
            const bool live = retained[v] || to[v] >= 0;
            if (!live)
                continue;
            val.has_region = true;
            Item it = {};
            it.bytes = value_bytes(val);
            it.from = val.is_input || retained[v] ? -1 : from[v];      // (-1 as well for a sum the prologue writes into)
            it.to = retained[v] ? kForever : to[v];
            if (val.is_input)
                it.from = -1;
            it.where = &at[v];
            items.push_back(it);
            if (val.needs_csr) {
                Item cs = it;
                cs.bytes = (size_t)(c->batch + 1) * 8;
                if (!val.dynamic) {                              // uploaded once at build: must outlive every run
                    cs.from = -1;
                    cs.to = kForever;
                }
                cs.where = &val.csr_offset;
                items.push_back(cs);
            }
        }
        // scratch and bit buffers
        for (int i = 0; i < nops; ++i) {
            Op &op = c->ops[(size_t)i];
            if (op.bits_id >= 0)
                items.push_back(Item{(size_t)c->batch, -1, kForever, &c->bits_offsets[(size_t)op.bits_id], 0});
            if (op.elided)
                continue;
            if (op.kind == 2 && op.expr < 0) {
                items.push_back(Item{csgn::decrypt_scratch_bytes(c->batch, c->values[op.a].total), i, i, &op.scratch, 0});
            } else if (op.kind == 2) {
                for (int ei : nodes_of(op.expr)) {
                    Expr &e = c->exprs[(size_t)ei];
                    if (ei != op.expr)                           // the root's bytes are the decrypt's result buffer
                        items.push_back(Item{(size_t)c->batch, i, i, &e.bits, 0});
                    if (e.kind < 0)
                        items.push_back(Item{csgn::decrypt_scratch_bytes(c->batch, c->values[e.value].total), i, i, &e.scratch, 0});
                }
            } else if (op.kind == 6) {
                items.push_back(Item{csgn::compact_scratch_bytes(c->n_bits, c->batch, c->values[op.out].total), i, i, &op.scratch, 0});
            } else if (op.kind == 1 && c->values[op.out].dynamic) {
                items.push_back(Item{(size_t)csgn::mul_ragged_async_plan_words(c->batch) * 8, i, i, &op.scratch, 0});
            }
        }
        if (c->has_encrypt)
            items.push_back(Item{8, -1, kForever, &c->epoch_offset, 0});
        if (c->n_hoist)
            items.push_back(Item{(size_t)c->n_hoist * sizeof(csgn::CopyEntry) + ((size_t)c->n_hoist + 1) * 4 + 16, -1, kForever,
                                 &c->hoist_offset, 0});

        Arena arena;
        // whatever outlives the run sits at the bottom and is never handed out again
        for (Item &it : items)
            if (it.to == kForever)
                *it.where = arena.take(it.bytes);
        arena.reuse = !tape() && (c->flags & CSGN_CIRCUIT_REUSE);
        // the rest along the chain of nodes: regions whose last reader has been emitted are free
        std::vector<Item *> timed;
        for (Item &it : items)
            if (it.to != kForever)
                timed.push_back(&it);
        std::stable_sort(timed.begin(), timed.end(), [](const Item *x, const Item *y) { return x->from < y->from; });
        std::vector<Item *> live;
        for (Item *it : timed) {
            for (size_t k = 0; k < live.size();) {
                if (live[k]->to < it->from) {
                    arena.give(*live[k]->where, live[k]->bytes);
                    live[k] = live.back();
                    live.pop_back();
                } else
                    ++k;
            }
            *it->where = arena.take(it->bytes);
            live.push_back(it);
        }
        c->bytes = arena.top;
        for (Item &it : items)
            it.at = *it.where;
        laid = items;
        // addresses of the values: roots from the allocator, placed values inside their root
        for (size_t v = 0; v < c->values.size(); ++v) {
            Value &val = c->values[v];
            if (val.parent < 0) {
                val.offset = at[v];
                val.pitch = val.terms * dl;
                val.addressable = val.has_region && (tape() || retained[v]);
                continue;
            }
            uint64_t off;
            const uint32_t root = root_of((uint32_t)v, &off);
            val.offset = at[root] + (size_t)off * 8;
            val.pitch = c->values[root].terms * dl;
        }
        for (const Op &op : c->ops)
            if (op.kind == 2 && op.expr >= 0)
                c->exprs[(size_t)op.expr].bits = c->bits_offsets[(size_t)op.bits_id];
    }

    void run()
    {
        for (Op &op : c->ops) {
            op.elided = op.placed_a = op.placed_b = op.hoist_a = op.hoist_b = false;
            op.expr = -1;
        }
        c->n_hoist = 0;
        for (Value &v : c->values) {
            v.parent = -1;
            v.parent_off = 0;
        }
        c->exprs.clear();
        for (size_t v = 0; v < c->values.size(); ++v)
            retained[v] = tape() || c->values[v].is_input || c->values[v].output;
        count_uses();
        if (!tape()) {
            if (c->flags & (CSGN_CIRCUIT_FUSE_DECRYPT | CSGN_CIRCUIT_PUSHDOWN))
                fuse_decrypts();
            count_uses();
            drop_dead();
            if (c->flags & CSGN_CIRCUIT_PLACE)
                place();
            if (c->flags & CSGN_CIRCUIT_HOIST)
                hoist();
        }
        layout();
    }
};

} // namespace

extern "C" {

int csgn_circuit_create(uint64_t n_bits, uint64_t batch, csgn_circuit **circuit)
{
    REQUIRE(circuit, "circuit is null");
    *circuit = nullptr;
    if (int rc = check_n(n_bits))
        return rc;
    REQUIRE(batch > 0, "batch must be > 0");
    csgn_circuit *c = new csgn_circuit();
    c->n_bits = n_bits;
    c->batch = batch;
    *circuit = c;
    return CSGN_OK;
}

void csgn_circuit_destroy(csgn_circuit *c)
{
    if (!c)
        return;
    if (c->exec)
        (void)hipGraphExecDestroy(c->exec);
    if (c->graph)
        (void)hipGraphDestroy(c->graph);
    if (c->block)
        (void)hipFree(c->block);
    for (auto &op : c->ops) {                 // the encrypt nodes' generator keys: not left in freed host memory
        volatile uint32_t *a = op.rng.key, *b = op.rng_b.key;
        for (int i = 0; i < 8; ++i)
            a[i] = b[i] = 0;
    }
    delete c;
}

int csgn_circuit_optimize(csgn_circuit *c, uint32_t flags)
{
    REQUIRE(c && !c->exec, "null circuit, or the circuit is already built");
    REQUIRE((flags & ~(uint32_t)(CSGN_CIRCUIT_ALL | CSGN_CIRCUIT_PUSHDOWN)) == 0, "unknown optimisation flag");
    REQUIRE(!(flags & CSGN_CIRCUIT_HOIST) || sizeof(csgn::CopyEntry) == 32, "internal: copy table layout");
    c->flags = flags;
    return CSGN_OK;
}

int csgn_circuit_output(csgn_circuit *c, uint32_t value)
{
    REQUIRE(c && !c->exec, "null circuit, or the circuit is already built");
    REQUIRE(value < c->values.size(), "value does not exist");
    c->values[value].output = true;
    return CSGN_OK;
}

int csgn_circuit_input(csgn_circuit *c, uint64_t terms, uint32_t *value)
{
    REQUIRE(c && value && !c->exec, "null circuit/value, or the circuit is already built");
    REQUIRE(terms > 0, "an input needs at least one term");
    const uint64_t dl = csgn_default_len(c->n_bits);
    if (!product_below(c->batch, terms, dl, 1ull << 57))
        return fail(CSGN_ERR_UNSUPPORTED, "input of %llu x %llu terms: size overflows",
                    (unsigned long long)c->batch, (unsigned long long)terms);
    c->values.push_back(make_uniform(c, terms, -1));
    *value = (uint32_t)(c->values.size() - 1);
    return CSGN_OK;
}

int csgn_circuit_input_ragged(csgn_circuit *c, const uint64_t *h_terms, uint32_t *value)
{
    REQUIRE(c && value && h_terms && !c->exec, "null argument, or the circuit is already built");
    const uint64_t dl = csgn_default_len(c->n_bits);
    Value v;
    v.is_input = true;
    v.needs_csr = true;
    v.per.assign(h_terms, h_terms + c->batch);
    for (uint64_t i = 0; i < c->batch; ++i) {
        REQUIRE(h_terms[i] < (1ull << 31), "element %llu has too many terms", (unsigned long long)i);
        v.total += h_terms[i];
        v.max_terms = h_terms[i] > v.max_terms ? h_terms[i] : v.max_terms;
    }
    if (!product_below(v.total ? v.total : 1, 1, dl, 1ull << 57))
        return fail(CSGN_ERR_UNSUPPORTED, "ragged input of %llu terms: size overflows", (unsigned long long)v.total);
    c->values.push_back(v);
    *value = (uint32_t)(c->values.size() - 1);
    return CSGN_OK;
}

static int circuit_binary(csgn_circuit *c, int kind, uint32_t a, uint32_t b, uint32_t *value)
{
    REQUIRE(c && value && !c->exec, "null circuit/value, or the circuit is already built");
    REQUIRE(a < c->values.size() && b < c->values.size(), "operand value does not exist");
    const uint64_t dl = csgn_default_len(c->n_bits);
    const bool ragged = !c->values[a].per.empty() || !c->values[b].per.empty();
    Op op = make_op(kind);
    op.a = a;
    op.b = b;
    if (ragged) {
        // per-element shapes are static: every downstream size is computed here, on the host
        Value v;
        v.per.resize(c->batch);
        for (uint64_t i = 0; i < c->batch; ++i) {
            const uint64_t ta = terms_of(c->values[a], i), tb = terms_of(c->values[b], i);
            if (kind && !product_below(ta, tb, dl, 1ull << 32))
                return fail(CSGN_ERR_UNSUPPORTED, "element %llu: product of %llu x %llu terms exceeds 2^32 words",
                            (unsigned long long)i, (unsigned long long)ta, (unsigned long long)tb);
            const uint64_t t = kind ? ta * tb : ta + tb;
            if (!kind && t * dl >= (1ull << 31))
                return fail(CSGN_ERR_UNSUPPORTED, "element %llu: sum of %llu terms exceeds 2^31 words",
                            (unsigned long long)i, (unsigned long long)t);
            v.per[i] = t;
            v.total += t;
            v.max_terms = t > v.max_terms ? t : v.max_terms;
        }
        if (!product_below(v.total ? v.total : 1, 1, dl, 1ull << 57))
            return fail(CSGN_ERR_UNSUPPORTED, "value of %llu terms: size overflows", (unsigned long long)v.total);
        v.dynamic = c->values[a].dynamic || c->values[b].dynamic;
        v.needs_csr = true;
        v.producer = (int)c->ops.size();
        c->values.push_back(v);
        c->values[a].needs_csr = c->values[b].needs_csr = true;
    } else {
        const uint64_t ta = c->values[a].terms, tb = c->values[b].terms;
        const uint64_t terms = kind ? ta * tb : ta + tb;
        if (kind && (ta >= (1ull << 31) || tb >= (1ull << 31) || !product_below(ta, tb, dl, 1ull << 32)))
            return fail(CSGN_ERR_UNSUPPORTED, "product of %llu x %llu terms exceeds 2^32 words",
                        (unsigned long long)ta, (unsigned long long)tb);
        if (!kind && (ta >= (1ull << 31) || tb >= (1ull << 31) || terms * dl >= (1ull << 31)))
            return fail(CSGN_ERR_UNSUPPORTED, "sum of %llu terms exceeds 2^31 words", (unsigned long long)terms);
        if (!product_below(c->batch, terms, dl, 1ull << 57))
            return fail(CSGN_ERR_UNSUPPORTED, "value of %llu x %llu terms: size overflows",
                        (unsigned long long)c->batch, (unsigned long long)terms);
        c->values.push_back(make_uniform(c, terms, (int)c->ops.size()));
    }
    op.out = (uint32_t)(c->values.size() - 1);
    c->ops.push_back(op);
    *value = op.out;
    return CSGN_OK;
}

int csgn_circuit_add(csgn_circuit *c, uint32_t a, uint32_t b, uint32_t *value) { return circuit_binary(c, 0, a, b, value); }
int csgn_circuit_mul(csgn_circuit *c, uint32_t a, uint32_t b, uint32_t *value) { return circuit_binary(c, 1, a, b, value); }

int csgn_circuit_compact(csgn_circuit *c, uint32_t a, uint32_t *value)
{
    REQUIRE(c && value && !c->exec, "null circuit/value, or the circuit is already built");
    REQUIRE(a < c->values.size(), "operand value does not exist");
    const Value &va = c->values[a];
    REQUIRE(va.total < (1ull << 31) && c->batch < (1ull << 31), "compaction handles fewer than 2^31 ciphertexts and terms");
    Value v;
    v.total = va.total;                              // bounds: nothing may cancel
    v.max_terms = va.max_terms;
    v.per.resize(c->batch);
    for (uint64_t i = 0; i < c->batch; ++i)
        v.per[i] = terms_of(va, i);
    v.dynamic = true;
    v.needs_csr = true;
    v.producer = (int)c->ops.size();
    c->values.push_back(v);
    c->values[a].needs_csr = true;
    Op op = make_op(6);
    op.a = a;
    op.b = a;
    op.out = (uint32_t)(c->values.size() - 1);
    c->ops.push_back(op);
    *value = op.out;
    return CSGN_OK;
}

int csgn_circuit_decrypt(csgn_circuit *c, uint32_t a, const uint64_t *d_mask, uint32_t *bits_id)
{
    REQUIRE(c && bits_id && d_mask && !c->exec, "null argument, or the circuit is already built");
    REQUIRE(a < c->values.size(), "operand value does not exist");
    Op op = make_op(2);
    op.a = a;
    op.mask = d_mask;
    c->bits_offsets.push_back(0);
    op.bits_id = (int)c->bits_offsets.size() - 1;
    c->ops.push_back(op);
    *bits_id = (uint32_t)op.bits_id;
    return CSGN_OK;
}

int csgn_circuit_permute(csgn_circuit *c, uint32_t a, const uint32_t *d_perm, uint32_t *value)
{
    REQUIRE(c && value && d_perm && !c->exec, "null argument, or the circuit is already built");
    REQUIRE(a < c->values.size(), "operand value does not exist");
    // reference semantics (src/Ciphertext.cpp:7-82): the result is ONE term, the permuted first term
    if (!c->values[a].per.empty())
        return fail(CSGN_ERR_UNSUPPORTED, "permutation of a ragged circuit value is not supported");
    c->values.push_back(make_uniform(c, 1, (int)c->ops.size()));
    Op op = make_op(3);
    op.a = a;
    op.out = (uint32_t)(c->values.size() - 1);
    op.mask = d_perm;
    c->ops.push_back(op);
    *value = op.out;
    return CSGN_OK;
}

int csgn_circuit_encrypt(csgn_circuit *c, uint64_t d, const uint8_t *d_plain, const uint64_t *d_key,
                         const uint64_t *d_mask, const csgn_rng *h_rng, uint64_t first_ciphertext,
                         uint32_t *value)
{
    REQUIRE(c && value && !c->exec, "null circuit/value, or the circuit is already built");
    REQUIRE(d_plain && d_key && d_mask && h_rng, "null argument");
    REQUIRE(d >= 1 && d < (1ull << 32), "d must be in [1, 2^32)");
    REQUIRE(h_rng->rounds == 8 || h_rng->rounds == 12 || h_rng->rounds == 20, "rng rounds must be 8, 12 or 20");
    REQUIRE(first_ciphertext + c->batch >= first_ciphertext && first_ciphertext + c->batch < (1ull << 56),
            "ciphertext index range too large");
    const uint64_t dl = csgn_default_len(c->n_bits);
    if (!product_below(c->batch, 1, dl, 1ull << 57))
        return fail(CSGN_ERR_UNSUPPORTED, "batch of %llu ciphertexts: size overflows", (unsigned long long)c->batch);
    c->has_encrypt = true;
    c->values.push_back(make_uniform(c, 1, (int)c->ops.size()));
    Op op = make_op(4);
    op.out = (uint32_t)(c->values.size() - 1);
    op.mask = d_mask;
    op.plain = d_plain;
    op.key = d_key;
    op.d = d;
    op.first = first_ciphertext;
    // The node encrypts under its OWN key, derived here from (h_rng->key, h_rng->nonce), and uses the
    // nonce words for nothing but the run number: run r of the node draws from (node key, nonce = r).
    // Round 2 added r to the caller's nonce, so run 2 under nonce N was run 1 under N + 1.
    op.rng = *h_rng;
    node_key_from(*h_rng, op.rng.key);
    op.rng.nonce = 0;
    c->ops.push_back(op);
    *value = op.out;
    return CSGN_OK;
}

int csgn_circuit_encrypt_mul(csgn_circuit *c, uint64_t d, const uint8_t *d_plain_a, const uint8_t *d_plain_b,
                             const uint64_t *d_key, const uint64_t *d_mask, const csgn_rng *h_rng_a,
                             const csgn_rng *h_rng_b, uint64_t first_ciphertext, uint32_t *value, uint32_t *bits_id)
{
    REQUIRE(c && value && !c->exec, "null circuit/value, or the circuit is already built");
    REQUIRE(d_plain_a && d_plain_b && d_key && d_mask && h_rng_a && h_rng_b, "null argument");
    REQUIRE(d >= 1 && d < (1ull << 32), "d must be in [1, 2^32)");
    REQUIRE(h_rng_a->rounds == 8 || h_rng_a->rounds == 12 || h_rng_a->rounds == 20, "rng rounds must be 8, 12 or 20");
    REQUIRE(h_rng_a->rounds == h_rng_b->rounds, "both generators must use the same number of rounds");
    REQUIRE(memcmp(h_rng_a->key, h_rng_b->key, sizeof(h_rng_a->key)) != 0 || h_rng_a->nonce != h_rng_b->nonce,
            "the two operands must draw from different streams (same key AND nonce given)");
    REQUIRE(first_ciphertext + c->batch >= first_ciphertext && first_ciphertext + c->batch < (1ull << 56),
            "ciphertext index range too large");
    const uint64_t dl = csgn_default_len(c->n_bits);
    if (!product_below(c->batch, 1, dl, 1ull << 57))
        return fail(CSGN_ERR_UNSUPPORTED, "batch of %llu ciphertexts: size overflows", (unsigned long long)c->batch);
    c->has_encrypt = true;
    c->values.push_back(make_uniform(c, 1, (int)c->ops.size()));
    Op op = make_op(5);
    op.out = (uint32_t)(c->values.size() - 1);
    op.mask = d_mask;
    op.plain = d_plain_a;
    op.plain_b = d_plain_b;
    op.key = d_key;
    op.d = d;
    op.first = first_ciphertext;
    // as csgn_circuit_encrypt: each operand under its own derived node key, nonce = run number
    op.rng = *h_rng_a;
    node_key_from(*h_rng_a, op.rng.key);
    op.rng.nonce = 0;
    op.rng_b = *h_rng_b;
    node_key_from(*h_rng_b, op.rng_b.key);
    op.rng_b.nonce = 0;
    op.want_bits = bits_id != nullptr;
    if (bits_id) {
        c->bits_offsets.push_back(0);
        op.bits_id = (int)c->bits_offsets.size() - 1;
        *bits_id = (uint32_t)op.bits_id;
    }
    c->ops.push_back(op);
    *value = op.out;
    return CSGN_OK;
}

uint64_t csgn_circuit_epoch(const csgn_circuit *c) { return c ? c->runs : 0; }

int csgn_circuit_node_key(const csgn_rng *h_rng, uint32_t h_node_key[8])
{
    REQUIRE(h_rng && h_node_key, "null argument");
    node_key_from(*h_rng, h_node_key);
    return CSGN_OK;
}

} // extern "C"

namespace {

// One decrypt expression, post-order: leaves are decrypts of materialised values, inner nodes one byte-wise kernel.
hipError_t emit_expr(csgn_circuit *c, unsigned char *base, const Op &op, int ei, std::vector<char> &done, uint64_t *alg_bytes,
                     uint64_t *kernels, hipStream_t s)
{
    Expr &e = c->exprs[(size_t)ei];
    if (done[(size_t)ei])
        return hipSuccess;
    done[(size_t)ei] = 1;
    const uint64_t dl = csgn_default_len(c->n_bits);
    if (e.kind < 0) {
        const Value &va = c->values[e.value];
        const u64 *A = reinterpret_cast<const u64 *>(base + va.offset);
        *alg_bytes += va.total * dl * 8 + c->batch;
        *kernels += 1;
        if (!va.per.empty())
            return csgn::decrypt(c->n_bits, c->batch, 0, va.total, A, reinterpret_cast<const u64 *>(base + va.csr_offset),
                                 (const u64 *)op.mask, base + e.bits, base + e.scratch, s, va.dynamic ? 0 : va.max_terms);
        return csgn::decrypt(c->n_bits, c->batch, va.terms, c->batch * va.terms, A, nullptr, (const u64 *)op.mask,
                             base + e.bits, base + e.scratch, s);
    }
    hipError_t err = emit_expr(c, base, op, e.l, done, alg_bytes, kernels, s);
    if (err == hipSuccess)
        err = emit_expr(c, base, op, e.r, done, alg_bytes, kernels, s);
    if (err != hipSuccess)
        return err;
    *alg_bytes += 3 * c->batch;
    *kernels += 1;
    return csgn::combine_bits(base + c->exprs[(size_t)e.l].bits, base + c->exprs[(size_t)e.r].bits, c->batch, e.kind == 1,
                              base + e.bits, s);
}

} // namespace

extern "C" {

int csgn_circuit_build(csgn_circuit *c)
{
    REQUIRE(c && !c->exec, "null circuit, or already built");
    REQUIRE(!c->ops.empty(), "the circuit has no operations");
    if (c->graph) {                     // an earlier build attempt failed at instantiation
        (void)hipGraphDestroy(c->graph);
        c->graph = nullptr;
    }
    if (c->block) {                     // ... or after the allocation
        (void)hipFree(c->block);
        c->block = nullptr;
    }
    Compiler(c).run();
    HIP_TRY(hipMalloc(&c->block, c->bytes ? c->bytes : 256));
    hipStream_t s = nullptr;
    {
        const hipError_t es = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (es != hipSuccess) {
            (void)hipFree(c->block);
            c->block = nullptr;
            return hip_fail(es, "hipStreamCreateWithFlags");
        }
    }
    unsigned char *base = static_cast<unsigned char *>(c->block);
    uint64_t *epoch = reinterpret_cast<uint64_t *>(base + c->epoch_offset);
    if (c->has_encrypt) {
        const hipError_t ez = hipMemset(epoch, 0, 8);
        if (ez != hipSuccess) {
            (void)hipStreamDestroy(s);
            return hip_fail(ez, "hipMemset (circuit epoch)");
        }
        c->runs = 0;
    }
    // CSR offsets of the ragged values (and of uniform values that meet one): known on the host
    for (size_t i = 0; i < c->values.size(); ++i) {
        const Value &v = c->values[i];
        if (!v.needs_csr || !v.has_region || v.dynamic)    // a dynamic value's offsets are written by the device
            continue;
        std::vector<uint64_t> off(c->batch + 1, 0);
        for (uint64_t k = 0; k < c->batch; ++k)
            off[k + 1] = off[k] + terms_of(v, k);
        const hipError_t eu = hipMemcpy(base + v.csr_offset, off.data(), off.size() * 8, hipMemcpyHostToDevice);
        if (eu != hipSuccess) {
            (void)hipStreamDestroy(s);
            return hip_fail(eu, "hipMemcpy (circuit CSR offsets)");
        }
    }
    const uint64_t dl = csgn_default_len(c->n_bits);
    uint64_t alg = 0, kernels = 0, placed = 0, fused = 0, dropped = 0;
    // the prologue's copy table (HOIST): every input operand of a uniform add, into its slice of the sum
    uint32_t hoist_blocks = 0;
    uint64_t hoist_alg = 0;
    if (c->n_hoist) {
        std::vector<csgn::CopyEntry> table;
        std::vector<u32> first(1, 0u);
        for (const Op &op : c->ops) {
            if (op.kind != 0 || op.elided || !(op.hoist_a || op.hoist_b))
                continue;
            const Value &va = c->values[op.a], &vb = c->values[op.b], &vo = c->values[op.out];
            for (int side = 0; side < 2; ++side) {
                if (!(side ? op.hoist_b : op.hoist_a))
                    continue;
                const Value &vs = side ? vb : va;
                csgn::CopyEntry en;
                en.src = reinterpret_cast<const u64 *>(base + vs.offset);
                en.dst = reinterpret_cast<u64 *>(base + vo.offset) + (side ? va.terms * dl : 0);
                en.elem_words = (u32)(vs.terms * dl);
                en.src_pitch = (u32)(vs.terms * dl);
                en.dst_pitch = (u32)vo.pitch;
                en.batch = (u32)c->batch;
                table.push_back(en);
                first.push_back(first.back() + csgn::copy_list_blocks(en));
                hoist_alg += 2 * c->batch * vs.terms * dl * 8;
            }
        }
        hoist_blocks = first.back();
        hipError_t eh = hipMemcpy(base + c->hoist_offset, table.data(), table.size() * sizeof(csgn::CopyEntry), hipMemcpyHostToDevice);
        if (eh == hipSuccess)
            eh = hipMemcpy(base + c->hoist_offset + table.size() * sizeof(csgn::CopyEntry), first.data(), first.size() * 4,
                           hipMemcpyHostToDevice);
        if (eh != hipSuccess || table.size() != c->n_hoist || c->batch >= (1ull << 32)) {
            (void)hipStreamDestroy(s);
            return hip_fail(eh != hipSuccess ? eh : hipErrorInvalidValue, "hipMemcpy (circuit prologue table)");
        }
    }
    std::vector<char> done(c->exprs.size(), 0);
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess && c->has_encrypt) {
        e = csgn::bump_epoch((u64 *)epoch, s);      // every replay encrypts under (node key, nonce = its own run number)
        kernels += 1;
    }
    if (e == hipSuccess && c->n_hoist) {
        e = csgn::copy_list(reinterpret_cast<const csgn::CopyEntry *>(base + c->hoist_offset),
                            reinterpret_cast<const u32 *>(base + c->hoist_offset + (size_t)c->n_hoist * sizeof(csgn::CopyEntry)),
                            c->n_hoist, hoist_blocks, s);
        alg += hoist_alg;
        kernels += 1;
    }
    for (size_t i = 0; e == hipSuccess && i < c->ops.size(); ++i) {
        const Op &op = c->ops[i];
        if (op.elided) {
            dropped += 1;
            continue;
        }
        if (op.kind == 4) {
            uint64_t *O = reinterpret_cast<uint64_t *>(base + c->values[op.out].offset);
            e = csgn::encrypt_keyed(c->n_bits, op.d, c->batch, op.first, op.plain, (const u64 *)op.key,
                                    (const u64 *)op.mask, op.rng.key, op.rng.nonce, op.rng.rounds,
                                    (const u64 *)epoch, (u64 *)O, s);
            alg += c->batch * (dl * 8 + 1);
            kernels += 1;
            continue;
        }
        if (op.kind == 5) {
            uint64_t *O = reinterpret_cast<uint64_t *>(base + c->values[op.out].offset);
            e = csgn::encrypt_mul_keyed(c->n_bits, op.d, c->batch, op.first, op.plain, op.plain_b, (const u64 *)op.key,
                                        (const u64 *)op.mask, op.rng.key, op.rng.nonce, op.rng_b.key, op.rng_b.nonce,
                                        op.rng.rounds, (const u64 *)epoch, (u64 *)O,
                                        op.want_bits ? base + c->bits_offsets[(size_t)op.bits_id] : nullptr, s);
            alg += c->batch * (dl * 8 + 2);
            kernels += 1;
            continue;
        }
        const Value &va = c->values[op.a];
        const uint64_t *A = reinterpret_cast<const uint64_t *>(base + va.offset);
        if (op.kind == 6) {
            const Value &vo = c->values[op.out];
            e = va.total ? csgn::compact(c->n_bits, c->batch, va.total, va.max_terms, (const u64 *)A,
                                         reinterpret_cast<const u64 *>(base + va.csr_offset),
                                         reinterpret_cast<u64 *>(base + vo.offset),
                                         reinterpret_cast<u64 *>(base + vo.csr_offset), base + op.scratch, s)
                         : csgn::circuit_zero_words(reinterpret_cast<u64 *>(base + vo.csr_offset), c->batch + 1, s);
            alg += 2 * va.total * dl * 8;              // the static bound: nothing may cancel
            kernels += 1;
            continue;
        }
        if (op.kind == 2) {
            uint8_t *bits = base + c->bits_offsets[(size_t)op.bits_id];
            if (op.expr >= 0) {
                fused += 1;
                e = emit_expr(c, base, op, op.expr, done, &alg, &kernels, s);
            } else if (!va.per.empty()) {
                e = csgn::decrypt(c->n_bits, c->batch, 0, va.total, (const u64 *)A,
                                  reinterpret_cast<const u64 *>(base + va.csr_offset), (const u64 *)op.mask,
                                  bits, base + op.scratch, s, va.dynamic ? 0 : va.max_terms);
            } else {
                e = csgn::decrypt(c->n_bits, c->batch, va.terms, c->batch * va.terms, (const u64 *)A, nullptr,
                                  (const u64 *)op.mask, bits, base + op.scratch, s);
            }
            if (op.expr < 0) {
                alg += va.total * dl * 8 + c->batch;
                kernels += 1;
            }
            continue;
        }
        const Value &vo = c->values[op.out];
        if (op.kind == 3) {
            e = csgn::permute(c->n_bits, c->batch, va.terms, false, (const u64 *)A, (const u32 *)op.mask,
                              reinterpret_cast<u64 *>(base + vo.offset), s);
            alg += 2 * c->batch * dl * 8;
            kernels += 1;
            continue;
        }
        const Value &vb = c->values[op.b];
        const u64 *B = reinterpret_cast<const u64 *>(base + vb.offset);
        u64 *O = reinterpret_cast<u64 *>(base + vo.offset);
        if (!vo.per.empty()) {
            // ragged add / multiply: the CSR forms, offsets already in the block
            const u64 *oa = reinterpret_cast<const u64 *>(base + va.csr_offset);
            const u64 *ob = reinterpret_cast<const u64 *>(base + vb.csr_offset);
            u64 *oo = reinterpret_cast<u64 *>(base + vo.csr_offset);
            if (op.kind && vo.dynamic)
                // sizes known to the device only: plan kernels + CSR multiply back to back, the static bound sizes the launch
                e = csgn::mul_ragged_async(c->n_bits, c->batch, (const u64 *)A, oa, B, ob, O, oo, vo.total,
                                           reinterpret_cast<u64 *>(base + op.scratch), s);
            else if (op.kind)
                e = vo.total ? csgn::mul_ragged(c->n_bits, c->batch, (const u64 *)A, oa, B, ob, O, oo, va.max_terms,
                                                vb.max_terms, vo.total, s, nullptr, va.total + vb.total)
                             : hipSuccess;
            else
                e = csgn::add_ragged(c->n_bits, c->batch, (const u64 *)A, oa, B, ob, O, oo, vo.total, s, vo.dynamic,
                                     vo.dynamic ? 0 : va.max_terms, vo.dynamic ? 0 : vb.max_terms);
            alg += op.kind ? (va.total + vb.total + vo.total) * dl * 8 : 2 * vo.total * dl * 8;
            kernels += 1;
            continue;
        }
        const uint64_t ta = va.terms, tb = vb.terms;
        const bool dense = vo.pitch == vo.terms * dl;
        if (op.kind) {
            e = csgn::mul_uniform(c->n_bits, c->batch, ta, tb, (const u64 *)A, B, O, 0, s, dense ? 0 : vo.pitch);
            alg += c->batch * (ta + tb + ta * tb) * dl * 8;
            kernels += 1;
            continue;
        }
        // uniform add: whatever neither its producer nor the prologue has already written into the sum is copied into its slice
        placed += (op.placed_a ? 1u : 0u) + (op.placed_b ? 1u : 0u);
        const bool done_a = op.placed_a || op.hoist_a, done_b = op.placed_b || op.hoist_b;
        if (done_a && done_b)
            continue;
        if (!done_a && !done_b) {
            e = csgn::add_uniform(c->n_bits, c->batch, ta, tb, (const u64 *)A, B, O, s, dense ? 0 : vo.pitch);
            alg += 2 * c->batch * (ta + tb) * dl * 8;
        } else if (done_a) {
            e = csgn::add_uniform(c->n_bits, c->batch, 0, tb, nullptr, B, O + ta * dl, s, vo.pitch);
            alg += 2 * c->batch * tb * dl * 8;
        } else {
            e = csgn::add_uniform(c->n_bits, c->batch, ta, 0, (const u64 *)A, nullptr, O, s, vo.pitch);
            alg += 2 * c->batch * ta * dl * 8;
        }
        kernels += 1;
    }
    hipGraph_t g = nullptr;
    const hipError_t e2 = hipStreamEndCapture(s, &g);
    (void)hipStreamDestroy(s);
    if (e != hipSuccess || e2 != hipSuccess) {
        if (g)
            (void)hipGraphDestroy(g);
        return hip_fail(e != hipSuccess ? e : e2, "csgn_circuit_build (stream capture)");
    }
    const hipError_t e3 = hipGraphInstantiate(&c->exec, g, nullptr, nullptr, 0);
    if (e3 != hipSuccess) {
        (void)hipGraphDestroy(g);
        c->exec = nullptr;
        return hip_fail(e3, "hipGraphInstantiate");
    }
    c->graph = g;
    uint64_t regions = 0;
    for (const Value &v : c->values)
        regions += v.has_region ? 1u : 0u;
    c->stats[0] = (uint64_t)c->bytes;
    c->stats[1] = alg;
    c->stats[2] = (uint64_t)c->ops.size();
    c->stats[3] = kernels;
    c->stats[4] = placed;
    c->stats[5] = fused;
    c->stats[6] = dropped;
    (void)regions;
    c->stats[7] = c->n_hoist;
    return CSGN_OK;
}

uint64_t csgn_circuit_block_bytes(const csgn_circuit *c) { return (c && c->block) ? (uint64_t)c->bytes : 0; }

int csgn_circuit_stats(const csgn_circuit *c, uint64_t h_stats[8])
{
    REQUIRE(c && h_stats, "null argument");
    REQUIRE(c->exec, "the circuit is not built");
    for (int i = 0; i < 8; ++i)
        h_stats[i] = c->stats[i];
    return CSGN_OK;
}

int csgn_circuit_plan_json(csgn_circuit *c, char *h_json, size_t cap)
{
    REQUIRE(c && h_json && cap, "null argument");
    REQUIRE(!c->exec, "the circuit is already built");
    REQUIRE(!c->ops.empty(), "the circuit has no operations");
    Compiler comp(c);
    comp.run();
    std::string out = "{\"bytes\": " + std::to_string(c->bytes) + ", \"values\": [";
    for (size_t v = 0; v < c->values.size(); ++v) {
        const Value &val = c->values[v];
        out += std::string(v ? ", " : "") + "{\"region\": " + (val.has_region ? "true" : "false") + ", \"addressable\": " +
               (val.addressable ? "true" : "false") + ", \"offset\": " + std::to_string(val.offset) + ", \"pitch\": " +
               std::to_string(val.pitch) + ", \"parent\": " + std::to_string(val.parent) + ", \"terms\": " +
               std::to_string(val.terms) + ", \"total\": " + std::to_string(val.total) + "}";
    }
    out += "], \"ops\": [";
    for (size_t i = 0; i < c->ops.size(); ++i) {
        const Op &op = c->ops[i];
        out += std::string(i ? ", " : "") + "{\"kind\": " + std::to_string(op.kind) + ", \"a\": " + std::to_string(op.a) +
               ", \"b\": " + std::to_string(op.b) + ", \"out\": " + std::to_string(op.out) + ", \"elided\": " +
               (op.elided ? "true" : "false") + ", \"placed_a\": " + (op.placed_a ? "true" : "false") + ", \"placed_b\": " +
               (op.placed_b ? "true" : "false") + ", \"hoist_a\": " + (op.hoist_a ? "true" : "false") + ", \"hoist_b\": " +
               (op.hoist_b ? "true" : "false") + ", \"expr\": " + std::to_string(op.expr) + "}";
    }
    out += "], \"exprs\": [";
    for (size_t i = 0; i < c->exprs.size(); ++i) {
        const Expr &e = c->exprs[i];
        out += std::string(i ? ", " : "") + "{\"kind\": " + std::to_string(e.kind) + ", \"value\": " + std::to_string(e.value) +
               ", \"l\": " + std::to_string(e.l) + ", \"r\": " + std::to_string(e.r) + "}";
    }
    out += "], \"regions\": [";
    for (size_t i = 0; i < comp.laid.size(); ++i) {
        const Compiler::Item &it = comp.laid[i];
        out += std::string(i ? ", " : "") + "{\"at\": " + std::to_string(it.at) + ", \"bytes\": " + std::to_string(it.bytes) +
               ", \"from\": " + std::to_string(it.from) + ", \"to\": " + std::to_string(it.to) + "}";
    }
    out += "]}";
    if (out.size() + 1 > cap)
        return fail(CSGN_ERR_INVALID, "csgn_circuit_plan_json: %zu bytes needed, %zu given", out.size() + 1, cap);
    memcpy(h_json, out.c_str(), out.size() + 1);
    return CSGN_OK;
}

uint64_t *csgn_circuit_value(csgn_circuit *c, uint32_t value)
{
    if (!c || !c->block || value >= c->values.size() || !c->values[value].addressable)
        return nullptr;
    return reinterpret_cast<uint64_t *>(static_cast<unsigned char *>(c->block) + c->values[value].offset);
}

uint64_t csgn_circuit_value_terms(csgn_circuit *c, uint32_t value)
{
    return (c && value < c->values.size()) ? c->values[value].terms : 0;
}

uint64_t csgn_circuit_value_total_terms(csgn_circuit *c, uint32_t value)
{
    return (c && value < c->values.size()) ? c->values[value].total : 0;
}

const uint64_t *csgn_circuit_value_offsets(csgn_circuit *c, uint32_t value)
{
    if (!c || !c->block || value >= c->values.size() || c->values[value].per.empty() || !c->values[value].addressable)
        return nullptr;
    return reinterpret_cast<const uint64_t *>(static_cast<unsigned char *>(c->block) + c->values[value].csr_offset);
}

uint8_t *csgn_circuit_bits(csgn_circuit *c, uint32_t bits_id)
{
    if (!c || !c->block || bits_id >= c->bits_offsets.size())
        return nullptr;
    return static_cast<unsigned char *>(c->block) + c->bits_offsets[bits_id];
}

int csgn_circuit_run(csgn_circuit *c, void *stream)
{
    REQUIRE(c && c->exec, "the circuit is not built");
    HIP_TRY(hipGraphLaunch(c->exec, S(stream)));
    c->runs += 1;
    return CSGN_OK;
}

} // extern "C"
