// csgn_compact.hip -- EXTENSION: mod-2 compaction of term lists (not reference behaviour).
// Hand-written CDNA4 (gfx950) HIP; shared helpers in csgn_device.h, design notes in DESIGN.md 4.8.
//
// Decryption XORs over terms (/root/reference/src/SecretKey.cpp:139), so a term occurring an even number
// of times contributes nothing and one occurring an odd number of times contributes once: each
// ciphertext is rewritten as its distinct odd-multiplicity terms, in order of first occurrence.  The
// reference never does this (add is pure concatenation, src/Ciphertext.cpp:107-122), so it is opt-in
// and never runs on a parity path.
//
// Shape of the work: 8*dL*T_in bytes read, 8*dL*T_out written; everything else must stay small beside
// that.  The batch is cut into GROUPS of consecutive ciphertexts of at most capT terms (one ciphertext
// of up to capT terms, or a run of smaller ones), and one 1024-thread workgroup takes one group:
//   1. its units go from HBM into REGISTERS, lane l holding units l, l+1024, ... (coalesced 16-byte
//      loads, R of them in flight per lane) -- they are read ONCE;
//   2. every unit is hashed where it sits (64x64->128 multiply-fold with a position tweak) and the
//      U partial hashes of a term are summed in LDS (ds_add_u64);
//   3. one lane per term inserts {48-bit tag, term index} into the ciphertext's own region of an
//      open-addressing table in LDS (2 slots per term): ds_cmpst claims an empty slot, ds_min keeps
//      the SMALLEST index of a tag class, ds_xor counts the class's parity;
//   4. every unit whose term joined somebody else's slot is compared with the same unit of that
//      representative (one 16-byte load from L2, U lanes per term, coalesced).  A difference means a
//      tag collision between unequal terms: the group is redone with full compares inside the probe
//      loop (exact, slow, practically never taken: 2^-48 per pair);
//   5. survivors = representatives of odd classes; a ballot scan ranks them; the group's output
//      offset comes from a decoupled look-back over one 8-byte {flag, count} granule per group
//      (tickets are handed out in group order, so a group only ever waits for groups that are
//      already running);
//   6. the registers are stored to their final place (16-byte stores, U lanes per term).
// Ciphertexts of more than capT terms ("large") cannot be deduplicated inside one workgroup: their
// terms are hashed chunk by chunk into an HBM table first (k_cl_*: the same protocol with global
// atomics), and the main kernel then moves their chunks exactly like a group, reading the keep
// decision from that table instead of making it.  Their terms are read twice.
#include "csgn_device.h"

namespace csgn {

namespace {

constexpr u32 kCT = 1024;                 // threads of a group's workgroup
constexpr int kCR = 10;                   // units per lane held in registers: 10 240 units = 1024 terms at N=1247
constexpr u32 kCapUnits = kCT * kCR;
constexpr u32 kMaxGroupTerms = 1536;      // LDS: 40 B per term + tables stay under 64 KiB

// control words (u64 each) at the head of the scratch block; zeroed, with the status granules, per call
enum { kCtrlTicket = 0, kCtrlGroups = 1, kCtrlChunks = 2, kCtrlCollision = 3, kCtrlWords = 32 };

struct Geom {
    u32 U;          // units per term
    u32 capT;       // terms per group at most
    u32 wshift;     // log2 of the window: ciphertexts whose first terms share a window may share a group
    u32 bigT;       // a ciphertext of more terms than this is a group by itself: capT - window
};

Geom make_geom(u32 U)
{
    Geom g;
    g.U = U;
    g.capT = min(kMaxGroupTerms, kCapUnits / U);
    g.wshift = 0;
    while ((2u << g.wshift) <= g.capT / 2)
        ++g.wshift;
    g.bigT = g.capT - (1u << g.wshift);
    return g;
}

// groups a call can produce at most: one per window, two per big ciphertext, the chunks of large ones
u64 group_bound(u64 total_terms, const Geom &g) { return 4 + 10 * (total_terms / g.capT + 1); }

struct Layout {
    u64 *ctrl, *status, *groups, *chunks, *partial, *hash, *tab;
    u32 *gpos, *par, *slot_of;
    size_t bytes, head_bytes;
};

Layout make_layout(void *scratch, u64 batch, u64 total_terms, const Geom &g)
{
    auto up = [](uintptr_t x) { return (x + 255) & ~(uintptr_t)255; };
    const u64 ng = group_bound(total_terms, g);
    uintptr_t p = up(reinterpret_cast<uintptr_t>(scratch));
    const uintptr_t p0 = p;
    auto take = [&](size_t bytes) {
        const uintptr_t at = p;
        p = up(p + bytes);
        return at;
    };
    Layout l;
    l.ctrl = reinterpret_cast<u64 *>(take((kCtrlWords + ng) * 8));
    l.status = l.ctrl + kCtrlWords;
    l.head_bytes = (kCtrlWords + ng) * 8;
    l.groups = reinterpret_cast<u64 *>(take(ng * 8));
    l.chunks = reinterpret_cast<u64 *>(take(ng * 8));
    l.gpos = reinterpret_cast<u32 *>(take(batch * 4));
    l.partial = reinterpret_cast<u64 *>(take((batch / 256 + 2) * 8));
    l.hash = reinterpret_cast<u64 *>(take(total_terms * 8));
    l.tab = reinterpret_cast<u64 *>(take(total_terms * 16));
    l.par = reinterpret_cast<u32 *>(take(total_terms * 8));
    l.slot_of = reinterpret_cast<u32 *>(take(total_terms * 4));
    l.bytes = (p - p0) + 256;
    return l;
}

// ------------------------------------------------------------------------------- hashing
constexpr u64 kHashA = 0xA0761D6478BD642Full, kHashB = 0xE7037ED1A0B428DBull, kHashP = 0x8EBC6AF09C88C6E3ull;

__device__ inline u64 fold_mul(u64 a, u64 b)
{
    const unsigned __int128 m = (unsigned __int128)a * b;
    return (u64)m ^ (u64)(m >> 64);
}
// contribution of unit k of a term to the term's hash; the contributions are summed mod 2^64
__device__ inline u64 unit_hash(unit16 v, u32 k)
{
    const u64 pk = (u64)(k + 1u) * kHashP;
    const u64 lo = ((u64)v.y << 32) | v.x, hi = ((u64)v.w << 32) | v.z;
    return fold_mul(lo ^ (kHashA + pk), hi ^ kHashB ^ ((pk >> 23) | (pk << 41)));
}
__device__ inline u64 unit_hash(unit8 v, u32 k)
{
    const u64 pk = (u64)(k + 1u) * kHashP;
    return fold_mul(v ^ (kHashA + pk), kHashB ^ ((pk >> 23) | (pk << 41)));
}
__device__ inline bool unit_same(unit16 a, unit16 b)
{
    const unit16 d = a ^ b;
    return (d.x | d.y | d.z | d.w) == 0u;
}
__device__ inline bool unit_same(unit8 a, unit8 b) { return a == b; }
template <typename Unit>
__device__ inline Unit unit_zero();
template <>
__device__ inline unit16 unit_zero<unit16>()
{
    unit16 z = {0u, 0u, 0u, 0u};
    return z;
}
template <>
__device__ inline unit8 unit_zero<unit8>()
{
    return 0ull;
}

__device__ inline bool words_equal(const u64 *x, const u64 *y, u32 dL)
{
    bool same = true;
    for (u32 k = 0; k < dL; ++k)
        same = same && (x[k] == y[k]);
    return same;
}

__device__ inline unsigned long long *ull(u64 *p) { return reinterpret_cast<unsigned long long *>(p); }

// ----------------------------------------------------------------- the groups of a batch
// How many groups START at ciphertext c (0: it continues the run before it).  A run = consecutive
// ciphertexts of at most bigT terms whose first terms lie in one window of 2^wshift terms: together
// they span fewer than capT terms.  A ciphertext of more than bigT terms is alone; one of more than
// capT terms ("large") is cut into chunks of capT terms, one group each.
__device__ inline u32 group_count(const u64 *__restrict__ off, u32 c, Geom g, bool &large)
{
    const u64 o0 = off[c], t = off[c + 1] - o0;
    large = t > g.capT;
    bool start = c == 0;
    if (!start) {
        const u64 om = off[c - 1];
        start = t > g.bigT || o0 - om > g.bigT || (o0 >> g.wshift) != (om >> g.wshift);
    }
    if (!start)
        return 0u;
    return large ? (u32)((t + g.capT - 1) / g.capT) : 1u;
}

__device__ inline u32 wave_incl_scan(u32 v)
{
    const u32 lane = threadIdx.x & (kWave - 1);
    for (u32 d = 1; d < kWave; d <<= 1) {
        const u32 n = __shfl_up(v, d, kWave);
        if (lane >= d)
            v += n;
    }
    return v;
}

__global__ void __launch_bounds__(256) k_cg_count(u32 batch, const u64 *__restrict__ off, Geom g,
                                                  u32 *__restrict__ gpos, u64 *__restrict__ partial)
{
    __shared__ u32 wsum[4];
    const u32 c = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    bool large;
    const u32 n = c < batch ? group_count(off, c, g, large) : 0u;
    const u32 incl = wave_incl_scan(n);
    if (lane == kWave - 1)
        wsum[wave] = incl;
    __syncthreads();
    u32 base = 0, tot = 0;
    for (u32 w = 0; w < 4; ++w) {
        base += w < wave ? wsum[w] : 0u;
        tot += wsum[w];
    }
    if (c < batch)
        gpos[c] = base + incl - n;                    // block-local; k_cg_fill adds the block's base
    if (threadIdx.x == 0)
        partial[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(1024) k_cg_scan(u64 nblocks, u64 *__restrict__ partial, u64 *__restrict__ ctrl)
{
    __shared__ u64 part[1024];
    const u32 tid = threadIdx.x;
    const u64 chunk = (nblocks + 1023) / 1024;
    const u64 c0 = min(nblocks, (u64)tid * chunk), c1 = min(nblocks, c0 + chunk);
    u64 sum = 0;
    for (u64 c = c0; c < c1; ++c)
        sum += partial[c];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        u64 run = 0;
        for (u32 t = 0; t < 1024; ++t) {
            const u64 v = part[t];
            part[t] = run;
            run += v;
        }
        ctrl[kCtrlGroups] = run;
    }
    __syncthreads();
    u64 run = part[tid];
    for (u64 c = c0; c < c1; ++c) {
        const u64 v = partial[c];
        partial[c] = run;
        run += v;
    }
}

// group descriptor = first ciphertext | chunk << 32; the chunks of large ciphertexts are listed a
// second time (any order) for the k_cl_* kernels
__global__ void __launch_bounds__(256) k_cg_fill(u32 batch, const u64 *__restrict__ off, Geom g,
                                                 const u32 *__restrict__ gpos, const u64 *__restrict__ partial,
                                                 u64 *__restrict__ groups, u64 *__restrict__ chunks,
                                                 u64 *__restrict__ ctrl)
{
    const u32 c = blockIdx.x * 256u + threadIdx.x;
    if (c >= batch)
        return;
    bool large;
    const u32 n = group_count(off, c, g, large);
    if (n == 0)
        return;
    const u64 base = partial[blockIdx.x] + gpos[c];
    for (u32 i = 0; i < n; ++i)
        groups[base + i] = (u64)c | ((u64)i << 32);
    if (large) {
        const u64 at = atomicAdd(ull(ctrl + kCtrlChunks), (unsigned long long)n);
        for (u32 i = 0; i < n; ++i)
            chunks[at + i] = (u64)c | ((u64)i << 32);
    }
}

// ---------------------------------------------------------------------- the main kernel
struct CompactArgs {
    const void *terms;
    void *out;
    const u64 *off;
    u64 *off_out;
    u64 *ctrl;
    u64 *status;
    const u64 *groups;
    const u64 *tab;          // HBM table of the large ciphertexts (k_cl_*), two slots per term
    const u32 *par;
    const u32 *slot_of;
    u64 tag_mask;            // all ones; a test narrows it to force tag collisions
    Geom g;
    FastDiv dU;
    u32 batch;
    u32 dL;
    u32 large_ready;         // the k_cl_* kernels ran: large ciphertexts have a keep decision
};

constexpr u64 kFlagAggregate = 1ull << 62, kFlagPrefix = 2ull << 62, kValueMask = (1ull << 62) - 1;

// Decoupled look-back (one wave): publish this group's count, add up the counts of the groups before
// it back to the nearest one whose inclusive prefix is known, publish the own inclusive prefix.  Every
// granule is ONE 8-byte agent-scope store/load carrying flag and value together, so no ordering
// between data and flag is needed.  Returns the exclusive prefix in every lane.
__device__ inline u64 lookback(u64 *status, u32 gid, u64 count)
{
    const u32 lane = threadIdx.x & (kWave - 1);
    if (lane == 0)
        __hip_atomic_store(status + gid, kFlagAggregate | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u64 excl = 0;
    long long base = (long long)gid - 1;
    while (base >= 0) {
        const long long idx = base - (long long)lane;
        const u64 v = idx >= 0 ? __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : kFlagPrefix;              // before group 0: prefix 0
        const u32 flag = (u32)(v >> 62);
        const u64 pending = __ballot(flag == 0u), known = __ballot(flag == 2u);
        const u32 first = known ? (u32)__builtin_ctzll(known) : kWave;     // nearest lane holding a prefix
        const u64 need = first < kWave - 1 ? (2ull << first) - 1ull : ~0ull;
        if (pending & need) {
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        u64 mine = lane <= first ? (v & kValueMask) : 0ull;
        for (u32 d = 32; d > 0; d >>= 1)
            mine += __shfl_xor(mine, d, kWave);
        excl += mine;
        if (first < kWave)
            break;
        base -= kWave;
    }
    if (lane == 0)
        __hip_atomic_store(status + gid, kFlagPrefix | (excl + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// One lane per term: find the term's slot in its ciphertext's region [2*a0, 2*b0) of the LDS table.
// EXACT compares the words of a tag match before joining its class (the redo after a collision).
template <bool EXACT>
__device__ inline u32 lds_insert(u64 *s_tab, u32 *s_par, u64 h, u64 tag_mask, u32 t, u32 a0, u32 b0,
                                 const u64 *words, u32 dL)
{
    const u32 ns = 2u * (b0 - a0);
    u32 slot = 2u * a0 + (u32)(((u64)(u32)h * ns) >> 32);
    const u64 tag = (h >> 16) & tag_mask;
    const u64 entry = (tag << 16) | (u64)(t + 1u);
    for (;;) {
        const u64 cur = atomicCAS(ull(s_tab + slot), 0ull, entry);
        if (cur == 0ull)
            break;
        if ((cur >> 16) == tag) {
            bool same = true;
            if (EXACT) {
                const u32 r = (u32)(cur & 0xFFFFu) - 1u;
                same = words_equal(words + (u64)t * dL, words + (u64)r * dL, dL);
            }
            if (same) {
                atomicMin(ull(s_tab + slot), entry);
                break;
            }
        }
        slot = slot + 1u == 2u * b0 ? 2u * a0 : slot + 1u;
    }
    atomicXor(s_par + slot, 1u);
    return slot;
}

template <typename Unit, int R>
__global__ void __launch_bounds__(kCT) k_compact_main(CompactArgs a)
{
    extern __shared__ u64 s_dyn[];
    const u32 capT = a.g.capT, U = a.g.U;
    u64 *s_hash = s_dyn;                                          // [capT] hash sums, then {slot, representative}
    u64 *s_tab = s_hash + capT;                                   // [2 capT] {tag, smallest index + 1}
    u32 *s_par = reinterpret_cast<u32 *>(s_tab + 2 * capT);       // [2 capT] parity of the class
    u32 *s_rk = s_par + 2 * capT;                                 // [capT + 1] rank << 1 | keep
    u32 *s_coff = s_rk + capT + 2;                                // [capT + 2] ciphertext starts inside the group
    __shared__ u32 s_ticket, s_flag, s_wsum[kCT / kWave];
    __shared__ u64 s_prefix;
    const Unit *__restrict__ terms = static_cast<const Unit *>(a.terms);
    Unit *__restrict__ out = static_cast<Unit *>(a.out);
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const u32 ngroups = (u32)a.ctrl[kCtrlGroups];
    constexpr int PASSES = (kMaxGroupTerms + kCT - 1) / kCT;      // term passes of the one-lane-per-term steps

    for (;;) {
        if (tid == 0) {
            s_ticket = atomicAdd(reinterpret_cast<u32 *>(a.ctrl + kCtrlTicket), 1u);
            s_flag = 0u;
        }
        __syncthreads();
        const u32 gid = s_ticket;
        if (gid >= ngroups)
            break;
        // opaque copy of the thread index: everything derived from it (unit -> term, position tweaks of
        // the hash) would otherwise be hoisted out of the ticket loop, ten units at a time, and spill
        u32 ltid = tid;
        asm volatile("" : "+v"(ltid));
        const u64 desc = a.groups[gid];
        const u32 c0 = (u32)desc, chunk = (u32)(desc >> 32);
        const u64 o0 = a.off[c0], o1 = a.off[c0 + 1];
        const bool large = o1 - o0 > capT;
        u32 c1;
        u64 tb, te;
        if (large) {
            c1 = c0 + 1u;
            tb = o0 + (u64)chunk * capT;
            te = min(tb + capT, o1);
        } else {
            c1 = gid + 1u < ngroups ? (u32)a.groups[gid + 1u] : a.batch;
            tb = o0;
            te = a.off[c1];
        }
        const u32 nt = (u32)(te - tb), nunits = nt * U, ncts = c1 - c0;
        const u64 ub = tb * U;
        const bool multi = !large && ncts > 1u, staged = multi && ncts <= capT + 1u;

        // 1. the group's units -> registers
        Unit reg[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const u32 j = (u32)i * kCT + ltid;
            reg[i] = j < nunits ? terms[ub + j] : unit_zero<Unit>();
        }
        for (u32 x = tid; x < nt; x += kCT)
            s_hash[x] = 0ull;
        for (u32 x = tid; x < 2u * nt; x += kCT) {
            s_tab[x] = 0ull;
            s_par[x] = 0u;
        }
        if (staged)
            for (u32 x = tid; x <= ncts; x += kCT)
                s_coff[x] = (u32)(a.off[c0 + x] - tb);
        __syncthreads();

        u32 keep[PASSES];
        if (!large) {
            // 2. term hashes
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const u32 j = (u32)i * kCT + ltid;
                if (j < nunits) {
                    const u32 t = csgn_fastdiv(j, a.dU), k = j - t * U;
                    atomicAdd(ull(s_hash + t), unit_hash(reg[i], k));
                }
                __builtin_amdgcn_sched_barrier(0);                // one hash at a time: ten interleaved ones spill
            }
            __syncthreads();
            // 3. one lane per term: its ciphertext's table region, then the insert
            u32 a0[PASSES], b0[PASSES], slot[PASSES];
            u64 h[PASSES];
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const u32 t = (u32)p * kCT + tid;
                a0[p] = 0u;
                b0[p] = nt;
                if (t < nt) {
                    if (staged) {                                 // largest i with s_coff[i] <= t
                        u32 lo = 0u, hi = ncts;
                        while (hi - lo > 1u) {
                            const u32 mid = (lo + hi) >> 1;
                            if (s_coff[mid] <= t)
                                lo = mid;
                            else
                                hi = mid;
                        }
                        a0[p] = s_coff[lo];
                        b0[p] = s_coff[lo + 1u];
                    } else if (multi) {
                        const u32 c = csr_find(a.off, c0, c1, tb + t);
                        a0[p] = (u32)(a.off[c] - tb);
                        b0[p] = (u32)(a.off[c + 1u] - tb);
                    }
                    h[p] = csgn_splitmix64(s_hash[t]);
                    slot[p] = lds_insert<false>(s_tab, s_par, h[p], a.tag_mask, t, a0[p], b0[p], nullptr, 0u);
                }
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const u32 t = (u32)p * kCT + tid;
                if (t < nt)
                    s_hash[t] = ((u64)slot[p] << 32) | (u64)((u32)(s_tab[slot[p]] & 0xFFFFu) - 1u);
            }
            __syncthreads();
            // 4. every unit of a term that joined a class is compared with its representative's
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const u32 j = (u32)i * kCT + ltid;
                if (j < nunits) {
                    const u32 t = csgn_fastdiv(j, a.dU), k = j - t * U;
                    const u32 rep = (u32)s_hash[t];
                    if (rep != t && !unit_same(terms[(tb + rep) * U + k], reg[i]))
                        s_flag = 1u;
                }
            }
            __syncthreads();
            if (s_flag) {                                         // a tag collision: redo with full compares
                for (u32 x = tid; x < 2u * nt; x += kCT) {
                    s_tab[x] = 0ull;
                    s_par[x] = 0u;
                }
                __syncthreads();
                const u64 *words = reinterpret_cast<const u64 *>(a.terms) + tb * a.dL;
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    const u32 t = (u32)p * kCT + tid;
                    if (t < nt)
                        slot[p] = lds_insert<true>(s_tab, s_par, h[p], a.tag_mask, t, a0[p], b0[p], words, a.dL);
                }
                __syncthreads();
            }
            // 5. survivors: the smallest index of a class of odd size
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const u32 t = (u32)p * kCT + tid;
                keep[p] = 0u;
                if (t < nt)
                    keep[p] = ((u32)(s_tab[slot[p]] & 0xFFFFu) - 1u == t && (s_par[slot[p]] & 1u)) ? 1u : 0u;
            }
        } else {
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const u32 t = (u32)p * kCT + tid;
                keep[p] = 0u;
                if (t < nt) {
                    keep[p] = 1u;                                 // no decision made: the chunk is copied as it is
                    if (a.large_ready) {
                        const u64 g = tb + t;
                        const u32 s = a.slot_of[g];
                        keep[p] = ((u32)a.tab[s] - 1u == (u32)g && (a.par[s] & 1u)) ? 1u : 0u;
                    }
                }
            }
        }
        // ranks of the survivors
        u32 count = 0u;
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            if ((u32)p * kCT < nt) {                              // uniform
                const u32 t = (u32)p * kCT + tid;
                const u64 m = __ballot(keep[p] != 0u);
                const u32 before = (u32)__popcll(m & ((1ull << lane) - 1ull));
                if (lane == 0)
                    s_wsum[wave] = (u32)__popcll(m);
                __syncthreads();
                u32 wbase = 0u, tot = 0u;
                for (u32 w = 0; w < kCT / kWave; ++w) {
                    const u32 v = s_wsum[w];
                    wbase += w < wave ? v : 0u;
                    tot += v;
                }
                if (t < nt)
                    s_rk[t] = ((count + wbase + before) << 1) | keep[p];
                count += tot;
                __syncthreads();
            }
        }
        if (tid == 0)
            s_rk[nt] = count << 1;
        // where the group's survivors go
        if (wave == 0) {
            const u64 excl = lookback(a.status, gid, count);
            if (lane == 0)
                s_prefix = excl;
        }
        __syncthreads();
        const u64 prefix = s_prefix;
        if (!large) {
            for (u32 x = tid; x < ncts; x += kCT) {
                const u32 rel = staged ? s_coff[x] : (u32)(a.off[c0 + x] - tb);
                a.off_out[c0 + x] = prefix + (s_rk[rel] >> 1);
            }
        } else if (chunk == 0u && tid == 0) {
            a.off_out[c0] = prefix;
        }
        if (gid == ngroups - 1u && tid == 0)
            a.off_out[a.batch] = prefix + count;
        // 6. registers -> their final place
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const u32 j = (u32)i * kCT + ltid;
            if (j < nunits) {
                const u32 t = csgn_fastdiv(j, a.dU), k = j - t * U;
                const u32 rk = s_rk[t];
                if (rk & 1u)
                    out[(prefix + (rk >> 1)) * U + k] = reg[i];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------- large ciphertexts: the table in HBM
struct LargeArgs {
    const void *terms;
    const u64 *off;
    u64 *ctrl;
    const u64 *chunks;
    u64 *hash;
    u64 *tab;
    u32 *par;
    u32 *slot_of;
    u64 tag_mask;
    Geom g;
    FastDiv dU;
    u32 dL;
};

struct Chunk {
    u32 c;
    u64 c_begin, c_terms;     // the ciphertext
    u64 tb;                   // the chunk's first term
    u32 nt;
};
__device__ inline Chunk chunk_of(const LargeArgs &a, u64 ci)
{
    const u64 desc = a.chunks[ci];
    Chunk k;
    k.c = (u32)desc;
    k.c_begin = a.off[k.c];
    k.c_terms = a.off[k.c + 1] - k.c_begin;
    k.tb = k.c_begin + (desc >> 32) * a.g.capT;
    k.nt = (u32)min((u64)a.g.capT, k.c_begin + k.c_terms - k.tb);
    return k;
}

// hash of every term of every chunk (the first read of a large ciphertext) + its table slots cleared
template <typename Unit>
__global__ void __launch_bounds__(kCT) k_cl_hash(LargeArgs a)
{
    extern __shared__ u64 s_dyn[];
    u64 *s_hash = s_dyn;
    const u64 nchunks = a.ctrl[kCtrlChunks];
    const Unit *__restrict__ terms = static_cast<const Unit *>(a.terms);
    const u32 tid = threadIdx.x, U = a.g.U;
    for (u64 ci = blockIdx.x; ci < nchunks; ci += gridDim.x) {
        const Chunk k = chunk_of(a, ci);
        const u32 nunits = k.nt * U;
        for (u32 x = tid; x < k.nt; x += kCT)
            s_hash[x] = 0ull;
        for (u32 x = tid; x < 2u * k.nt; x += kCT) {
            a.tab[2 * k.tb + x] = 0ull;
            a.par[2 * k.tb + x] = 0u;
        }
        __syncthreads();
        for (u32 j = tid; j < nunits; j += kCT) {
            const u32 t = csgn_fastdiv(j, a.dU), kk = j - t * U;
            atomicAdd(ull(s_hash + t), unit_hash(terms[k.tb * U + j], kk));
        }
        __syncthreads();
        for (u32 x = tid; x < k.nt; x += kCT)
            a.hash[k.tb + x] = csgn_splitmix64(s_hash[x]);
        __syncthreads();
    }
}

// one lane per term: insert into the ciphertext's region [2*begin, 2*(begin+terms)) of the HBM table;
// entries are {32-bit tag, GLOBAL term index + 1}
template <bool EXACT>
__global__ void __launch_bounds__(256) k_cl_insert(LargeArgs a)
{
    const u64 nchunks = a.ctrl[kCtrlChunks];
    if (EXACT && a.ctrl[kCtrlCollision] == 0ull)
        return;
    const u64 *words = static_cast<const u64 *>(a.terms);
    for (u64 ci = blockIdx.x; ci < nchunks; ci += gridDim.x) {
        const Chunk k = chunk_of(a, ci);
        const u64 base = 2 * k.c_begin, ns = 2 * k.c_terms;
        for (u32 t = threadIdx.x; t < k.nt; t += 256u) {
            const u64 g = k.tb + t, h = a.hash[g];
            u64 slot = ((u64)(u32)h * ns) >> 32;
            const u64 tag = (h >> 32) & a.tag_mask;
            const u64 entry = (tag << 32) | (g + 1);
            for (;;) {
                u64 cur = __hip_atomic_load(a.tab + base + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == 0ull) {
                    cur = atomicCAS(ull(a.tab + base + slot), 0ull, entry);
                    if (cur == 0ull)
                        break;
                }
                if ((cur >> 32) == tag) {
                    bool same = true;
                    if (EXACT)
                        same = words_equal(words + g * a.dL, words + (u64)((u32)cur - 1u) * a.dL, a.dL);
                    if (same) {
                        atomicMin(ull(a.tab + base + slot), entry);
                        break;
                    }
                }
                slot = slot + 1 == ns ? 0 : slot + 1;
            }
            atomicXor(a.par + base + slot, 1u);
            a.slot_of[g] = (u32)(base + slot);
        }
    }
}

// every unit of a term that joined a class against the same unit of the class's smallest member
template <typename Unit>
__global__ void __launch_bounds__(256) k_cl_verify(LargeArgs a)
{
    const u64 nchunks = a.ctrl[kCtrlChunks];
    const Unit *__restrict__ terms = static_cast<const Unit *>(a.terms);
    const u32 U = a.g.U;
    bool bad = false;
    for (u64 ci = blockIdx.x; ci < nchunks; ci += gridDim.x) {
        const Chunk k = chunk_of(a, ci);
        const u32 nunits = k.nt * U;
        for (u32 j = threadIdx.x; j < nunits; j += 256u) {
            const u32 t = csgn_fastdiv(j, a.dU), kk = j - t * U;
            const u64 g = k.tb + t;
            const u64 rep = (u64)((u32)a.tab[a.slot_of[g]] - 1u);
            if (rep != g && !unit_same(terms[g * U + kk], terms[rep * U + kk]))
                bad = true;
        }
    }
    if (bad)
        a.ctrl[kCtrlCollision] = 1ull;
}

__global__ void __launch_bounds__(256) k_cl_clear(LargeArgs a)
{
    const u64 nchunks = a.ctrl[kCtrlChunks];
    if (a.ctrl[kCtrlCollision] == 0ull)
        return;
    for (u64 ci = blockIdx.x; ci < nchunks; ci += gridDim.x) {
        const Chunk k = chunk_of(a, ci);
        for (u32 x = threadIdx.x; x < 2u * k.nt; x += 256u) {
            a.tab[2 * k.tb + x] = 0ull;
            a.par[2 * k.tb + x] = 0u;
        }
    }
}

template <typename Unit>
hipError_t compact_launch(u32 U, u64 dL, u64 batch, u64 total_terms, u64 max_terms, const u64 *terms,
                          const u64 *off, u64 *out, u64 *off_out, void *scratch, hipStream_t s)
{
    const Geom g = make_geom(U);
    const Layout l = make_layout(scratch, batch, total_terms, g);
    hipError_t e;
    if ((e = hipMemsetAsync(l.ctrl, 0, l.head_bytes, s)) != hipSuccess)
        return e;
    const u32 cblocks = ceil_div_u64(batch, 256);
    k_cg_count<<<cblocks, 256, 0, s>>>((u32)batch, off, g, l.gpos, l.partial);
    k_cg_scan<<<1, 1024, 0, s>>>(cblocks, l.partial, l.ctrl);
    k_cg_fill<<<cblocks, 256, 0, s>>>((u32)batch, off, g, l.gpos, l.partial, l.groups, l.chunks, l.ctrl);

    // tag bits kept by the tables: all of them, unless a test asks for collisions
    const int bits = tune(TUNE_COMPACT_TAG_BITS);
    const u64 tag_mask = (bits > 0 && bits < 48) ? (1ull << bits) - 1ull : ~0ull;
    const u64 ng = group_bound(total_terms, g);
    const bool maybe_large = (max_terms ? max_terms : total_terms) > g.capT;
    if (maybe_large) {
        LargeArgs la;
        la.terms = terms;
        la.off = off;
        la.ctrl = l.ctrl;
        la.chunks = l.chunks;
        la.hash = l.hash;
        la.tab = l.tab;
        la.par = l.par;
        la.slot_of = l.slot_of;
        la.tag_mask = tag_mask;
        la.g = g;
        la.dU = csgn_fastdiv_make(U);
        la.dL = (u32)dL;
        const u32 grid = (u32)min(ng, (u64)2048);
        k_cl_hash<Unit><<<min(grid, 512u), kCT, g.capT * 8, s>>>(la);
        k_cl_insert<false><<<grid, 256, 0, s>>>(la);
        k_cl_verify<Unit><<<grid, 256, 0, s>>>(la);
        k_cl_clear<<<grid, 256, 0, s>>>(la);
        k_cl_insert<true><<<grid, 256, 0, s>>>(la);
    }
    CompactArgs a;
    a.terms = terms;
    a.out = out;
    a.off = off;
    a.off_out = off_out;
    a.ctrl = l.ctrl;
    a.status = l.status;
    a.groups = l.groups;
    a.tab = l.tab;
    a.par = l.par;
    a.slot_of = l.slot_of;
    a.tag_mask = tag_mask;
    a.g = g;
    a.dU = csgn_fastdiv_make(U);
    a.batch = (u32)batch;
    a.dL = (u32)dL;
    a.large_ready = maybe_large ? 1u : 0u;
    const size_t lds = (size_t)g.capT * 40 + 64;
    k_compact_main<Unit, kCR><<<(u32)min(ng, (u64)512), kCT, lds, s>>>(a);
    return hipGetLastError();
}

} // namespace

// ------------------------------------------------------------------------------ public
bool compact_supported(u64 n_bits)
{
    const u64 dL = (n_bits + 63) / 64;
    return dL <= kCapUnits / 2;                    // a group must hold two terms at least (8-byte units)
}

size_t compact_scratch_bytes(u64 n_bits, u64 batch, u64 total_terms)
{
    const u64 dL = (n_bits + 63) / 64;
    // the 8-byte-unit geometry has the smaller groups, hence the larger bound
    const Geom g = make_geom((u32)min(dL, (u64)kCapUnits / 2));
    return make_layout(nullptr, batch, total_terms, g).bytes;
}

hipError_t compact(u64 n_bits, u64 batch, u64 total_terms, u64 max_terms, const u64 *terms, const u64 *off,
                   u64 *out, u64 *off_out, void *scratch, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0)
        return hipSuccess;
    // term and slot indices (up to 2 * total_terms) are kept in 32 bits
    if (batch >= (1ull << 31) || total_terms >= (1ull << 31) || !compact_supported(n_bits))
        return hipErrorInvalidValue;
    if (dL % 2 == 0 && aligned16(terms) && aligned16(out))
        return compact_launch<unit16>((u32)(dL / 2), dL, batch, total_terms, max_terms, terms, off, out, off_out,
                                      scratch, s);
    return compact_launch<unit8>((u32)dL, dL, batch, total_terms, max_terms, terms, off, out, off_out, scratch, s);
}

} // namespace csgn
