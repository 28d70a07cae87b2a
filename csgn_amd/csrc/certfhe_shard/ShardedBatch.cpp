// ShardedBatch.cpp -- one logical batch over the GPUs of a node (extension, see ShardedBatch.h).
// Built on the two C ABIs only (csgn_hip.h for the kernels, csgn_shard.h for partition + RCCL) and on
// the public SecretKey/Context interface; one host thread, one HIP stream and one RCCL communicator
// per GPU.  -> libcertFHE_shard.so
#include "ShardedBatch.h"

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "csgn_hip.h"
#include "csgn_shard.h"

namespace certFHE {
namespace detail {

namespace {
std::string lastErrors()
{
    return std::string(csgn_last_error()) + " / " + csgn_shard_last_error();
}

void ck(int rc, const char *what)
{
    if (rc != CSGN_OK)
        throw std::runtime_error(std::string(what) + " failed [" + std::to_string(rc) + "]: " + lastErrors());
}
} // namespace

// One GPU of the group: its thread, its communicator, its stream, a pool of HBM blocks that are
// handed out again in stream order (hipMalloc/hipFree per operation would cost more than a
// million-element kernel).
struct ShardWorker {
    int rank = 0, device = 0;
    csgn_comm *comm = nullptr;
    void *stream = nullptr;
    std::thread thread;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()> > queue;
    bool quit = false;
    std::atomic<bool> fail_next{false};                       // test hook (set by the caller's thread)
    std::vector<std::pair<size_t, void *> > pool;             // (bytes, ptr) free blocks of this GPU
    size_t pooled = 0;

    void *take(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes == 0)
            bytes = 256;
        for (size_t i = 0; i < pool.size(); ++i)
            if (pool[i].first == bytes) {
                void *p = pool[i].second;
                pool[i] = pool.back();
                pool.pop_back();
                pooled -= bytes;
                return p;
            }
        void *p = nullptr;
        int rc = csgn_malloc(&p, bytes);
        if (rc != CSGN_OK && pooled) {                        // out of HBM: drop the pool and retry once
            drop();
            rc = csgn_malloc(&p, bytes);
        }
        ck(rc, "csgn_malloc");
        return p;
    }
    void give(void *p, size_t bytes)
    {
        if (!p)
            return;
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes == 0)
            bytes = 256;
        if (pooled + bytes > ((size_t)64 << 30)) {
            csgn_free(p);
            return;
        }
        pool.push_back(std::make_pair(bytes, p));
        pooled += bytes;
    }
    void drop()
    {
        for (size_t i = 0; i < pool.size(); ++i)
            csgn_free(pool[i].second);
        pool.clear();
        pooled = 0;
    }
};

struct ShardGroupImpl {
    std::vector<std::unique_ptr<ShardWorker> > w;
    std::atomic<bool> dead{false};
    std::mutex err_mu;
    std::string first_error;
    int rccl_runtime = 0, rccl_header = 0;
    std::string rccl_path;

    ~ShardGroupImpl()
    {
        for (auto &x : w) {
            {
                std::lock_guard<std::mutex> g(x->mu);
                x->quit = true;
            }
            x->cv.notify_all();
        }
        for (auto &x : w)
            if (x->thread.joinable())
                x->thread.join();
        for (auto &x : w) {
            x->drop();
            csgn_comm_destroy(x->comm);
        }
    }

    void abortAll()
    {
        for (auto &x : w)
            if (x->comm)
                (void)csgn_comm_abort(x->comm);
    }

    void recordFailure(int rank, const std::string &what)
    {
        {
            std::lock_guard<std::mutex> g(err_mu);
            if (first_error.empty())
                first_error = "rank " + std::to_string(rank) + " (device " + std::to_string(w[rank]->device) + "): " + what;
        }
        dead.store(true);
        abortAll();                                           // releases every peer blocked in a collective
    }

    void threadMain(ShardWorker *me)
    {
        bool up = false;
        for (;;) {
            std::function<void()> task;
            {
                std::unique_lock<std::mutex> lk(me->mu);
                me->cv.wait(lk, [&] { return me->quit || !me->queue.empty(); });
                if (me->queue.empty())
                    return;
                task = std::move(me->queue.front());
                me->queue.pop_front();
            }
            if (!up) {
                (void)csgn_init(me->device);                  // binds this thread to its GPU; errors surface in the task
                up = true;
            }
            task();
        }
    }

    // f(worker) on every GPU's thread; returns when all have finished.  A task that throws kills the
    // group (abortAll) and the call rethrows the FIRST failure after every thread has come back.
    void runAll(const std::function<void(ShardWorker &)> &f)
    {
        if (dead.load())
            throw std::runtime_error("certFHE::ShardGroup is dead after an earlier failure: " + first_error);
        std::mutex done_mu;
        std::condition_variable done_cv;
        size_t left = w.size();
        for (auto &x : w) {
            ShardWorker *me = x.get();
            auto task = [this, me, &f, &done_mu, &done_cv, &left] {
                try {
                    if (me->fail_next.exchange(false)) {
                        throw std::runtime_error("injected failure (ShardGroup::injectFailure)");
                    }
                    if (dead.load())
                        throw std::runtime_error("group already failed");
                    f(*me);
                } catch (const std::exception &e) {
                    recordFailure(me->rank, e.what());
                } catch (...) {
                    recordFailure(me->rank, "unknown exception");
                }
                std::lock_guard<std::mutex> g(done_mu);
                --left;
                done_cv.notify_all();
            };
            {
                std::lock_guard<std::mutex> g(me->mu);
                me->queue.push_back(task);
            }
            me->cv.notify_all();
        }
        std::unique_lock<std::mutex> lk(done_mu);
        done_cv.wait(lk, [&] { return left == 0; });
        lk.unlock();
        if (dead.load())
            throw std::runtime_error("certFHE::ShardedBatch: " + first_error);
    }

    // fire-and-forget on one worker (returning blocks to its pool)
    void post(int rank, const std::function<void()> &f)
    {
        ShardWorker *me = w[rank].get();
        {
            std::lock_guard<std::mutex> g(me->mu);
            if (me->quit)
                return;
            me->queue.push_back(f);
        }
        me->cv.notify_all();
    }
};

// The shards of one logical batch: rank r holds elements [lo[r], hi[r]) as (hi-lo)*terms*dL words.
struct ShardedData {
    std::shared_ptr<ShardGroupImpl> g;
    Context ctx;
    uint64_t count, terms;
    std::vector<void *> ptr;
    std::vector<size_t> bytes;
    std::vector<uint64_t> lo, hi;

    ShardedData(const std::shared_ptr<ShardGroupImpl> &grp, const Context &c, uint64_t n, uint64_t t)
        : g(grp), ctx(c), count(n), terms(t), ptr(grp->w.size(), nullptr), bytes(grp->w.size(), 0),
          lo(grp->w.size(), 0), hi(grp->w.size(), 0)
    {
        for (size_t r = 0; r < g->w.size(); ++r)
            ck(csgn_shard_range(count, (int)r, (int)g->w.size(), &lo[r], &hi[r]), "csgn_shard_range");
    }
    ~ShardedData()
    {
        for (size_t r = 0; r < ptr.size(); ++r) {
            if (!ptr[r])
                continue;
            void *p = ptr[r];
            const size_t b = bytes[r];
            ShardWorker *me = g->w[r].get();
            g->post((int)r, [me, p, b] { me->give(p, b); });   // back to that GPU's pool, on its own thread
        }
    }
    uint64_t mine(int r) const { return hi[r] - lo[r]; }
    uint64_t *words(int r) const { return static_cast<uint64_t *>(ptr[r]); }
    // allocate this rank's shard (called on the rank's thread)
    void alloc(ShardWorker &me)
    {
        bytes[me.rank] = (size_t)(mine(me.rank) * terms * ctx.getDefaultN() * 8);
        ptr[me.rank] = me.take(bytes[me.rank]);
    }
};

} // namespace detail

using detail::ShardedData;
using detail::ShardGroupImpl;
using detail::ShardWorker;
using detail::ck;

// ------------------------------------------------------------------------------ ShardGroup

ShardGroup::ShardGroup(const std::vector<int> &devices) : impl(std::make_shared<ShardGroupImpl>())
{
    int visible = 0;
    ck(csgn_comm_device_count(&visible), "csgn_comm_device_count");
    if (visible <= 0)
        throw std::runtime_error("certFHE::ShardGroup: no HIP device visible; there is no CPU fallback");
    std::vector<int> devs = devices;
    if (devs.empty())
        for (int i = 0; i < visible; ++i)
            devs.push_back(i);
    char path[1024] = "";
    ck(csgn_comm_rccl_info(&impl->rccl_runtime, &impl->rccl_header, path, sizeof(path)), "csgn_comm_rccl_info");
    impl->rccl_path = path;
    std::vector<csgn_comm *> comms(devs.size(), nullptr);
    // strict: the RCCL this process bound must be the one libcsgn_shard.so was built for
    const int rc = csgn_comm_init_all_ex((int)devs.size(), devs.data(), CSGN_COMM_STRICT, comms.data());
    if (rc != CSGN_OK) {
        const std::string msg = detail::lastErrors();
        for (size_t i = 0; i < comms.size(); ++i)
            if (comms[i])
                csgn_comm_destroy(comms[i]);
        throw std::runtime_error("certFHE::ShardGroup: csgn_comm_init_all failed [" + std::to_string(rc) + "]: " + msg);
    }
    for (size_t i = 0; i < devs.size(); ++i) {
        std::unique_ptr<ShardWorker> x(new ShardWorker());
        x->rank = (int)i;
        x->device = devs[i];
        x->comm = comms[i];
        x->stream = csgn_comm_stream(comms[i]);
        impl->w.push_back(std::move(x));
    }
    ShardGroupImpl *raw = impl.get();
    for (auto &x : impl->w) {
        ShardWorker *me = x.get();
        me->thread = std::thread([raw, me] { raw->threadMain(me); });
    }
    // bring every GPU up (and fail here, not in the first operation, if one is not a gfx950 part)
    impl->runAll([](ShardWorker &me) { ck(csgn_init(me.device), "csgn_init"); });
}

int ShardGroup::size() const { return (int)impl->w.size(); }
int ShardGroup::device(int rank) const { return impl->w.at(rank)->device; }
bool ShardGroup::healthy() const { return !impl->dead.load(); }

std::string ShardGroup::collective() const
{
    auto v = [](int c) { return std::to_string(c / 10000) + "." + std::to_string((c / 100) % 100) + "." + std::to_string(c % 100); };
    return "RCCL " + v(impl->rccl_runtime) + " (" + impl->rccl_path + "), header " + v(impl->rccl_header);
}

void ShardGroup::setTimeoutMs(uint64_t ms)
{
    for (auto &x : impl->w)
        ck(csgn_comm_set_timeout_ms(x->comm, ms), "csgn_comm_set_timeout_ms");
}

void ShardGroup::injectFailure(int rank) { impl->w.at(rank)->fail_next.store(true); }

void ShardGroup::setTuning(const std::string &key, int value)
{
    // knobs are per host thread (csgn_hip.h): the group's kernels run on its worker threads, not on the caller's.
    // The name is checked here first: a typo must not count as a rank failure (which aborts the communicators).
    int unused = 0;
    if (csgn_get_tuning(key.c_str(), &unused) != CSGN_OK)
        throw std::invalid_argument("certFHE::ShardGroup::setTuning: no knob named '" + key + "'");
    impl->runAll([&](ShardWorker &) { ck(csgn_set_tuning(key.c_str(), value), "csgn_set_tuning"); });
}

void ShardGroup::forceGroupedBroadcast(bool on)
{
    for (auto &x : impl->w)
        ck(csgn_comm_set_option(x->comm, CSGN_COMM_OPT_FORCE_GROUPED_BROADCAST, on ? 1 : 0), "csgn_comm_set_option");
}

// ------------------------------------------------------------------------------ ShardedBatch

namespace {

struct KeyMaterial {                       // host copies; wiped by the destructor
    std::vector<uint64_t> idx, mask;
    uint64_t n, d, dl;
    explicit KeyMaterial(const SecretKey &key, const Context &ctx)
        : n(ctx.getN()), d(key.getLength()), dl(ctx.getDefaultN())
    {
        idx.assign(key.getKey(), key.getKey() + d);
        mask.assign(dl, 0);
        ck(csgn_key_mask(n, idx.data(), d, mask.data()), "csgn_key_mask");
    }
    ~KeyMaterial()
    {
        volatile uint64_t *a = idx.data();
        for (size_t i = 0; i < idx.size(); ++i)
            a[i] = 0;
        volatile uint64_t *b = mask.data();
        for (size_t i = 0; i < mask.size(); ++i)
            b[i] = 0;
    }
};

// [key indices (d words)][mask (dl words)][bytes...] in one pooled block of the rank, wiped on release
struct Staging {
    ShardWorker &me;
    void *p;
    size_t bytes;
    uint64_t d, dl;
    Staging(ShardWorker &w, const KeyMaterial &k, size_t extra_bytes) : me(w), d(k.d), dl(k.dl)
    {
        bytes = (size_t)(d + dl) * 8 + ((extra_bytes + 7) & ~(size_t)7);
        p = me.take(bytes);
        ck(csgn_memcpy_h2d(p, k.idx.data(), (size_t)d * 8, me.stream), "csgn_memcpy_h2d");
        ck(csgn_memcpy_h2d(static_cast<uint64_t *>(p) + d, k.mask.data(), (size_t)dl * 8, me.stream), "csgn_memcpy_h2d");
    }
    const uint64_t *key() const { return static_cast<const uint64_t *>(p); }
    const uint64_t *mask() const { return static_cast<const uint64_t *>(p) + d; }
    uint8_t *extra() const { return reinterpret_cast<uint8_t *>(static_cast<uint64_t *>(p) + d + dl); }
    ~Staging()
    {
        (void)csgn_memset(p, 0, (size_t)(d + dl) * 8, me.stream);   // the block held the secret indices
        me.give(p, bytes);
    }
};

} // namespace

const Context &ShardedBatch::keyContext(const SecretKey &key)
{
    if (!key.certFHEContext)
        throw std::logic_error("certFHE::ShardedBatch: key has no Context");
    return *key.certFHEContext;
}

ShardedBatch ShardedBatch::encryptWith(ShardGroup &group, const SecretKey &key, const std::vector<unsigned char> &bits,
                                       const void *rng_ptr, uint64_t first)
{
    const csgn_rng &rng = *static_cast<const csgn_rng *>(rng_ptr);
    const Context &ctx = keyContext(key);
    std::shared_ptr<ShardedData> out = std::make_shared<ShardedData>(group.impl, ctx, (uint64_t)bits.size(), 1);
    KeyMaterial km(key, ctx);
    ShardedData *o = out.get();
    group.impl->runAll([&](ShardWorker &me) {
        const uint64_t mine = o->mine(me.rank), lo = o->lo[me.rank];
        o->alloc(me);
        if (mine == 0)
            return;
        Staging st(me, km, (size_t)mine);
        ck(csgn_memcpy_h2d(st.extra(), bits.data() + lo, (size_t)mine, me.stream), "csgn_memcpy_h2d");
        // element i of the shard is element lo + i of the batch: it draws stream position first + lo + i
        ck(csgn_encrypt_keyed(km.n, km.d, mine, first + lo, st.extra(), st.key(), st.mask(), &rng, o->words(me.rank),
                              me.stream),
           "csgn_encrypt_keyed");
        ck(csgn_stream_sync(me.stream), "csgn_stream_sync");      // `bits` and the staging block die on return
    });
    return ShardedBatch(out);
}

ShardedBatch ShardedBatch::encrypt(ShardGroup &group, const SecretKey &key, const std::vector<unsigned char> &bits)
{
    csgn_rng rng;                       // ONE 256-bit generator key + nonce from the OS for the whole batch;
    ck(csgn_rng_from_os(&rng, 8), "csgn_rng_from_os");        // the GPUs draw disjoint position ranges of it
    ShardedBatch r = encryptWith(group, key, bits, &rng, 0);
    volatile uint32_t *wipe = rng.key;
    for (int i = 0; i < 8; ++i)
        wipe[i] = 0;
    return r;
}

ShardedBatch ShardedBatch::encrypt(ShardGroup &group, const SecretKey &key, const std::vector<unsigned char> &bits,
                                   uint64_t seed, uint64_t first_ciphertext)
{
    csgn_rng rng;
    ck(csgn_rng_from_seed(&rng, seed, 8), "csgn_rng_from_seed");
    return encryptWith(group, key, bits, &rng, first_ciphertext);
}

ShardedBatch ShardedBatch::encryptProduct(ShardGroup &group, const SecretKey &key,
                                          const std::vector<unsigned char> &bits_a,
                                          const std::vector<unsigned char> &bits_b, uint64_t seed_a, uint64_t seed_b,
                                          uint64_t first_ciphertext)
{
    if (bits_a.size() != bits_b.size())
        throw std::invalid_argument("certFHE::ShardedBatch::encryptProduct: one bit of each operand per element");
    const Context &ctx = keyContext(key);
    csgn_rng ra, rb;
    ck(csgn_rng_from_seed(&ra, seed_a, 8), "csgn_rng_from_seed");
    ck(csgn_rng_from_seed(&rb, seed_b, 8), "csgn_rng_from_seed");
    std::shared_ptr<ShardedData> out = std::make_shared<ShardedData>(group.impl, ctx, (uint64_t)bits_a.size(), 1);
    KeyMaterial km(key, ctx);
    ShardedData *o = out.get();
    group.impl->runAll([&](ShardWorker &me) {
        const uint64_t mine = o->mine(me.rank), lo = o->lo[me.rank];
        o->alloc(me);
        if (mine == 0)
            return;
        Staging st(me, km, (size_t)mine * 2);
        ck(csgn_memcpy_h2d(st.extra(), bits_a.data() + lo, (size_t)mine, me.stream), "csgn_memcpy_h2d");
        ck(csgn_memcpy_h2d(st.extra() + mine, bits_b.data() + lo, (size_t)mine, me.stream), "csgn_memcpy_h2d");
        ck(csgn_encrypt_mul_keyed(km.n, km.d, mine, first_ciphertext + lo, st.extra(), st.extra() + mine, st.key(),
                                  st.mask(), &ra, &rb, o->words(me.rank), nullptr, me.stream),
           "csgn_encrypt_mul_keyed");
        ck(csgn_stream_sync(me.stream), "csgn_stream_sync");
    });
    return ShardedBatch(out);
}

ShardedBatch ShardedBatch::synthetic(ShardGroup &group, const Context &context, uint64_t count, uint64_t terms,
                                     uint64_t seed)
{
    std::shared_ptr<ShardedData> out = std::make_shared<ShardedData>(group.impl, context, count, terms);
    ShardedData *o = out.get();
    group.impl->runAll([&](ShardWorker &me) {
        o->alloc(me);
        const uint64_t per = terms * o->ctx.getDefaultN();
        ck(csgn_synth_fill(seed, o->ctx.getN(), o->lo[me.rank] * per, o->mine(me.rank) * per, o->words(me.rank), me.stream),
           "csgn_synth_fill");
    });
    return ShardedBatch(out);
}

static void requireSame(const ShardedData &a, const ShardedData &b)
{
    if (a.g != b.g)
        throw std::invalid_argument("certFHE::ShardedBatch: operands live on different ShardGroups");
    if (a.ctx.getN() != b.ctx.getN() || a.count != b.count)
        throw std::invalid_argument("certFHE::ShardedBatch: operands differ in N or element count");
}

ShardedBatch ShardedBatch::operator*(const ShardedBatch &rhs) const
{
    requireSame(*data, *rhs.data);
    const ShardedData *a = data.get(), *b = rhs.data.get();
    std::shared_ptr<ShardedData> out = std::make_shared<ShardedData>(a->g, a->ctx, a->count, a->terms * b->terms);
    ShardedData *o = out.get();
    a->g->runAll([&](ShardWorker &me) {
        o->alloc(me);
        ck(csgn_mul_uniform(a->ctx.getN(), a->mine(me.rank), a->terms, b->terms, a->words(me.rank), b->words(me.rank),
                            o->words(me.rank), 0, me.stream),
           "csgn_mul_uniform");
    });
    return ShardedBatch(out);
}

ShardedBatch ShardedBatch::operator+(const ShardedBatch &rhs) const
{
    requireSame(*data, *rhs.data);
    const ShardedData *a = data.get(), *b = rhs.data.get();
    std::shared_ptr<ShardedData> out = std::make_shared<ShardedData>(a->g, a->ctx, a->count, a->terms + b->terms);
    ShardedData *o = out.get();
    a->g->runAll([&](ShardWorker &me) {
        o->alloc(me);
        ck(csgn_add_uniform(a->ctx.getN(), a->mine(me.rank), a->terms, b->terms, a->words(me.rank), b->words(me.rank),
                            o->words(me.rank), me.stream),
           "csgn_add_uniform");
    });
    return ShardedBatch(out);
}

ShardedBatch ShardedBatch::applyPermutation(const Permutation &permutation) const
{
    const ShardedData *a = data.get();
    const uint64_t n = a->ctx.getN();
    if (permutation.getLength() < n)
        throw std::invalid_argument("certFHE::ShardedBatch::applyPermutation: permutation shorter than N");
    std::vector<uint32_t> table(n);
    const uint64_t *src = permutation.getPermutation();
    for (uint64_t i = 0; i < n; ++i)
        table[i] = (uint32_t)src[i];
    std::shared_ptr<ShardedData> out = std::make_shared<ShardedData>(a->g, a->ctx, a->count, 1);
    ShardedData *o = out.get();
    a->g->runAll([&](ShardWorker &me) {
        o->alloc(me);
        const uint64_t mine = a->mine(me.rank);
        if (mine == 0)
            return;
        const size_t tb = ((size_t)n * 4 + 255) & ~(size_t)255;
        void *d_perm = me.take(tb);
        try {
            ck(csgn_memcpy_h2d(d_perm, table.data(), (size_t)n * 4, me.stream), "csgn_memcpy_h2d");
            // `table` is a pageable local that dies when runAll returns: the copy must have left it by then
            // (csgn_hip.h, csgn_memcpy_h2d: the host buffer belongs to the call until the stream has passed it)
            ck(csgn_stream_sync(me.stream), "csgn_stream_sync");
            ck(csgn_permute_uniform(n, mine, a->terms, 0, a->words(me.rank), static_cast<const uint32_t *>(d_perm),
                                    o->words(me.rank), me.stream),
               "csgn_permute_uniform");
        } catch (...) {
            me.give(d_perm, tb);
            throw;
        }
        me.give(d_perm, tb);              // stream-ordered: the next user of the block queues behind the kernel
    });
    return ShardedBatch(out);
}

std::vector<unsigned char> ShardedBatch::decryptWith(const ShardedBatch *rhs, bool product, const SecretKey &key) const
{
    const ShardedData *a = data.get();
    const ShardedData *b = rhs ? rhs->data.get() : nullptr;
    if (b)
        requireSame(*a, *b);
    std::vector<unsigned char> bits(a->count, 0);
    if (a->count == 0)
        return bits;
    KeyMaterial km(key, a->ctx);
    a->g->runAll([&](ShardWorker &me) {
        const uint64_t mine = a->mine(me.rank);
        Staging st(me, km, 0);
        const size_t need = b ? csgn_decrypt_combined_scratch_bytes(mine, a->terms, b->terms)
                              : csgn_decrypt_scratch_bytes(mine, mine * a->terms);
        const size_t scratch = (need + 255) & ~(size_t)255;
        const size_t local_b = ((size_t)mine + 255) & ~(size_t)255;
        const size_t total = scratch + local_b + (size_t)a->count;
        void *work = me.take(total);
        uint8_t *d_local = static_cast<uint8_t *>(work) + scratch, *d_all = d_local + local_b;
        try {
            if (mine && !b)
                ck(csgn_decrypt_uniform(a->ctx.getN(), mine, a->terms, a->words(me.rank), st.mask(), d_local, work, me.stream),
                   "csgn_decrypt_uniform");
            else if (mine && product)
                ck(csgn_decrypt_product_uniform(a->ctx.getN(), mine, a->terms, b->terms, a->words(me.rank),
                                                b->words(me.rank), st.mask(), d_local, work, me.stream),
                   "csgn_decrypt_product_uniform");
            else if (mine)
                ck(csgn_decrypt_sum_uniform(a->ctx.getN(), mine, a->terms, b->terms, a->words(me.rank), b->words(me.rank),
                                            st.mask(), d_local, work, me.stream),
                   "csgn_decrypt_sum_uniform");
            // the second exchange of SURVEY 8e: one byte per element
            ck(csgn_comm_gather_bytes(me.comm, d_local, a->count, d_all, me.stream), "csgn_comm_gather_bytes");
            ck(csgn_comm_barrier(me.comm, me.stream), "csgn_comm_barrier");
            if (me.rank == 0) {
                ck(csgn_memcpy_d2h(bits.data(), d_all, (size_t)a->count, me.stream), "csgn_memcpy_d2h");
                ck(csgn_stream_sync(me.stream), "csgn_stream_sync");
            }
        } catch (...) {
            me.give(work, total);
            throw;
        }
        me.give(work, total);
    });
    return bits;
}

std::vector<unsigned char> ShardedBatch::decrypt(const SecretKey &key) const { return decryptWith(nullptr, false, key); }

std::vector<unsigned char> ShardedBatch::decryptProduct(const ShardedBatch &rhs, const SecretKey &key) const
{
    return decryptWith(&rhs, true, key);
}

std::vector<unsigned char> ShardedBatch::decryptSum(const ShardedBatch &rhs, const SecretKey &key) const
{
    return decryptWith(&rhs, false, key);
}

std::vector<uint64_t> ShardedBatch::termCounts() const
{
    const ShardedData *a = data.get();
    std::vector<uint64_t> counts(a->count, 0);
    if (a->count == 0)
        return counts;
    std::vector<uint64_t> digests(a->g->w.size(), 0);
    a->g->runAll([&](ShardWorker &me) {
        const uint64_t mine = a->mine(me.rank);
        const size_t local_b = ((size_t)(mine ? mine : 1) * 8 + 255) & ~(size_t)255;
        const size_t total = local_b + (size_t)a->count * 8 + 256;
        void *work = me.take(total);
        uint64_t *d_local = static_cast<uint64_t *>(work);
        uint64_t *d_all = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(work) + local_b);
        uint64_t *d_dig = d_all + a->count;
        try {
            // every element of a uniform batch has `terms` terms (for a product: t1*t2, newlen/dL of
            // src/Ciphertext.cpp:146)
            ck(csgn_shard_product_counts(mine, nullptr, nullptr, a->terms, 1, d_local, me.stream), "csgn_shard_product_counts");
            ck(csgn_comm_gather_counts(me.comm, d_local, a->count, d_all, me.stream), "csgn_comm_gather_counts");
            ck(csgn_memset(d_dig, 0, 8, me.stream), "csgn_memset");
            ck(csgn_digest(d_all, a->count, 0, d_dig, me.stream), "csgn_digest");
            ck(csgn_comm_barrier(me.comm, me.stream), "csgn_comm_barrier");
            ck(csgn_memcpy_d2h(&digests[me.rank], d_dig, 8, me.stream), "csgn_memcpy_d2h");
            if (me.rank == 0)
                ck(csgn_memcpy_d2h(counts.data(), d_all, (size_t)a->count * 8, me.stream), "csgn_memcpy_d2h");
            ck(csgn_stream_sync(me.stream), "csgn_stream_sync");
        } catch (...) {
            me.give(work, total);
            throw;
        }
        me.give(work, total);
    });
    for (size_t r = 1; r < digests.size(); ++r)
        if (digests[r] != digests[0])
            throw std::runtime_error("certFHE::ShardedBatch::termCounts: rank " + std::to_string(r) +
                                     " received a different vector than rank 0");
    return counts;
}

uint64_t ShardedBatch::size() const { return data->count; }
uint64_t ShardedBatch::terms() const { return data->terms; }
int ShardedBatch::shards() const { return (int)data->g->w.size(); }
std::pair<uint64_t, uint64_t> ShardedBatch::shardRange(int rank) const
{
    return std::make_pair(data->lo.at(rank), data->hi.at(rank));
}
const Context &ShardedBatch::context() const { return data->ctx; }

std::vector<uint64_t> ShardedBatch::values(uint64_t i) const
{
    const ShardedData *a = data.get();
    if (i >= a->count)
        throw std::out_of_range("certFHE::ShardedBatch::values");
    const int owner = csgn_shard_owner(i, a->count, (int)a->g->w.size());
    const uint64_t per = a->terms * a->ctx.getDefaultN();
    std::vector<uint64_t> out(per, 0);
    a->g->runAll([&](ShardWorker &me) {
        if (me.rank != owner)
            return;
        ck(csgn_memcpy_d2h(out.data(), a->words(me.rank) + (i - a->lo[me.rank]) * per, (size_t)per * 8, me.stream),
           "csgn_memcpy_d2h");
        ck(csgn_stream_sync(me.stream), "csgn_stream_sync");
    });
    return out;
}

uint64_t ShardedBatch::digest() const
{
    const ShardedData *a = data.get();
    std::vector<uint64_t> part(a->g->w.size(), 0);
    a->g->runAll([&](ShardWorker &me) {
        const uint64_t per = a->terms * a->ctx.getDefaultN();
        void *d = me.take(256);
        try {
            ck(csgn_memset(d, 0, 8, me.stream), "csgn_memset");
            if (a->mine(me.rank))
                ck(csgn_digest(a->words(me.rank), a->mine(me.rank) * per, a->lo[me.rank] * per, static_cast<uint64_t *>(d),
                               me.stream),
                   "csgn_digest");
            ck(csgn_memcpy_d2h(&part[me.rank], d, 8, me.stream), "csgn_memcpy_d2h");
            ck(csgn_stream_sync(me.stream), "csgn_stream_sync");
        } catch (...) {
            me.give(d, 256);
            throw;
        }
        me.give(d, 256);
    });
    uint64_t sum = 0;
    for (size_t r = 0; r < part.size(); ++r)
        sum += part[r];
    return sum;
}

void ShardedBatch::synchronize() const
{
    data->g->runAll([](ShardWorker &me) { ck(csgn_stream_sync(me.stream), "csgn_stream_sync"); });
}

} // namespace certFHE
