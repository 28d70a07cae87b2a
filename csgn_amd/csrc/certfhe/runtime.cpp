// runtime.cpp -- device bring-up, error translation and HBM buffers for the certFHE:: classes.
#include "runtime.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace certFHE {
namespace detail {

namespace {
thread_local int g_device = -1;       // -1: not initialised on this thread
thread_local int g_requested = -1;    // Library::useDevice

// Block cache.  hipMalloc/hipFree cost ~10 us each and hipFree synchronises the device, which
// would dominate the value-semantic class API (every operator allocates a result, every
// temporary frees one).  Freed blocks are kept per power-of-two size class and handed out
// again; re-use is ordered by the single stream all class operations run on, so a block that
// an in-flight kernel still reads is only overwritten by work queued behind that kernel.
struct BlockCache {
    // Size classes: powers of two up to 1 MiB, four per octave above (1, 1.25, 1.5, 1.75 x 2^k), so a
    // 168 MB product holds 176 MB of HBM instead of 256 MB.
    static const int kClasses = 48 * 4;
    static const size_t kMaxCachedBlock = (size_t)256 << 20;   // larger blocks go straight back
    static const size_t kMaxCachedTotal = (size_t)4 << 30;
    std::vector<void *> free_list[kClasses];
    size_t cached_bytes;
    BlockCache() : cached_bytes(0) {}
    ~BlockCache() { release(); }
    // class of the smallest capacity >= bytes; *capacity receives that capacity
    static int size_class(size_t bytes, size_t *capacity)
    {
        int k = 8;                                  // 256-byte minimum
        while (((size_t)1 << k) < bytes)
            ++k;
        if (k <= 20 || bytes <= ((size_t)1 << (k - 1))) {
            *capacity = (size_t)1 << k;
            return 4 * k;
        }
        const size_t base = (size_t)1 << (k - 1), step = base >> 2;     // bytes in (base, 2*base]
        const size_t q = (bytes - base + step - 1) / step;              // 1..4 quarters above base
        *capacity = base + q * step;
        return q == 4 ? 4 * k : 4 * (k - 1) + (int)q;
    }
    void *take(size_t bytes, size_t *capacity)
    {
        const int c = size_class(bytes, capacity);
        if (c < kClasses && !free_list[c].empty()) {
            void *p = free_list[c].back();
            free_list[c].pop_back();
            cached_bytes -= *capacity;
            return p;
        }
        void *p = nullptr;
        int rc = csgn_malloc(&p, *capacity);
        if (rc != CSGN_OK && cached_bytes) {        // out of HBM: drop the cache and retry once
            release();
            rc = csgn_malloc(&p, *capacity);
        }
        check(rc, "csgn_malloc");
        return p;
    }
    void give(void *p, size_t capacity)
    {
        static const bool disabled = getenv("CSGN_NO_BLOCK_CACHE") != nullptr;   // A/B switch
        size_t same = 0;
        const int c = size_class(capacity, &same);   // capacities are class sizes: same == capacity
        if (!disabled && c < kClasses && capacity <= kMaxCachedBlock && cached_bytes + capacity <= kMaxCachedTotal) {
            free_list[c].push_back(p);
            cached_bytes += capacity;
        } else {
            csgn_free(p);
        }
    }
    void release()
    {
        for (int c = 0; c < kClasses; ++c) {
            for (size_t i = 0; i < free_list[c].size(); ++i)
                csgn_free(free_list[c][i]);
            free_list[c].clear();
        }
        cached_bytes = 0;
    }
};
// The cache lives on the heap behind a thread_local guard: objects with static storage can
// outlive the guard at process exit, and then simply free their block directly.
thread_local BlockCache *g_cache_ptr = nullptr;
struct CacheGuard {
    ~CacheGuard()
    {
        delete g_cache_ptr;
        g_cache_ptr = nullptr;
        dead = true;
    }
    bool dead = false;
};
thread_local CacheGuard g_cache_guard;

BlockCache *cache()
{
    if (!g_cache_ptr && !g_cache_guard.dead)
        g_cache_ptr = new BlockCache();
    return g_cache_ptr;
}
}

// One pinned, GPU-addressable 64-byte slot per host thread and device for single-value answers
// (SecretKey::decrypt): the kernel writes the bit straight into host memory, a stream
// synchronise makes it visible, and the device-to-host copy of one byte -- a third of the call's
// latency -- disappears.  Slots are never freed before process exit (one per thread and GPU).
namespace {
struct ResultSlot {
    void *host = nullptr;
    void *dev = nullptr;
};
thread_local ResultSlot g_slots[16];
}

volatile unsigned char *resultSlot(void **dev_alias)
{
    ensureDevice();
    ResultSlot local;
    ResultSlot &sl = (g_device >= 0 && g_device < 16) ? g_slots[g_device] : local;
    if (!sl.host)
        check(csgn_host_alloc(&sl.host, &sl.dev, 64), "csgn_host_alloc");
    *dev_alias = sl.dev;
    return static_cast<volatile unsigned char *>(sl.host);
}

void awaitByte(volatile unsigned char *slot, unsigned char armed)
{
    for (int spin = 0; spin < 200000; ++spin) {
        if (*slot != armed)
            return;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    syncDevice();
}

void check(int rc, const char *what)
{
    if (rc == CSGN_OK)
        return;
    std::string msg = std::string("certFHE (MI355X): ") + what + " failed [" + std::to_string(rc) +
                      "]: " + csgn_last_error();
    throw std::runtime_error(msg);
}

void selectDevice(int device)
{
    g_requested = device;
    g_device = -1;
    ensureDevice();
}

int activeDevice()
{
    ensureDevice();
    return g_device;
}

void ensureDevice()
{
    if (g_device >= 0)
        return;
    int dev = g_requested;
    if (dev < 0) {
        const char *env = getenv("CSGN_DEVICE");
        dev = (env && *env) ? atoi(env) : 0;
    }
    check(csgn_init(dev), "csgn_init");
    g_device = dev;
    // One round trip through every kind of call the classes make.  Measured on ROCm 7.2: the
    // first host-to-device copy of a process draws from libc's rand(); later calls do not.
    // Doing it here keeps a caller's `srand(seed); encrypt...` sequence bit-reproducible
    // (SecretKey's constructor and Library::initializeLibrary reach this point first).
    void *scratch = nullptr;
    uint64_t probe[4] = {1, 2, 3, 4};
    check(csgn_malloc(&scratch, sizeof(probe)), "csgn_malloc");
    check(csgn_memcpy_h2d(scratch, probe, sizeof(probe), stream()), "csgn_memcpy_h2d");
    check(csgn_synth_fill(0, 64, 0, 4, static_cast<uint64_t *>(scratch), stream()), "csgn_synth_fill");
    check(csgn_memcpy_d2h(probe, scratch, sizeof(probe), stream()), "csgn_memcpy_d2h");
    check(csgn_free(scratch), "csgn_free");
}

DevicePayload::~DevicePayload()
{
    if (!ptr || parent)                 // a view: the parent owns the bytes
        return;
    // only the cache of a thread bound to the SAME device may recycle the block
    BlockCache *c = (device == g_device) ? cache() : nullptr;
    if (c)
        c->give(ptr, capacity);
    else
        csgn_free(ptr);
}

std::shared_ptr<DevicePayload> allocBytes(size_t bytes)
{
    ensureDevice();
    std::shared_ptr<DevicePayload> p = std::make_shared<DevicePayload>();
    p->device = g_device;
    if (bytes) {
        BlockCache *c = cache();
        if (c) {
            p->ptr = c->take(bytes, &p->capacity);
        } else {                                   // the thread is ending and its cache is gone: straight from the driver
            p->capacity = bytes;
            check(csgn_malloc(&p->ptr, bytes), "csgn_malloc");
        }
    }
    p->words = bytes / 8;
    return p;
}

void releaseBlockCache()
{
    if (g_cache_ptr)
        g_cache_ptr->release();
}

std::shared_ptr<DevicePayload> allocWords(uint64_t words) { return allocBytes((size_t)words * 8); }

std::shared_ptr<DevicePayload> uploadWords(const uint64_t *host, uint64_t words)
{
    std::shared_ptr<DevicePayload> p = allocWords(words);
    if (words) {
        check(csgn_memcpy_h2d(p->ptr, host, (size_t)words * 8, stream()), "csgn_memcpy_h2d");
        check(csgn_stream_sync(stream()), "csgn_stream_sync");   // host buffer may die right after
    }
    return p;
}

void downloadBytes(void *host, const void *dev, size_t bytes)
{
    if (bytes)
        check(csgn_memcpy_d2h(host, dev, bytes, stream()), "csgn_memcpy_d2h");
}

// ---- pinned host blocks, process-wide (see runtime.h)
namespace {
struct PinnedPool {
    std::mutex lock;
    std::vector<void *> free_list[BlockCache::kClasses];
    size_t cached = 0;
};
// never destroyed: objects with static storage may drop their mirrors after any destructor of ours has run, and
// page-locked memory goes back to the system with the process
PinnedPool &pinnedPool()
{
    static PinnedPool *pool = new PinnedPool();
    return *pool;
}
} // namespace

void *pinnedTake(size_t bytes, size_t *capacity)
{
    ensureDevice();
    const int c = BlockCache::size_class(bytes, capacity);
    PinnedPool &pool = pinnedPool();
    if (c < BlockCache::kClasses) {
        std::lock_guard<std::mutex> hold(pool.lock);
        if (!pool.free_list[c].empty()) {
            void *p = pool.free_list[c].back();
            pool.free_list[c].pop_back();
            pool.cached -= *capacity;
            return p;
        }
    }
    void *host = nullptr, *alias = nullptr;
    int rc = csgn_host_alloc(&host, &alias, *capacity);
    if (rc != CSGN_OK && pool.cached) {            // the system refuses more locked pages: drop what is cached, retry once
        releasePinnedPool();
        rc = csgn_host_alloc(&host, &alias, *capacity);
    }
    check(rc, "csgn_host_alloc");
    return host;
}

void pinnedGive(void *host, size_t capacity)
{
    if (!host)
        return;
    size_t same = 0;
    const int c = BlockCache::size_class(capacity, &same);
    PinnedPool &pool = pinnedPool();
    {
        std::lock_guard<std::mutex> hold(pool.lock);
        if (c < BlockCache::kClasses && pool.cached + capacity <= kPinnedPoolBytes) {
            pool.free_list[c].push_back(host);
            pool.cached += capacity;
            return;
        }
    }
    (void)csgn_host_free(host);
}

void releasePinnedPool()
{
    PinnedPool &pool = pinnedPool();
    std::lock_guard<std::mutex> hold(pool.lock);
    for (int c = 0; c < BlockCache::kClasses; ++c) {
        for (size_t i = 0; i < pool.free_list[c].size(); ++i)
            (void)csgn_host_free(pool.free_list[c][i]);
        pool.free_list[c].clear();
    }
    pool.cached = 0;
}

// Staging buffers: two per host thread and device, made on first use and FREED when the thread ends (a server
// with thread churn would otherwise leak 16 MiB of locked pages per thread that ever serialised something).
namespace {
struct StagePair {
    void *host[2] = {nullptr, nullptr};
    void *dev[2] = {nullptr, nullptr};
};
struct StageSet {
    StagePair pair[17];                            // one per device, the last for device numbers past 15
    ~StageSet()
    {
        for (int d = 0; d < 17; ++d)
            for (int w = 0; w < 2; ++w)
                if (pair[d].host[w])
                    (void)csgn_host_free(pair[d].host[w]);
    }
};
thread_local StageSet g_stage;
} // namespace

void *stageBuffer(int which)
{
    ensureDevice();
    StagePair &sp = g_stage.pair[(g_device >= 0 && g_device < 16) ? g_device : 16];
    if (!sp.host[which])
        check(csgn_host_alloc(&sp.host[which], &sp.dev[which], kStageBytes), "csgn_host_alloc");
    return sp.host[which];
}

void downloadStaged(const void *dev, size_t bytes, void (*consume)(void *, const void *, size_t), void *ctx)
{
    if (bytes == 0)
        return;
    void *buf[2] = {stageBuffer(0), stageBuffer(1)};
    const char *src = static_cast<const char *>(dev);
    size_t done = 0, inflight = bytes < kStageBytes ? bytes : kStageBytes;
    check(csgn_memcpy_d2h(buf[0], src, inflight, stream()), "csgn_memcpy_d2h");
    for (int cur = 0; done < bytes; cur ^= 1) {
        check(csgn_stream_sync(stream()), "csgn_stream_sync");          // piece `cur` has landed
        const size_t have = inflight;
        const size_t next_at = done + have;
        inflight = 0;
        if (next_at < bytes) {                                          // start the next piece, then handle this one
            inflight = bytes - next_at < kStageBytes ? bytes - next_at : kStageBytes;
            check(csgn_memcpy_d2h(buf[cur ^ 1], src + next_at, inflight, stream()), "csgn_memcpy_d2h");
        }
        consume(ctx, buf[cur], have);
        done = next_at;
    }
}

void uploadStaged(void *dev, size_t bytes, void (*produce)(void *, void *, size_t), void *ctx)
{
    if (bytes == 0)
        return;
    void *buf[2] = {stageBuffer(0), stageBuffer(1)};
    char *dst = static_cast<char *>(dev);
    size_t done = 0;
    for (int cur = 0; done < bytes; cur ^= 1) {
        const size_t n = bytes - done < kStageBytes ? bytes - done : kStageBytes;
        produce(ctx, buf[cur], n);                                      // while the previous piece is on the link
        check(csgn_stream_sync(stream()), "csgn_stream_sync");          // the other buffer is free again after this
        check(csgn_memcpy_h2d(dst + done, buf[cur], n, stream()), "csgn_memcpy_h2d");
        done += n;
    }
    check(csgn_stream_sync(stream()), "csgn_stream_sync");
}

void syncDevice()
{
    if (g_device >= 0)
        check(csgn_stream_sync(stream()), "csgn_stream_sync");
}

// ------------------------------------------------------------------ deferred small operations (see runtime.h)
namespace {
std::atomic<bool> g_defer_on(getenv("CSGN_NO_DEFER") == nullptr);
}

struct DeferQueue {
    std::mutex lock;
    std::vector<std::shared_ptr<LazyNode> > pending;
    int device = -1;
    uint64_t n_bits = 0;
    // the records of a flush live in pinned, device-addressable memory: the kernel reads them over the link, no copy
    // is enqueued.  A ring of slots, each with an event recorded behind its launches: a slot is rewritten only after
    // that event has passed (it has, long since, unless the host runs eight flushes ahead of the GPU).
    static const int kSlots = 8;
    csgn_small_op *h_ops = nullptr, *d_ops = nullptr;
    void *slot_event[kSlots] = {nullptr};
    bool slot_used[kSlots] = {false};
    int slot = 0;

    ~DeferQueue()
    {
        for (int i = 0; i < kSlots; ++i)
            if (slot_event[i])
                (void)csgn_event_destroy(slot_event[i]);
        if (h_ops)
            (void)csgn_host_free(h_ops);
    }

    // evaluate everything that is pending.  Called with `lock` held, on ANY thread: the work goes to the queue's device.
    void flushLocked()
    {
        if (pending.empty())
            return;
        const int caller_device = g_device;
        const bool foreign = caller_device != device;
        if (foreign)
            check(csgn_init(device), "csgn_init (flush of another thread's queue)");
        try {
            evaluate(foreign);
        } catch (...) {
            if (foreign && caller_device >= 0)
                (void)csgn_init(caller_device);
            throw;
        }
        if (foreign && caller_device >= 0)
            check(csgn_init(caller_device), "csgn_init");
    }

    void evaluate(bool foreign)
    {
        const size_t n = pending.size();
        // One or two operations (a short sum in front of a large product: `big * (a + b)`): the ordinary kernels, one
        // launch each, in queue order -- the record path's fixed costs (the event, records read over the link: 8 us a
        // flush against 2.7 us a launch) only pay from a handful of operations on.
        if (n <= 2 && !foreign) {
            for (size_t i = 0; i < n; ++i) {
                LazyNode &nd = *pending[i];
                const uint64_t *A = nd.la ? nd.la->value->data() : nd.pa->data();
                const uint64_t *B = nd.lb ? nd.lb->value->data() : nd.pb->data();
                std::shared_ptr<DevicePayload> v = allocWords((nd.product ? (uint64_t)nd.t1 * nd.t2 : (uint64_t)nd.t1 + nd.t2) * nd.dl);
                if (nd.product)
                    check(csgn_mul_uniform(nd.n_bits, 1, nd.t1, nd.t2, A, B, v->data(), 0, stream()), "csgn_mul_uniform");
                else
                    check(csgn_add_uniform(nd.n_bits, 1, nd.t1, nd.t2, A, B, v->data(), stream()), "csgn_add_uniform");
                nd.value = v;
                nd.queue = nullptr;
            }
            for (size_t i = 0; i < n; ++i) {
                pending[i]->pa.reset();
                pending[i]->pb.reset();
                pending[i]->la.reset();
                pending[i]->lb.reset();
            }
            pending.clear();
            return;
        }
        if (!h_ops) {
            void *h = nullptr, *d = nullptr;
            check(csgn_host_alloc(&h, &d, sizeof(csgn_small_op) * kDeferBatch * kSlots), "csgn_host_alloc");
            h_ops = static_cast<csgn_small_op *>(h);
            d_ops = static_cast<csgn_small_op *>(d);
        }
        // where every result goes: side by side in one block, 16-byte aligned
        std::vector<uint64_t> at(n);
        uint64_t total = 0;
        int max_level = 0;
        for (size_t i = 0; i < n; ++i) {
            LazyNode &nd = *pending[i];
            nd.index = (int)i;
            nd.level = 0;
            if (nd.la)
                nd.level = std::max(nd.level, nd.la->level + 1);
            if (nd.lb)
                nd.level = std::max(nd.level, nd.lb->level + 1);
            max_level = std::max(max_level, nd.level);
            at[i] = total;
            const uint64_t words = (nd.product ? (uint64_t)nd.t1 * nd.t2 : (uint64_t)nd.t1 + nd.t2) * nd.dl;
            total += (words + 1) & ~1ull;
        }
        std::shared_ptr<DevicePayload> block;
        if (!foreign) {
            block = allocWords(total);
        } else {                                   // another thread's device: not through this thread's block cache
            block = std::make_shared<DevicePayload>();
            block->device = device;
            block->capacity = (size_t)total * 8;
            block->words = total;
            check(csgn_malloc(&block->ptr, block->capacity), "csgn_malloc");
        }
        uint64_t *base = block->data();
        const int cur = slot;
        slot = (slot + 1) % kSlots;
        if (!slot_event[cur])
            check(csgn_event_create(&slot_event[cur]), "csgn_event_create");
        if (slot_used[cur])
            check(csgn_event_sync(slot_event[cur]), "csgn_event_sync");     // the launch that read this slot has run
        csgn_small_op *h = h_ops + (size_t)cur * kDeferBatch, *d = d_ops + (size_t)cur * kDeferBatch;
        // records level by level: a level's operations only read payloads and results of earlier levels
        std::vector<size_t> start((size_t)max_level + 2, 0);
        for (size_t i = 0; i < n; ++i)
            start[(size_t)pending[i]->level + 1] += 1;
        for (size_t l = 1; l < start.size(); ++l)
            start[l] += start[l - 1];
        std::vector<size_t> fill(start.begin(), start.end() - 1);
        for (size_t i = 0; i < n; ++i) {
            const LazyNode &nd = *pending[i];
            csgn_small_op &op = h[fill[(size_t)nd.level]++];
            op.left = nd.la ? base + at[(size_t)nd.la->index] : nd.pa->data();
            op.right = nd.lb ? base + at[(size_t)nd.lb->index] : nd.pb->data();
            op.out = base + at[i];
            op.t1 = nd.t1;
            op.t2 = nd.t2;
            op.kind = nd.product ? 1u : 0u;
            op.reserved = 0;
        }
        for (int l = 0; l <= max_level; ++l) {
            const size_t cnt = start[(size_t)l + 1] - start[(size_t)l];
            if (cnt)
                check(csgn_small_ops(n_bits, cnt, d + start[(size_t)l], stream()), "csgn_small_ops");
        }
        check(csgn_event_record(slot_event[cur], stream()), "csgn_event_record");
        slot_used[cur] = true;
        for (size_t i = 0; i < n; ++i) {
            LazyNode &nd = *pending[i];
            std::shared_ptr<DevicePayload> v = std::make_shared<DevicePayload>();
            v->ptr = base + at[i];
            v->words = (nd.product ? (uint64_t)nd.t1 * nd.t2 : (uint64_t)nd.t1 + nd.t2) * nd.dl;
            v->capacity = (size_t)v->words * 8;
            v->device = device;
            v->parent = block;
            nd.value = v;
            nd.pa.reset();
            nd.pb.reset();
            nd.queue = nullptr;
        }
        for (size_t i = 0; i < n; ++i) {              // (after every node has its value: a node may be another's operand)
            pending[i]->la.reset();
            pending[i]->lb.reset();
        }
        pending.clear();
    }
};

namespace {
thread_local DeferQueue *g_queue_ptr = nullptr;
struct QueueGuard {
    bool dead = false;
    ~QueueGuard()
    {
        if (g_queue_ptr) {
            try {
                std::lock_guard<std::mutex> hold(g_queue_ptr->lock);
                g_queue_ptr->flushLocked();          // results somebody still points at must exist
            } catch (...) {
            }
            delete g_queue_ptr;
            g_queue_ptr = nullptr;
        }
        dead = true;
    }
};
thread_local QueueGuard g_queue_guard;

DeferQueue *queue()
{
    if (!g_queue_ptr && !g_queue_guard.dead) {
        ensureDevice();
        g_queue_ptr = new DeferQueue();
        g_queue_ptr->device = g_device;
    }
    return g_queue_ptr;
}
} // namespace

void setDeferral(bool on)
{
    if (!on)
        flushDeferred();
    g_defer_on.store(on);
}
bool deferralOn() { return g_defer_on.load(); }

void flushDeferred()
{
    if (!g_queue_ptr)
        return;
    std::lock_guard<std::mutex> hold(g_queue_ptr->lock);
    g_queue_ptr->flushLocked();
}

std::shared_ptr<DevicePayload> valueOf(const std::shared_ptr<LazyNode> &node)
{
    if (!node)
        return std::shared_ptr<DevicePayload>();
    for (;;) {
        DeferQueue *q = node->queue;                 // (read without the lock: nullptr only ever replaces a queue)
        if (!q)
            return node->value;
        std::lock_guard<std::mutex> hold(q->lock);
        if (node->queue == q) {                      // still pending there
            q->flushLocked();
            return node->value;
        }
    }
}

std::shared_ptr<LazyNode> deferSmallOp(bool product, uint64_t n_bits, uint64_t dl, uint64_t t1, uint64_t t2,
                                       const std::shared_ptr<DevicePayload> &pa, const std::shared_ptr<LazyNode> &la,
                                       const std::shared_ptr<DevicePayload> &pb, const std::shared_ptr<LazyNode> &lb)
{
    if (!g_defer_on.load() || t1 == 0 || t2 == 0 || t1 > kDeferMaxTerms || t2 > kDeferMaxTerms || dl == 0)
        return std::shared_ptr<LazyNode>();
    DeferQueue *q = queue();
    if (!q || q->device != g_device)                 // thread shutting down, or it moved to another GPU meanwhile
        return std::shared_ptr<LazyNode>();
    // operands pending in ANOTHER thread's queue (or already evaluated) are taken as finished payloads, so that a
    // queue only ever refers to its own nodes
    std::shared_ptr<DevicePayload> fa = pa, fb = pb;
    std::shared_ptr<LazyNode> na = la, nb = lb;
    if (na && na->queue != q) {
        fa = valueOf(na);
        na.reset();
    }
    if (nb && nb->queue != q) {
        fb = valueOf(nb);
        nb.reset();
    }
    if ((!fa && !na) || (!fb && !nb))
        return std::shared_ptr<LazyNode>();
    if ((fa && fa->device != q->device) || (fb && fb->device != q->device))
        return std::shared_ptr<LazyNode>();          // operands of another GPU: the immediate path reports that
    std::lock_guard<std::mutex> hold(q->lock);
    if (!q->pending.empty() && q->n_bits != n_bits)
        q->flushLocked();                            // one launch serves one term size
    if (na && !na->queue) {                          // (evaluated by the flush just above)
        fa = na->value;
        na.reset();
    }
    if (nb && !nb->queue) {
        fb = nb->value;
        nb.reset();
    }
    q->n_bits = n_bits;
    std::shared_ptr<LazyNode> nd = std::make_shared<LazyNode>();
    nd->pa = fa;
    nd->pb = fb;
    nd->la = na;
    nd->lb = nb;
    nd->n_bits = n_bits;
    nd->dl = dl;
    nd->t1 = (uint32_t)t1;
    nd->t2 = (uint32_t)t2;
    nd->product = product;
    nd->queue = q;
    nd->index = nd->level = 0;
    q->pending.push_back(nd);
    if (q->pending.size() >= kDeferBatch)
        q->flushLocked();
    return nd;
}

} // namespace detail
} // namespace certFHE
