// runtime.cpp -- device bring-up, error translation and HBM buffers for the certFHE:: classes.
#include "runtime.h"

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
    if (!ptr)
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
    if (bytes)
        p->ptr = cache()->take(bytes, &p->capacity);
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

} // namespace detail
} // namespace certFHE
