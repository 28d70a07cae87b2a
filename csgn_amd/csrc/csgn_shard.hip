// csgn_shard.hip -- libcsgn_shard.so (include/csgn_shard.h): contiguous batch partition across the
// GPUs of a node and the RCCL-over-xGMI all-gather of per-pair result term counts.  Links librccl
// directly; no torch, no MPI.  One communicator per GPU, driven by one host thread (or process)
// each.  Nothing here touches ciphertext words: products stay on the GPU that computed them.
#include "csgn_shard.h"
#include "csgn_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

static_assert(CSGN_COMM_ID_BYTES == sizeof(ncclUniqueId), "id buffer must hold an ncclUniqueId");

// Use of `nccl` against csgn_comm_abort from another thread: ncclCommAbort FREES the
// communicator, so nobody may be on its way into an RCCL call with the old handle when it runs.
//   * every RCCL call sits between comm_enter() (refuses once aborted; counts the caller in) and
//     comm_leave();
//   * csgn_comm_abort sets the flag and, when nobody is inside, aborts at once; otherwise the LAST caller
//     to leave does it.  A caller that is stuck inside RCCL (a peer that never connects) is waited for
//     for kAbortGraceMs and then aborted under its feet -- what ncclCommAbort is for -- which is the one
//     case left where the handle is freed while in use.
struct csgn_comm {
    ncclComm_t nccl = nullptr;      // guarded by `m` (nullptr once aborted)
    std::mutex m;
    std::condition_variable cv;
    int inside = 0;                 // threads between comm_enter and comm_leave
    hipStream_t stream = nullptr;
    int rank = 0, world = 1, device = 0;
    int *d_flag = nullptr;          // 1-element buffer of the barrier all-reduce
    hipEvent_t done = nullptr;      // marks the end of the barrier's all-reduce (bounded wait)
    std::atomic<bool> aborted{false};
    std::atomic<uint64_t> timeout_ms{120000};
    std::atomic<int> force_grouped{0};
};

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(e_ == hipErrorNoDevice ? CSGN_ERR_NO_DEVICE : CSGN_ERR_HIP,    \
                        "%s: %s", #expr, hipGetErrorString(e_));                       \
    } while (0)

#define NCCL_TRY(expr)                                                                 \
    do {                                                                               \
        ncclResult_t r_ = (expr);                                                      \
        if (r_ != ncclSuccess)                                                         \
            return fail(CSGN_ERR_HIP, "%s: %s", #expr, ncclGetErrorString(r_));        \
    } while (0)

#define REQUIRE(cond, ...)                               \
    do {                                                 \
        if (!(cond))                                     \
            return fail(CSGN_ERR_INVALID, __VA_ARGS__);  \
    } while (0)

inline uint64_t ceil_mul_div(uint64_t a, uint64_t b, uint64_t d)
{
    // ceil(a*b/d) with a 128-bit product (a = rank <= world = d, b = total pairs)
    const unsigned __int128 p = (unsigned __int128)a * b;
    return (uint64_t)((p + d - 1) / d);
}

// Puts the calling thread's current device back when it goes out of scope: the calls that walk over
// several GPUs (csgn_comm_init_all*, csgn_comm_destroy) must not leave their caller on another device.
struct DeviceGuard {
    int saved = -1;
    DeviceGuard()
    {
        if (hipGetDevice(&saved) != hipSuccess) {
            saved = -1;
            (void)hipGetLastError();
        }
    }
    ~DeviceGuard()
    {
        if (saved >= 0)
            (void)hipSetDevice(saved);
    }
};

constexpr int kAbortGraceMs = 2000;

// -> the handle to use, or nullptr when the communicator is (being) aborted
ncclComm_t comm_enter(csgn_comm *c)
{
    std::lock_guard<std::mutex> g(c->m);
    if (c->aborted.load() || !c->nccl)
        return nullptr;
    ++c->inside;
    return c->nccl;
}

void comm_leave(csgn_comm *c)
{
    ncclComm_t victim = nullptr;
    {
        std::lock_guard<std::mutex> g(c->m);
        if (--c->inside == 0 && c->aborted.load()) {
            victim = c->nccl;               // the abort that was asked for while we were inside
            c->nccl = nullptr;
        }
    }
    if (victim)
        (void)ncclCommAbort(victim);
    c->cv.notify_all();
}

struct CommUse {
    csgn_comm *c;
    ncclComm_t nccl;
    explicit CommUse(csgn_comm *comm) : c(comm), nccl(comm_enter(comm)) {}
    ~CommUse()
    {
        if (nccl)
            comm_leave(c);
    }
    CommUse(const CommUse &) = delete;
    CommUse &operator=(const CommUse &) = delete;
};

int finish_comm(csgn_comm *c)
{
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void **)&c->d_flag, sizeof(int)));
    HIP_TRY(hipMemset(c->d_flag, 0, sizeof(int)));
    HIP_TRY(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    return CSGN_OK;
}

// `stream` of the C ABI -> hipStream_t: NULL is the legacy default stream, as everywhere in
// csgn_hip.h; the communicator's own stream has to be asked for by name.
inline hipStream_t pick_stream(const csgn_comm *c, void *stream)
{
    return stream == CSGN_STREAM_OF_COMM ? c->stream : reinterpret_cast<hipStream_t>(stream);
}

// One RCCL per process, and the one this file was compiled for: compare the library the dynamic
// loader really bound with the header's version.
int check_rccl(unsigned flags)
{
    int runtime = 0;
    NCCL_TRY(ncclGetVersion(&runtime));
    const int header = NCCL_VERSION_CODE;
    const int rmaj = runtime / 10000, rmin = (runtime / 100) % 100;
    const int hmaj = header / 10000, hmin = (header / 100) % 100;
    Dl_info info;
    const char *path = (dladdr(reinterpret_cast<void *>(&ncclGetVersion), &info) && info.dli_fname) ? info.dli_fname : "?";
    if (rmaj != hmaj)
        return fail(CSGN_ERR_UNSUPPORTED, "RCCL major version mismatch: runtime %d.%d.%d (%s) vs header %d.%d.%d",
                    rmaj, rmin, runtime % 100, path, hmaj, hmin, header % 100);
    if (rmin != hmin && !(flags & CSGN_COMM_ALLOW_MINOR_SKEW))
        return fail(CSGN_ERR_UNSUPPORTED,
                    "RCCL minor version mismatch: runtime %d.%d.%d (%s) vs header %d.%d.%d; the process bound a "
                    "different librccl than libcsgn_shard.so was built for (pass CSGN_COMM_ALLOW_MINOR_SKEW to "
                    "accept the one already mapped)",
                    rmaj, rmin, runtime % 100, path, hmaj, hmin, header % 100);
    return CSGN_OK;
}

__global__ void __launch_bounds__(256) k_product_counts(uint64_t batch, const uint64_t *__restrict__ offL,
                                                        const uint64_t *__restrict__ offR, uint64_t uniform,
                                                        uint64_t *__restrict__ counts)
{
    const uint64_t b = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (b < batch)
        counts[b] = offL ? (offL[b + 1] - offL[b]) * (offR[b + 1] - offR[b]) : uniform;
}

int gather_plan(uint64_t total, int world, uint64_t *lo, uint64_t *len, int *equal)
{
    REQUIRE(world > 0 && lo && len, "bad gather plan arguments");
    bool eq = true;
    for (int r = 0; r < world; ++r) {
        uint64_t l = 0, h = 0;
        csgn_shard_range(total, r, world, &l, &h);
        lo[r] = l;
        len[r] = h - l;
        eq = eq && len[r] == len[0];
    }
    if (equal)
        *equal = eq ? 1 : 0;
    return CSGN_OK;
}

template <typename T>
int gather(csgn_comm *c, const T *d_local, uint64_t total, T *d_all, ncclDataType_t dt, void *stream)
{
    REQUIRE(c, "null communicator");
    CommUse use(c);
    REQUIRE(use.nccl, "communicator was aborted");
    if (total == 0)
        return CSGN_OK;
    REQUIRE(d_all, "d_all is null");
    hipStream_t s = pick_stream(c, stream);
    std::vector<uint64_t> lo(c->world), len(c->world);
    int equal = 0;
    if (int rc = gather_plan(total, c->world, lo.data(), len.data(), &equal))
        return rc;
    REQUIRE(d_local || len[c->rank] == 0, "d_local is null");
    if (equal && !c->force_grouped.load()) {
        // equal shards: the slices of d_all are exactly the all-gather layout
        NCCL_TRY(ncclAllGather(d_local, d_all, (size_t)len[c->rank], dt, use.nccl, s));
        return CSGN_OK;
    }
    // uneven shards: rank r broadcasts its len[r] elements into d_all[lo[r] ..); one group.  A rank's own
    // slice is sent from d_local (out of place); the others receive in place.
    NCCL_TRY(ncclGroupStart());
    for (int r = 0; r < c->world; ++r) {
        if (len[r] == 0)
            continue;
        const ncclResult_t res = ncclBroadcast(r == c->rank ? (const void *)d_local : (const void *)(d_all + lo[r]),
                                               d_all + lo[r], (size_t)len[r], dt, r, use.nccl, s);
        if (res != ncclSuccess) {
            (void)ncclGroupEnd();
            return fail(CSGN_ERR_HIP, "ncclBroadcast: %s", ncclGetErrorString(res));
        }
    }
    NCCL_TRY(ncclGroupEnd());
    return CSGN_OK;
}

} // namespace

extern "C" {

const char *csgn_shard_last_error(void) { return g_err; }

int csgn_shard_range(uint64_t total_pairs, int rank, int world, uint64_t *lo, uint64_t *hi)
{
    REQUIRE(lo && hi, "null output");
    *lo = *hi = 0;
    REQUIRE(world > 0 && rank >= 0 && rank < world, "bad rank %d / world %d", rank, world);
    const uint64_t l = ceil_mul_div((uint64_t)rank, total_pairs, (uint64_t)world);
    uint64_t h = ceil_mul_div((uint64_t)rank + 1, total_pairs, (uint64_t)world);
    if (h > total_pairs)
        h = total_pairs;
    *lo = l;
    *hi = h;
    return CSGN_OK;
}

int csgn_shard_owner(uint64_t pair, uint64_t total_pairs, int world)
{
    if (world <= 0 || pair >= total_pairs)
        return -1;
    // the rank r with ceil(r*B/G) <= p < ceil((r+1)*B/G)  <=>  r = floor(p*G/B)
    return (int)(((unsigned __int128)pair * (uint64_t)world) / total_pairs);
}

int csgn_comm_device_count(int *h_count)
{
    REQUIRE(h_count, "h_count is null");
    *h_count = 0;
    HIP_TRY(hipGetDeviceCount(h_count));
    return CSGN_OK;
}

int csgn_comm_rccl_info(int *h_runtime, int *h_header, char *h_path, size_t cap)
{
    int runtime = 0;
    NCCL_TRY(ncclGetVersion(&runtime));
    if (h_runtime)
        *h_runtime = runtime;
    if (h_header)
        *h_header = NCCL_VERSION_CODE;
    if (h_path && cap) {
        Dl_info info;
        const char *path = (dladdr(reinterpret_cast<void *>(&ncclGetVersion), &info) && info.dli_fname) ? info.dli_fname : "";
        snprintf(h_path, cap, "%s", path);
    }
    return CSGN_OK;
}

int csgn_shard_gather_plan(uint64_t total_pairs, int world, uint64_t *h_lo, uint64_t *h_len, int *h_equal)
{
    return gather_plan(total_pairs, world, h_lo, h_len, h_equal);
}

int csgn_comm_init_all(int ndev, const int *devices, csgn_comm **comms)
{
    return csgn_comm_init_all_ex(ndev, devices, CSGN_COMM_STRICT, comms);
}

int csgn_comm_init_all_ex(int ndev, const int *devices, unsigned flags, csgn_comm **comms)
{
    REQUIRE(comms && ndev > 0, "bad arguments");
    if (int rc = check_rccl(flags))
        return rc;
    int have = 0;
    HIP_TRY(hipGetDeviceCount(&have));
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; ++i) {
        devs[i] = devices ? devices[i] : i;
        REQUIRE(devs[i] >= 0 && devs[i] < have, "device %d not visible (have %d)", devs[i], have);
        comms[i] = nullptr;
    }
    DeviceGuard guard;
    std::vector<ncclComm_t> nc(ndev, nullptr);
    NCCL_TRY(ncclCommInitAll(nc.data(), ndev, devs.data()));
    for (int i = 0; i < ndev; ++i) {                 // every RCCL communicator gets its owner first, so that a
        csgn_comm *c = new csgn_comm();              // failure below leaves nothing the caller cannot destroy
        c->nccl = nc[i];
        c->rank = i;
        c->world = ndev;
        c->device = devs[i];
        comms[i] = c;
    }
    for (int i = 0; i < ndev; ++i)
        if (int rc = finish_comm(comms[i]))
            return rc;                               // the caller destroys comms[0..ndev)
    return CSGN_OK;
}

int csgn_comm_unique_id(unsigned char h_id[CSGN_COMM_ID_BYTES])
{
    REQUIRE(h_id, "h_id is null");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(h_id, &id, sizeof(id));
    return CSGN_OK;
}

int csgn_comm_init_rank(const unsigned char h_id[CSGN_COMM_ID_BYTES], int rank, int world, int device,
                        csgn_comm **comm)
{
    return csgn_comm_init_rank_ex(h_id, rank, world, device, CSGN_COMM_STRICT, comm);
}

int csgn_comm_init_rank_ex(const unsigned char h_id[CSGN_COMM_ID_BYTES], int rank, int world, int device,
                           unsigned flags, csgn_comm **comm)
{
    REQUIRE(h_id && comm, "null argument");
    *comm = nullptr;
    if (int rc = check_rccl(flags))
        return rc;
    REQUIRE(world > 0 && rank >= 0 && rank < world, "bad rank %d / world %d", rank, world);
    int have = 0;
    HIP_TRY(hipGetDeviceCount(&have));
    REQUIRE(device >= 0 && device < have, "device %d not visible (have %d)", device, have);
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, h_id, sizeof(id));
    ncclComm_t nc = nullptr;
    NCCL_TRY(ncclCommInitRank(&nc, world, id, rank));
    csgn_comm *c = new csgn_comm();
    c->nccl = nc;
    c->rank = rank;
    c->world = world;
    c->device = device;
    if (int rc = finish_comm(c)) {
        csgn_comm_destroy(c);
        return rc;
    }
    *comm = c;
    return CSGN_OK;
}

int csgn_comm_destroy(csgn_comm *c)
{
    if (!c)
        return CSGN_OK;
    DeviceGuard guard;
    (void)hipSetDevice(c->device);
    const bool aborted = c->aborted.load();
    if (c->stream && !aborted)
        (void)hipStreamSynchronize(c->stream);
    if (c->nccl && !aborted)
        (void)ncclCommDestroy(c->nccl);         // an aborted communicator was already freed by ncclCommAbort
    if (c->d_flag)
        (void)hipFree(c->d_flag);
    if (c->done)
        (void)hipEventDestroy(c->done);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
    return CSGN_OK;
}

int csgn_comm_abort(csgn_comm *c)
{
    REQUIRE(c, "null communicator");
    ncclComm_t victim = nullptr;
    {
        std::unique_lock<std::mutex> g(c->m);
        if (c->aborted.exchange(true))
            return CSGN_OK;                      // once: ncclCommAbort frees the communicator
        // somebody inside an RCCL call: the last one out aborts (comm_leave); give them a moment
        c->cv.wait_for(g, std::chrono::milliseconds(kAbortGraceMs), [c] { return c->inside == 0; });
        victim = c->nccl;                        // still set: nobody was inside, or they are stuck in there
        c->nccl = nullptr;
    }
    if (victim)
        NCCL_TRY(ncclCommAbort(victim));
    return CSGN_OK;
}

int csgn_comm_check(csgn_comm *c)
{
    REQUIRE(c, "null communicator");
    CommUse use(c);
    if (!use.nccl)
        return fail(CSGN_ERR_HIP, "communicator was aborted");
    ncclResult_t async = ncclSuccess;
    NCCL_TRY(ncclCommGetAsyncError(use.nccl, &async));
    if (async != ncclSuccess && async != ncclInProgress)
        return fail(CSGN_ERR_HIP, "RCCL asynchronous error: %s", ncclGetErrorString(async));
    return CSGN_OK;
}

int csgn_comm_set_timeout_ms(csgn_comm *c, uint64_t timeout_ms)
{
    REQUIRE(c, "null communicator");
    c->timeout_ms.store(timeout_ms);
    return CSGN_OK;
}

int csgn_comm_set_option(csgn_comm *c, int option, int value)
{
    REQUIRE(c, "null communicator");
    REQUIRE(option == CSGN_COMM_OPT_FORCE_GROUPED_BROADCAST, "unknown option %d", option);
    c->force_grouped.store(value);
    return CSGN_OK;
}

int csgn_comm_rank(const csgn_comm *c) { return c ? c->rank : -1; }
int csgn_comm_world(const csgn_comm *c) { return c ? c->world : 0; }
int csgn_comm_device(const csgn_comm *c) { return c ? c->device : -1; }
void *csgn_comm_stream(const csgn_comm *c) { return c ? c->stream : nullptr; }

int csgn_comm_gather_counts(csgn_comm *c, const uint64_t *d_local, uint64_t total_pairs, uint64_t *d_all,
                            void *stream)
{
    return gather<uint64_t>(c, d_local, total_pairs, d_all, ncclUint64, stream);
}

int csgn_comm_gather_bytes(csgn_comm *c, const uint8_t *d_local, uint64_t total_pairs, uint8_t *d_all,
                           void *stream)
{
    return gather<uint8_t>(c, d_local, total_pairs, d_all, ncclUint8, stream);
}

int csgn_comm_barrier(csgn_comm *c, void *stream)
{
    REQUIRE(c, "null communicator");
    hipStream_t s = pick_stream(c, stream);
    {
        CommUse use(c);
        REQUIRE(use.nccl, "communicator was aborted");
        NCCL_TRY(ncclAllReduce(c->d_flag, c->d_flag, 1, ncclInt32, ncclSum, use.nccl, s));
    }
    HIP_TRY(hipEventRecord(c->done, s));
    // bounded wait: a peer that never arrives must not hold this rank for ever
    const uint64_t limit = c->timeout_ms.load();
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;) {
        const hipError_t q = hipEventQuery(c->done);
        if (q == hipSuccess)
            break;
        if (q != hipErrorNotReady)
            return fail(CSGN_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(q));
        if (c->aborted.load())
            return fail(CSGN_ERR_HIP, "communicator was aborted while waiting in the barrier");
        if (++spins > 2000) {                    // past the first ~ms: look at the clock and at RCCL, then sleep
            ncclResult_t async = ncclSuccess;
            bool broken = false;
            {
                CommUse use(c);                  // (not across the sleep: an abort must not wait for a poller)
                if (!use.nccl)
                    return fail(CSGN_ERR_HIP, "communicator was aborted while waiting in the barrier");
                broken = ncclCommGetAsyncError(use.nccl, &async) == ncclSuccess && async != ncclSuccess &&
                         async != ncclInProgress;
            }
            if (broken) {
                (void)csgn_comm_abort(c);
                return fail(CSGN_ERR_HIP, "RCCL asynchronous error in the barrier: %s", ncclGetErrorString(async));
            }
            const uint64_t ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(
                                    std::chrono::steady_clock::now() - t0).count();
            if (limit && ms > limit) {
                (void)csgn_comm_abort(c);
                return fail(CSGN_ERR_TIMEOUT, "barrier: rank %d of %d waited %llu ms for its peers; communicator aborted",
                            c->rank, c->world, (unsigned long long)ms);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    return CSGN_OK;
}

int csgn_shard_product_counts(uint64_t batch, const uint64_t *d_off_left, const uint64_t *d_off_right,
                              uint64_t t1, uint64_t t2, uint64_t *d_counts, void *stream)
{
    if (batch == 0)
        return CSGN_OK;
    REQUIRE(d_counts, "d_counts is null");
    REQUIRE((d_off_left == nullptr) == (d_off_right == nullptr), "pass both offset arrays or neither");
    REQUIRE(batch < (1ull << 40), "batch too large");
    const uint64_t blocks = (batch + 255) / 256;
    REQUIRE(blocks < (1ull << 24), "batch too large for one launch");
    REQUIRE(stream != CSGN_STREAM_OF_COMM, "csgn_shard_product_counts takes a real stream (no communicator here)");
    k_product_counts<<<(unsigned)blocks, 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(
        batch, d_off_left, d_off_right, t1 * t2, d_counts);
    HIP_TRY(hipGetLastError());
    return CSGN_OK;
}

} // extern "C"
