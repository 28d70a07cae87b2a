// csgn_capi.hip -- the extern "C" surface of libcsgn_hip.so (declared in include/csgn_hip.h).
// Argument validation, error reporting and stream plumbing only; the kernels live in
// csgn_{mul,add,decrypt,encrypt,permute,compact,harness}.hip.  There is deliberately no CPU fallback anywhere in this library.
#include "csgn_capi_util.h"
#include "csgn_tuning.h"

#include <sys/random.h>

#include <cerrno>
#include <cstdarg>
#include <vector>
#include <cstdio>
#include <cstring>

namespace csgn {
namespace capi {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
        return fail(CSGN_ERR_NO_DEVICE, "%s: %s (no usable HIP device; this library has no CPU path)",
                    what, hipGetErrorString(e));
    return fail(CSGN_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

// a*b*c < limit, evaluated without wrapping (operands may be anything up to 2^64-1)
bool product_below(uint64_t a, uint64_t b, uint64_t c, uint64_t limit)
{
    unsigned long long ab, abc;
    if (__builtin_mul_overflow((unsigned long long)a, (unsigned long long)b, &ab) ||
        __builtin_mul_overflow(ab, (unsigned long long)c, &abc))
        return false;
    return abc < limit;
}

// ChaCha20 block on the host (D. J. Bernstein's function, 64-bit counter / 64-bit nonce layout) with
// caller-chosen constant words: only used to derive a circuit encrypt node's key.
void host_chacha20(const uint32_t sigma[4], const uint32_t key[8], uint64_t nonce, uint64_t counter, uint32_t out[16])
{
    uint32_t in[16], x[16];
    for (int i = 0; i < 4; ++i)
        in[i] = sigma[i];
    for (int i = 0; i < 8; ++i)
        in[4 + i] = key[i];
    in[12] = (uint32_t)counter;
    in[13] = (uint32_t)(counter >> 32);
    in[14] = (uint32_t)nonce;
    in[15] = (uint32_t)(nonce >> 32);
    memcpy(x, in, sizeof(x));
    auto rotl = [](uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
    auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 16);
        x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 12);
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 8);
        x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 7);
    };
    for (int r = 0; r < 20; r += 2) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i)
        out[i] = x[i] + in[i];
}

// key of a circuit's encrypt node = words 0..7 of ChaCha20(constants "csgn node key v1", key, nonce, counter 0)
void node_key_from(const csgn_rng &rng, uint32_t node_key[8])
{
    static const uint32_t sigma[4] = {0x6e677363u, 0x646f6e20u, 0x656b2065u, 0x31762079u};   // "csgn node key v1"
    uint32_t block[16];
    host_chacha20(sigma, rng.key, rng.nonce, 0, block);
    for (int i = 0; i < 8; ++i)
        node_key[i] = block[i];
    volatile uint32_t *wipe = block;
    for (int i = 0; i < 16; ++i)
        wipe[i] = 0;
}

// Shape limits shared by every compute entry point.
int check_n(uint64_t n_bits)
{
    if (n_bits == 0)
        return fail(CSGN_ERR_INVALID, "n_bits must be > 0");
    if (((n_bits + 63) / 64) * 8 > 16384)
        return fail(CSGN_ERR_UNSUPPORTED, "n_bits=%llu: terms above 16384 bytes are not supported",
                    (unsigned long long)n_bits);
    return CSGN_OK;
}

} // namespace capi
} // namespace csgn

using namespace csgn::capi;

extern "C" {

int csgn_abi_version(void) { return CSGN_ABI_VERSION; }

const char *csgn_last_error(void) { return g_err; }

int csgn_device_count(int *h_count)
{
    REQUIRE(h_count, "h_count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *h_count = 0;
        return hip_fail(e, "hipGetDeviceCount");
    }
    *h_count = n;
    return CSGN_OK;
}

int csgn_device_info(int device, char *h_name, size_t cap, int *h_cu_count, uint64_t *h_hbm_bytes)
{
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (h_name && cap) {
        strncpy(h_name, prop.gcnArchName, cap - 1);
        h_name[cap - 1] = 0;
    }
    if (h_cu_count)
        *h_cu_count = prop.multiProcessorCount;
    if (h_hbm_bytes)
        *h_hbm_bytes = (uint64_t)prop.totalGlobalMem;
    return CSGN_OK;
}

int csgn_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(CSGN_ERR_NO_DEVICE,
                    "csgn_init: no HIP device visible (%s); libcsgn_hip has no CPU fallback",
                    e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    REQUIRE(device >= 0 && device < n, "csgn_init: device %d out of range (have %d)", device, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CSGN_ERR_NO_DEVICE, "csgn_init: device %d is %s; the kernels are built for gfx950 only",
                    device, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    return CSGN_OK;
}

int csgn_malloc(void **d_ptr, size_t bytes)
{
    REQUIRE(d_ptr, "d_ptr is null");
    *d_ptr = nullptr;
    if (bytes == 0)
        return CSGN_OK;
    HIP_TRY(hipMalloc(d_ptr, bytes));
    return CSGN_OK;
}

int csgn_free(void *d_ptr)
{
    if (d_ptr)
        HIP_TRY(hipFree(d_ptr));
    return CSGN_OK;
}

int csgn_host_alloc(void **h_ptr, void **d_alias, size_t bytes)
{
    REQUIRE(h_ptr && d_alias && bytes, "null pointer or zero size");
    *h_ptr = *d_alias = nullptr;
    HIP_TRY(hipHostMalloc(h_ptr, bytes, hipHostMallocMapped));
    hipError_t e = hipHostGetDevicePointer(d_alias, *h_ptr, 0);
    if (e != hipSuccess) {
        (void)hipHostFree(*h_ptr);
        *h_ptr = *d_alias = nullptr;
        return hip_fail(e, "hipHostGetDevicePointer");
    }
    return CSGN_OK;
}

int csgn_host_free(void *h_ptr)
{
    if (h_ptr)
        HIP_TRY(hipHostFree(h_ptr));
    return CSGN_OK;
}

int csgn_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream)
{
    if (bytes == 0)
        return CSGN_OK;
    REQUIRE(d_dst && h_src, "null pointer");
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, S(stream)));
    return CSGN_OK;
}

int csgn_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream)
{
    if (bytes == 0)
        return CSGN_OK;
    REQUIRE(h_dst && d_src, "null pointer");
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    return CSGN_OK;
}

int csgn_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes, void *stream)
{
    if (bytes == 0)
        return CSGN_OK;
    REQUIRE(d_dst && d_src, "null pointer");
    HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, S(stream)));
    return CSGN_OK;
}

int csgn_memset(void *d_dst, int value, size_t bytes, void *stream)
{
    if (bytes == 0)
        return CSGN_OK;
    REQUIRE(d_dst, "null pointer");
    HIP_TRY(hipMemsetAsync(d_dst, value, bytes, S(stream)));
    return CSGN_OK;
}

int csgn_stream_create(void **stream)
{
    REQUIRE(stream, "stream is null");
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return CSGN_OK;
}

int csgn_stream_destroy(void *stream)
{
    if (stream)
        HIP_TRY(hipStreamDestroy(S(stream)));
    return CSGN_OK;
}

int csgn_stream_sync(void *stream)
{
    HIP_TRY(hipStreamSynchronize(S(stream)));
    return CSGN_OK;
}

int csgn_event_create(void **event)
{
    REQUIRE(event, "event is null");
    hipEvent_t ev;
    HIP_TRY(hipEventCreate(&ev));
    *event = ev;
    return CSGN_OK;
}

int csgn_event_destroy(void *event)
{
    if (event)
        HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return CSGN_OK;
}

int csgn_event_record(void *event, void *stream)
{
    REQUIRE(event, "event is null");
    HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(event), S(stream)));
    return CSGN_OK;
}

int csgn_event_sync(void *event)
{
    REQUIRE(event, "event is null");
    HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return CSGN_OK;
}

int csgn_event_elapsed_ms(void *start, void *stop, float *h_ms)
{
    REQUIRE(start && stop && h_ms, "null argument");
    HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    HIP_TRY(hipEventElapsedTime(h_ms, reinterpret_cast<hipEvent_t>(start),
                                reinterpret_cast<hipEvent_t>(stop)));
    return CSGN_OK;
}

/* ------------------------------------------------------------ host-side metadata -- */

uint64_t csgn_default_len(uint64_t n_bits) { return n_bits / 64 + ((n_bits % 64) ? 1 : 0); }

uint64_t csgn_context_s(uint64_t n_bits, uint64_t d) { return d ? n_bits / (2 * d) : 0; }

uint64_t csgn_mul_len(uint64_t n_bits, uint64_t len1, uint64_t len2)
{
    const uint64_t dl = csgn_default_len(n_bits);
    if (dl == 0)
        return 0;
    if (len1 == dl && len1 == len2)
        return len1;
    return ((len1 / dl) * len2) / dl * dl;
}

int csgn_bitlen_canonical(uint64_t n_bits, uint64_t terms, uint64_t *h_bitlen)
{
    REQUIRE(n_bits > 0, "n_bits must be > 0");
    REQUIRE(h_bitlen || terms == 0, "h_bitlen is null");
    const uint64_t dl = csgn_default_len(n_bits), rem = n_bits % 64;
    for (uint64_t t = 0; t < terms; ++t)
        for (uint64_t k = 0; k < dl; ++k)
            h_bitlen[t * dl + k] = (rem && k == dl - 1) ? rem : 64;
    return CSGN_OK;
}

int csgn_key_mask(uint64_t n_bits, const uint64_t *h_key, uint64_t d, uint64_t *h_mask)
{
    REQUIRE(n_bits > 0 && h_key && h_mask && d > 0, "bad argument");
    const uint64_t dl = csgn_default_len(n_bits);
    memset(h_mask, 0, dl * sizeof(uint64_t));
    for (uint64_t i = 0; i < d; ++i) {
        REQUIRE(h_key[i] < n_bits, "key index %llu (slot %llu) is outside [0,%llu)",
                (unsigned long long)h_key[i], (unsigned long long)i, (unsigned long long)n_bits);
        h_mask[h_key[i] / 64] |= 1ull << (63 - (h_key[i] % 64));
    }
    return CSGN_OK;
}

/* ---------------------------------------------------------------------- hot path -- */

int csgn_mul_uniform(uint64_t n_bits, uint64_t batch, uint64_t t1, uint64_t t2,
                     const uint64_t *d_left, const uint64_t *d_right, uint64_t *d_out,
                     uint64_t out_slots, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    if (batch == 0 || t1 == 0 || t2 == 0)
        return CSGN_OK;
    REQUIRE(d_left && d_right && d_out, "null device pointer");
    const uint64_t dl = csgn_default_len(n_bits);
    if (t1 >= (1ull << 31) || t2 >= (1ull << 31) || !product_below(t1, t2, dl, 1ull << 32))
        return fail(CSGN_ERR_UNSUPPORTED, "pair product of %llu x %llu terms exceeds 2^32 words",
                    (unsigned long long)t1, (unsigned long long)t2);
    if (!product_below(batch, t1 + t2, dl, 1ull << 60))
        return fail(CSGN_ERR_UNSUPPORTED, "batch of %llu pairs: operand size overflows", (unsigned long long)batch);
    HIP_TRY(csgn::mul_uniform(n_bits, batch, t1, t2, (const u64 *)d_left, (const u64 *)d_right,
                              (u64 *)d_out, out_slots, S(stream)));
    return CSGN_OK;
}

} // extern "C"

namespace {

// the plan step behind csgn_mul_ragged_plan and csgn_mul_plan_ragged; h_head_out (optional): the whole head block
struct PlanScratch {
    u64 *p = nullptr;
    size_t words = 0;
};

// owned: a csgn_mul_plan's own device block (it has to outlive the call: the class lists stay in it);
// nullptr: the calling thread's grow-only block
int plan_ragged(uint64_t batch, const uint64_t *d_off_left, const uint64_t *d_off_right,
                uint64_t *d_off_out, uint64_t h_plan[4], uint64_t *h_head_out, PlanScratch *owned, void *stream)
{
    REQUIRE(d_off_left && d_off_right && d_off_out && h_plan, "null pointer");
    // The call returns host numbers, so it ends with a stream synchronise anyway; its small device
    // scratch is kept per host thread and device (grow-only) because hipMalloc + hipFree around
    // every plan cost more than the plan (hipFree synchronises the whole device).
    static thread_local PlanScratch cache[16];
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const size_t need = csgn::mul_ragged_plan_scratch_words(batch);
    PlanScratch local;
    PlanScratch &sc = owned ? *owned : (dev >= 0 && dev < 16) ? cache[dev] : local;
    if (sc.words < need) {
        if (sc.p)
            (void)hipFree(sc.p);
        sc.p = nullptr;
        sc.words = 0;
        const size_t grow = need + need / 2 + 1024;
        HIP_TRY(hipMalloc((void **)&sc.p, grow * sizeof(u64)));
        sc.words = grow;
    }
    hipError_t e = csgn::mul_ragged_plan(batch, (const u64 *)d_off_left, (const u64 *)d_off_right,
                                         (u64 *)d_off_out, sc.p, S(stream));
    // the four plan numbers and, in the same copy, the plan's notes on huge pairs (csgn_mul_ragged
    // gives each of those a uniform launch of its own)
    std::vector<u64> head(csgn::mul_ragged_plan_head_words(), 0);
    if (e == hipSuccess)
        e = hipMemcpyAsync(head.data(), sc.p, head.size() * sizeof(u64), hipMemcpyDeviceToHost, S(stream));
    if (e == hipSuccess)
        e = hipStreamSynchronize(S(stream));
    if (e == hipSuccess) {
        memcpy(h_plan, head.data(), 4 * sizeof(u64));
        if (h_head_out)
            memcpy(h_head_out, head.data(), head.size() * sizeof(u64));
    }
    if (&sc == &local && local.p)
        (void)hipFree(local.p);
    if (e != hipSuccess)
        return hip_fail(e, "csgn_mul_ragged_plan");
    return CSGN_OK;
}

} // namespace

extern "C" {

int csgn_mul_ragged_plan(uint64_t batch, const uint64_t *d_off_left, const uint64_t *d_off_right,
                         uint64_t *d_off_out, uint64_t h_plan[4], void *stream)
{
    return plan_ragged(batch, d_off_left, d_off_right, d_off_out, h_plan, nullptr, nullptr, stream);
}

/* ---- the plan as an object of the caller's ---- */
struct csgn_mul_plan {
    csgn::MulPlanNotes notes;
    bool planned = false;
    bool trust = false;
    u64 *d_sum = nullptr;        // one device word for the checksum check
    int device = -1;
    PlanScratch work;            // the plan kernels' device block: the size-class lists live here
    int work_device = -1;
};

int csgn_mul_plan_create(csgn_mul_plan **plan)
{
    REQUIRE(plan, "plan is null");
    *plan = new csgn_mul_plan();
    return CSGN_OK;
}

void csgn_mul_plan_destroy(csgn_mul_plan *plan)
{
    if (!plan)
        return;
    if (plan->d_sum)
        (void)hipFree(plan->d_sum);
    if (plan->work.p)
        (void)hipFree(plan->work.p);
    delete plan;
}

int csgn_mul_plan_trust(csgn_mul_plan *plan, int trust)
{
    REQUIRE(plan, "plan is null");
    plan->trust = trust != 0;
    return CSGN_OK;
}

int csgn_mul_plan_ragged(csgn_mul_plan *plan, uint64_t batch, const uint64_t *d_off_left,
                         const uint64_t *d_off_right, uint64_t *d_off_out, uint64_t h_plan[4], void *stream)
{
    REQUIRE(plan, "plan is null");
    plan->planned = false;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (plan->work.p && plan->work_device != dev) {          // the object moved to another GPU: start over there
        (void)hipFree(plan->work.p);
        plan->work = PlanScratch();
    }
    plan->work_device = dev;
    std::vector<u64> head(csgn::mul_ragged_plan_head_words(), 0);
    if (int rc = plan_ragged(batch, d_off_left, d_off_right, d_off_out, h_plan, reinterpret_cast<uint64_t *>(head.data()),
                             &plan->work, stream))
        return rc;
    csgn::mul_plan_notes_from_head(plan->notes, (const u64 *)d_off_left, (const u64 *)d_off_right,
                                   (const u64 *)d_off_out, batch, head.data(), plan->work.p);
    plan->planned = true;
    return CSGN_OK;
}

int csgn_mul_plan_validate(csgn_mul_plan *plan, void *stream)
{
    REQUIRE(plan && plan->planned, "no plan: call csgn_mul_plan_ragged first");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (plan->d_sum && plan->device != dev) {
        (void)hipFree(plan->d_sum);
        plan->d_sum = nullptr;
    }
    const u32 slots = csgn::offsets_checksum_words();
    if (!plan->d_sum) {
        HIP_TRY(hipMalloc((void **)&plan->d_sum, slots * 8));
        plan->device = dev;
    }
    const csgn::MulPlanNotes &n = plan->notes;
    HIP_TRY(csgn::offsets_checksum(n.batch, n.offL, n.offR, n.offOut, plan->d_sum, S(stream)));
    std::vector<u64> part(slots, 0);
    HIP_TRY(hipMemcpyAsync(part.data(), plan->d_sum, slots * 8, hipMemcpyDeviceToHost, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    u64 now = 0;
    for (u64 v : part)
        now += v;
    if (now != n.checksum)
        return fail(CSGN_ERR_INVALID, "the offset arrays changed since csgn_mul_plan_ragged: plan again");
    return CSGN_OK;
}

int csgn_mul_planned(csgn_mul_plan *plan, uint64_t n_bits, const uint64_t *d_left, const uint64_t *d_right,
                     uint64_t *d_out, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    REQUIRE(plan && plan->planned, "no plan: call csgn_mul_plan_ragged first");
    const csgn::MulPlanNotes &n = plan->notes;
    if (n.batch == 0 || n.max_t1 == 0 || n.max_t2 == 0 || n.total == 0)
        return CSGN_OK;
    REQUIRE(d_left && d_right && d_out, "null device pointer");
    const uint64_t dl = csgn_default_len(n_bits);
    if (n.max_t1 >= (1ull << 31) || n.max_t2 >= (1ull << 31) || !product_below(n.max_t1, n.max_t2, dl, 1ull << 32))
        return fail(CSGN_ERR_UNSUPPORTED, "pair product of %llu x %llu terms exceeds 2^32 words",
                    (unsigned long long)n.max_t1, (unsigned long long)n.max_t2);
    // host copies of offsets are only used for huge pairs: that is when stale offsets would give wrong words
    if (n.n != 0 && !plan->trust)
        if (int rc = csgn_mul_plan_validate(plan, stream))
            return rc;
    hipError_t e = csgn::mul_ragged(n_bits, n.batch, (const u64 *)d_left, n.offL, (const u64 *)d_right, n.offR,
                                    (u64 *)d_out, n.offOut, n.max_t1, n.max_t2, n.total, S(stream), &plan->notes);
    if (e == hipErrorInvalidValue)
        return fail(CSGN_ERR_UNSUPPORTED, "ragged batch too large for one call (2^32 pairs / one pair's tile grid)");
    HIP_TRY(e);
    return CSGN_OK;
}

uint64_t csgn_mul_ragged_async_plan_words(uint64_t batch) { return csgn::mul_ragged_async_plan_words(batch); }

int csgn_mul_ragged_async(uint64_t n_bits, uint64_t batch,
                          const uint64_t *d_left, const uint64_t *d_off_left,
                          const uint64_t *d_right, const uint64_t *d_off_right,
                          uint64_t *d_out, uint64_t *d_off_out, uint64_t out_capacity_terms,
                          uint64_t *d_plan, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    if (batch == 0)
        return CSGN_OK;
    REQUIRE(d_off_left && d_off_right && d_off_out && d_plan, "null pointer");
    REQUIRE(out_capacity_terms == 0 || (d_left && d_right && d_out), "null device pointer");
    if (!product_below(out_capacity_terms, csgn_default_len(n_bits), 1, 1ull << 60))
        return fail(CSGN_ERR_UNSUPPORTED, "output capacity overflows");
    hipError_t e = csgn::mul_ragged_async(n_bits, batch, (const u64 *)d_left, (const u64 *)d_off_left,
                                          (const u64 *)d_right, (const u64 *)d_off_right, (u64 *)d_out,
                                          (u64 *)d_off_out, out_capacity_terms, (u64 *)d_plan, S(stream));
    if (e == hipErrorInvalidValue)
        return fail(CSGN_ERR_UNSUPPORTED, "ragged batch too large for one call (2^32 pairs)");
    HIP_TRY(e);
    return CSGN_OK;
}

int csgn_mul_ragged_async_result(const uint64_t *d_plan, uint64_t h_result[5], void *stream)
{
    REQUIRE(d_plan && h_result, "null pointer");
    u64 head[12] = {0};                       // [gate: 8 words][plan4]
    HIP_TRY(hipMemcpyAsync(head, d_plan, sizeof(head), hipMemcpyDeviceToHost, S(stream)));
    HIP_TRY(hipStreamSynchronize(S(stream)));
    for (int i = 0; i < 4; ++i)
        h_result[i] = head[8 + i];
    h_result[4] = head[1];
    return CSGN_OK;
}

int csgn_mul_ragged(uint64_t n_bits, uint64_t batch,
                    const uint64_t *d_left, const uint64_t *d_off_left,
                    const uint64_t *d_right, const uint64_t *d_off_right,
                    uint64_t *d_out, const uint64_t *d_off_out,
                    uint64_t max_t1, uint64_t max_t2, uint64_t total_out_terms, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    if (batch == 0 || max_t1 == 0 || max_t2 == 0 || total_out_terms == 0)
        return CSGN_OK;
    REQUIRE(d_left && d_right && d_out && d_off_left && d_off_right && d_off_out, "null device pointer");
    const uint64_t dl = csgn_default_len(n_bits);
    if (max_t1 >= (1ull << 31) || max_t2 >= (1ull << 31) || !product_below(max_t1, max_t2, dl, 1ull << 32))
        return fail(CSGN_ERR_UNSUPPORTED, "pair product of %llu x %llu terms exceeds 2^32 words",
                    (unsigned long long)max_t1, (unsigned long long)max_t2);
    hipError_t e = csgn::mul_ragged(n_bits, batch, (const u64 *)d_left, (const u64 *)d_off_left,
                                    (const u64 *)d_right, (const u64 *)d_off_right, (u64 *)d_out,
                                    (const u64 *)d_off_out, max_t1, max_t2, total_out_terms, S(stream));
    if (e == hipErrorInvalidValue)
        return fail(CSGN_ERR_UNSUPPORTED, "ragged batch too large for one call (2^32 pairs / one pair's tile grid)");
    HIP_TRY(e);
    return CSGN_OK;
}

int csgn_add_uniform(uint64_t n_bits, uint64_t batch, uint64_t t1, uint64_t t2,
                     const uint64_t *d_left, const uint64_t *d_right, uint64_t *d_out, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    const uint64_t dl = csgn_default_len(n_bits);
    if (t1 >= (1ull << 31) || t2 >= (1ull << 31) || (t1 + t2) * dl >= (1ull << 31))
        return fail(CSGN_ERR_UNSUPPORTED, "sum of %llu + %llu terms exceeds 2^31 words per pair",
                    (unsigned long long)t1, (unsigned long long)t2);
    if (batch == 0 || t1 + t2 == 0)
        return CSGN_OK;
    REQUIRE(d_out && (d_left || t1 == 0) && (d_right || t2 == 0), "null device pointer");
    if (!product_below(batch, t1 + t2, dl, 1ull << 60))
        return fail(CSGN_ERR_UNSUPPORTED, "batch of %llu pairs: size overflows", (unsigned long long)batch);
    HIP_TRY(csgn::add_uniform(n_bits, batch, t1, t2, (const u64 *)d_left, (const u64 *)d_right,
                              (u64 *)d_out, S(stream)));
    return CSGN_OK;
}

int csgn_add_ragged(uint64_t n_bits, uint64_t batch,
                    const uint64_t *d_left, const uint64_t *d_off_left,
                    const uint64_t *d_right, const uint64_t *d_off_right,
                    uint64_t *d_out, uint64_t *d_off_out, uint64_t total_terms_out, void *stream)
{
    return csgn_add_ragged_bounded(n_bits, batch, 0, 0, d_left, d_off_left, d_right, d_off_right, d_out, d_off_out,
                                   total_terms_out, stream);
}

int csgn_add_ragged_bounded(uint64_t n_bits, uint64_t batch, uint64_t max_t1, uint64_t max_t2,
                            const uint64_t *d_left, const uint64_t *d_off_left,
                            const uint64_t *d_right, const uint64_t *d_off_right,
                            uint64_t *d_out, uint64_t *d_off_out, uint64_t total_terms_out, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    REQUIRE(d_off_left && d_off_right && d_off_out, "null offset pointer");
    REQUIRE(total_terms_out == 0 || batch == 0 || (d_left && d_right && d_out), "null device pointer");
    const bool bounded = max_t1 != 0 || max_t2 != 0;
    REQUIRE(!bounded || (max_t1 < (1ull << 31) && max_t2 < (1ull << 31) &&
                         !product_below(batch, max_t1 + max_t2, 1, total_terms_out)),
            "bounds %llu + %llu cannot hold: %llu sums of at most that many terms are fewer than total_terms_out = %llu",
            (unsigned long long)max_t1, (unsigned long long)max_t2, (unsigned long long)batch, (unsigned long long)total_terms_out);
    hipError_t e = csgn::add_ragged(n_bits, batch, (const u64 *)d_left, (const u64 *)d_off_left,
                                    (const u64 *)d_right, (const u64 *)d_off_right, (u64 *)d_out,
                                    (u64 *)d_off_out, total_terms_out, S(stream), false, max_t1, max_t2);
    if (e == hipErrorInvalidValue)
        return fail(CSGN_ERR_UNSUPPORTED, "ragged batch of 2^32 or more pairs; split it");
    HIP_TRY(e);
    return CSGN_OK;
}

int csgn_small_ops(uint64_t n_bits, uint64_t count, const csgn_small_op *d_ops, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    if (count == 0)
        return CSGN_OK;
    REQUIRE(d_ops, "d_ops is null");
    const hipError_t e = csgn::small_ops(n_bits, count, d_ops, S(stream));
    if (e == hipErrorInvalidValue)
        return fail(CSGN_ERR_UNSUPPORTED, "more than 2^24 - 1 operations in one call; split the list");
    HIP_TRY(e);
    return CSGN_OK;
}

size_t csgn_decrypt_scratch_bytes(uint64_t batch, uint64_t total_terms)
{
    return csgn::decrypt_scratch_bytes(batch, total_terms);
}

int csgn_decrypt_uniform(uint64_t n_bits, uint64_t batch, uint64_t terms,
                         const uint64_t *d_terms, const uint64_t *d_mask,
                         uint8_t *d_bits, void *d_scratch, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    if (batch == 0)
        return CSGN_OK;
    REQUIRE(d_mask && d_bits && d_scratch && (d_terms || terms == 0), "null device pointer");
    if (!product_below(batch, terms, csgn_default_len(n_bits), 1ull << 60))
        return fail(CSGN_ERR_UNSUPPORTED, "batch of %llu x %llu terms: size overflows",
                    (unsigned long long)batch, (unsigned long long)terms);
    HIP_TRY(csgn::decrypt(n_bits, batch, terms, batch * terms, (const u64 *)d_terms, nullptr,
                          (const u64 *)d_mask, d_bits, d_scratch, S(stream)));
    return CSGN_OK;
}

int csgn_decrypt_ragged_bounded(uint64_t n_bits, uint64_t batch, uint64_t total_terms, uint64_t max_terms,
                                const uint64_t *d_terms, const uint64_t *d_off, const uint64_t *d_mask,
                                uint8_t *d_bits, void *d_scratch, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    if (batch == 0)
        return CSGN_OK;
    REQUIRE(d_off && d_mask && d_bits && d_scratch && (d_terms || total_terms == 0),
            "null device pointer");
    REQUIRE(max_terms == 0 || !product_below(batch, max_terms, 1, total_terms),
            "max_terms = %llu cannot hold: %llu ciphertexts of at most that many terms are fewer than total_terms = %llu",
            (unsigned long long)max_terms, (unsigned long long)batch, (unsigned long long)total_terms);
    HIP_TRY(csgn::decrypt(n_bits, batch, 0, total_terms, (const u64 *)d_terms, (const u64 *)d_off,
                          (const u64 *)d_mask, d_bits, d_scratch, S(stream), max_terms));
    return CSGN_OK;
}

int csgn_decrypt_ragged(uint64_t n_bits, uint64_t batch, uint64_t total_terms,
                        const uint64_t *d_terms, const uint64_t *d_off, const uint64_t *d_mask,
                        uint8_t *d_bits, void *d_scratch, void *stream)
{
    return csgn_decrypt_ragged_bounded(n_bits, batch, total_terms, 0, d_terms, d_off, d_mask, d_bits, d_scratch, stream);
}

size_t csgn_decrypt_combined_scratch_bytes(uint64_t batch, uint64_t t1, uint64_t t2)
{
    return csgn::decrypt_combined_scratch_bytes(batch, t1, t2);
}

int csgn_decrypt_product_uniform(uint64_t n_bits, uint64_t batch, uint64_t t1, uint64_t t2,
                                 const uint64_t *d_left, const uint64_t *d_right,
                                 const uint64_t *d_mask, uint8_t *d_bits, void *d_scratch, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    if (batch == 0)
        return CSGN_OK;
    REQUIRE(d_mask && d_bits && d_scratch && (d_left || t1 == 0) && (d_right || t2 == 0),
            "null device pointer");
    if (!product_below(batch, t1, csgn_default_len(n_bits), 1ull << 60) ||
        !product_below(batch, t2, csgn_default_len(n_bits), 1ull << 60))
        return fail(CSGN_ERR_UNSUPPORTED, "batch of %llu ciphertexts: size overflows", (unsigned long long)batch);
    HIP_TRY(csgn::decrypt_combined(n_bits, batch, t1, t2, (const u64 *)d_left, (const u64 *)d_right,
                                   (const u64 *)d_mask, true, d_bits, d_scratch, S(stream)));
    return CSGN_OK;
}

int csgn_decrypt_sum_uniform(uint64_t n_bits, uint64_t batch, uint64_t t1, uint64_t t2,
                             const uint64_t *d_left, const uint64_t *d_right,
                             const uint64_t *d_mask, uint8_t *d_bits, void *d_scratch, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    if (batch == 0)
        return CSGN_OK;
    REQUIRE(d_mask && d_bits && d_scratch && (d_left || t1 == 0) && (d_right || t2 == 0),
            "null device pointer");
    if (!product_below(batch, t1, csgn_default_len(n_bits), 1ull << 60) ||
        !product_below(batch, t2, csgn_default_len(n_bits), 1ull << 60))
        return fail(CSGN_ERR_UNSUPPORTED, "batch of %llu ciphertexts: size overflows", (unsigned long long)batch);
    HIP_TRY(csgn::decrypt_combined(n_bits, batch, t1, t2, (const u64 *)d_left, (const u64 *)d_right,
                                   (const u64 *)d_mask, false, d_bits, d_scratch, S(stream)));
    return CSGN_OK;
}

size_t csgn_compact_scratch_bytes(uint64_t n_bits, uint64_t batch, uint64_t total_terms)
{
    if (n_bits == 0 || !csgn::compact_supported(n_bits))
        return 0;
    return csgn::compact_scratch_bytes(n_bits, batch, total_terms);
}

int csgn_compact_ragged(uint64_t n_bits, uint64_t batch, uint64_t total_terms, uint64_t max_terms,
                        const uint64_t *d_terms, const uint64_t *d_off,
                        uint64_t *d_out, uint64_t *d_off_out, void *d_scratch, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    if (batch == 0)
        return CSGN_OK;
    REQUIRE(d_off && d_off_out && d_scratch && ((d_terms && d_out) || total_terms == 0),
            "null device pointer");
    REQUIRE(d_terms != d_out || total_terms == 0, "compaction is not done in place");
    hipError_t e = csgn::compact(n_bits, batch, total_terms, max_terms, (const u64 *)d_terms, (const u64 *)d_off,
                                 (u64 *)d_out, (u64 *)d_off_out, d_scratch, S(stream));
    if (e == hipErrorInvalidValue)
        return fail(CSGN_ERR_UNSUPPORTED,
                    "compaction handles fewer than 2^31 ciphertexts and terms per call");
    HIP_TRY(e);
    return CSGN_OK;
}

int csgn_encrypt_explicit(uint64_t n_bits, uint64_t d, uint64_t batch,
                          const uint8_t *d_plain, const uint64_t *d_rnd,
                          const uint32_t *d_chosen, const uint8_t *d_last,
                          const uint64_t *d_mask, uint64_t *d_out, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    if (batch == 0)
        return CSGN_OK;
    REQUIRE(d >= 1, "d must be >= 1");
    REQUIRE(d_plain && d_rnd && d_chosen && d_last && d_mask && d_out, "null device pointer");
    HIP_TRY(csgn::encrypt(n_bits, d, batch, d_plain, (const u64 *)d_rnd, d_chosen, d_last,
                          (const u64 *)d_mask, (u64 *)d_out, S(stream)));
    return CSGN_OK;
}

int csgn_rng_from_os(csgn_rng *h_rng, uint32_t rounds)
{
    REQUIRE(h_rng, "h_rng is null");
    REQUIRE(rounds == 8 || rounds == 12 || rounds == 20, "rounds must be 8, 12 or 20");
    unsigned char buf[40];
    size_t got = 0;
    while (got < sizeof(buf)) {
        const ssize_t r = getrandom(buf + got, sizeof(buf) - got, 0);
        if (r < 0) {
            if (errno == EINTR)
                continue;
            return fail(CSGN_ERR_INVALID, "getrandom failed: %s", strerror(errno));
        }
        got += (size_t)r;
    }
    memcpy(h_rng->key, buf, 32);
    memcpy(&h_rng->nonce, buf + 32, 8);
    h_rng->rounds = rounds;
    h_rng->reserved = 0;
    memset(buf, 0, sizeof(buf));
    return CSGN_OK;
}

int csgn_rng_from_seed(csgn_rng *h_rng, uint64_t seed, uint32_t rounds)
{
    REQUIRE(h_rng, "h_rng is null");
    REQUIRE(rounds == 8 || rounds == 12 || rounds == 20, "rounds must be 8, 12 or 20");
    for (int i = 0; i < 4; ++i) {
        const uint64_t w = csgn_splitmix64(seed + CSGN_GOLDEN * (uint64_t)(i + 1));
        h_rng->key[2 * i] = (uint32_t)w;
        h_rng->key[2 * i + 1] = (uint32_t)(w >> 32);
    }
    h_rng->nonce = csgn_splitmix64(seed ^ 0xD1B54A32D192ED03ull);
    h_rng->rounds = rounds;
    h_rng->reserved = 0;
    return CSGN_OK;
}

int csgn_encrypt_keyed_layout(uint64_t n_bits, uint32_t *h_units, uint32_t *h_passes, uint32_t *h_group)
{
    if (int rc = check_n(n_bits))
        return rc;
    REQUIRE(h_units && h_passes && h_group, "null output");
    csgn::encrypt_keyed_layout(n_bits, h_units, h_passes, h_group);
    return CSGN_OK;
}

int csgn_encrypt_keyed(uint64_t n_bits, uint64_t d, uint64_t batch, uint64_t first_ciphertext,
                       const uint8_t *d_plain, const uint64_t *d_key, const uint64_t *d_mask,
                       const csgn_rng *h_rng, uint64_t *d_out, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    REQUIRE(h_rng, "h_rng is null");
    REQUIRE(h_rng->rounds == 8 || h_rng->rounds == 12 || h_rng->rounds == 20, "rng rounds must be 8, 12 or 20");
    if (batch == 0)
        return CSGN_OK;
    REQUIRE(d >= 1 && d < (1ull << 32), "d must be in [1, 2^32)");
    REQUIRE(d_plain && d_key && d_mask && d_out, "null device pointer");
    REQUIRE(first_ciphertext + batch >= first_ciphertext && first_ciphertext + batch < (1ull << 56),
            "ciphertext index range too large");
    hipError_t e = csgn::encrypt_keyed(n_bits, d, batch, first_ciphertext, d_plain, (const u64 *)d_key,
                                       (const u64 *)d_mask, h_rng->key, h_rng->nonce, h_rng->rounds, nullptr,
                                       (u64 *)d_out, S(stream));
    if (e == hipErrorInvalidValue)
        return fail(CSGN_ERR_UNSUPPORTED, "batch too large for one launch");
    HIP_TRY(e);
    return CSGN_OK;
}

int csgn_encrypt_mul_keyed(uint64_t n_bits, uint64_t d, uint64_t batch, uint64_t first_ciphertext,
                           const uint8_t *d_plain_a, const uint8_t *d_plain_b, const uint64_t *d_key,
                           const uint64_t *d_mask, const csgn_rng *h_rng_a, const csgn_rng *h_rng_b,
                           uint64_t *d_out, uint8_t *d_bits, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    REQUIRE(h_rng_a && h_rng_b, "h_rng is null");
    REQUIRE(h_rng_a->rounds == 8 || h_rng_a->rounds == 12 || h_rng_a->rounds == 20, "rng rounds must be 8, 12 or 20");
    REQUIRE(h_rng_a->rounds == h_rng_b->rounds, "both generators must use the same number of rounds");
    REQUIRE(memcmp(h_rng_a->key, h_rng_b->key, sizeof(h_rng_a->key)) != 0 || h_rng_a->nonce != h_rng_b->nonce,
            "the two operands must draw from different streams (same key AND nonce given)");
    if (batch == 0)
        return CSGN_OK;
    REQUIRE(d >= 1 && d < (1ull << 32), "d must be in [1, 2^32)");
    REQUIRE(d_plain_a && d_plain_b && d_key && d_mask && d_out, "null device pointer");
    REQUIRE(first_ciphertext + batch >= first_ciphertext && first_ciphertext + batch < (1ull << 56),
            "ciphertext index range too large");
    hipError_t e = csgn::encrypt_mul_keyed(n_bits, d, batch, first_ciphertext, d_plain_a, d_plain_b, (const u64 *)d_key,
                                           (const u64 *)d_mask, h_rng_a->key, h_rng_a->nonce, h_rng_b->key,
                                           h_rng_b->nonce, h_rng_a->rounds, nullptr, (u64 *)d_out, d_bits, S(stream));
    if (e == hipErrorInvalidValue)
        return fail(CSGN_ERR_UNSUPPORTED, "batch too large for one launch");
    HIP_TRY(e);
    return CSGN_OK;
}

int csgn_encrypt_device_rng(uint64_t n_bits, uint64_t d, uint64_t batch,
                            const uint8_t *d_plain, const uint64_t *d_key,
                            const uint64_t *d_mask, uint64_t seed, uint64_t *d_out, void *stream)
{
    csgn_rng rng;
    if (int rc = csgn_rng_from_seed(&rng, seed, 8))
        return rc;
    return csgn_encrypt_keyed(n_bits, d, batch, 0, d_plain, d_key, d_mask, &rng, d_out, stream);
}

int csgn_permute_uniform(uint64_t n_bits, uint64_t batch, uint64_t terms_in, int per_term,
                         const uint64_t *d_terms, const uint32_t *d_perm, uint64_t *d_out,
                         void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    if (batch == 0)
        return CSGN_OK;
    REQUIRE(d_perm && d_out && (d_terms || terms_in == 0), "null device pointer");
    HIP_TRY(csgn::permute(n_bits, batch, terms_in, per_term != 0, (const u64 *)d_terms, d_perm,
                          (u64 *)d_out, S(stream)));
    return CSGN_OK;
}

size_t csgn_bitlen_scratch_bytes(uint64_t len_words) { return csgn::bitlen_scratch_bytes(len_words); }

int csgn_decrypt_bitlen(uint64_t n_bits, uint64_t d, uint64_t len_words, const uint64_t *d_v,
                        const uint64_t *d_bitlen, const uint64_t *d_key, uint8_t *d_bit, void *d_scratch,
                        void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    REQUIRE(d >= 1, "d must be >= 1");
    REQUIRE(d_key && d_bit && d_scratch && ((d_v && d_bitlen) || len_words == 0), "null device pointer");
    REQUIRE(len_words < (1ull << 40), "ciphertext too long");
    hipError_t e = csgn::decrypt_bitlen(n_bits, d, len_words, (const u64 *)d_v, (const u64 *)d_bitlen,
                                        (const u64 *)d_key, d_bit, d_scratch, S(stream));
    if (e == hipErrorInvalidValue)
        return fail(CSGN_ERR_UNSUPPORTED, "ciphertext too long for one launch");
    HIP_TRY(e);
    return CSGN_OK;
}

int csgn_permute_bitlen(uint64_t n_bits, uint64_t len_words, const uint64_t *d_v, const uint64_t *d_bitlen,
                        const uint32_t *d_perm, uint64_t *d_out, void *d_scratch, void *stream)
{
    if (int rc = check_n(n_bits))
        return rc;
    REQUIRE(d_perm && d_out && d_scratch && ((d_v && d_bitlen) || len_words == 0), "null device pointer");
    REQUIRE(len_words < (1ull << 40), "ciphertext too long");
    hipError_t e = csgn::permute_bitlen(n_bits, len_words, (const u64 *)d_v, (const u64 *)d_bitlen, d_perm,
                                        (u64 *)d_out, d_scratch, S(stream));
    if (e == hipErrorInvalidValue)
        return fail(CSGN_ERR_UNSUPPORTED, "ciphertext too long for one launch");
    HIP_TRY(e);
    return CSGN_OK;
}

int csgn_synth_fill(uint64_t seed, uint64_t n_bits, uint64_t first_word, uint64_t n_words,
                    uint64_t *d_out, void *stream)
{
    REQUIRE(n_bits > 0, "n_bits must be > 0");
    if (n_words == 0)
        return CSGN_OK;
    REQUIRE(d_out, "null device pointer");
    HIP_TRY(csgn::synth_fill(seed, n_bits, first_word, n_words, (u64 *)d_out, S(stream)));
    return CSGN_OK;
}

int csgn_digest(const uint64_t *d_words, uint64_t n_words, uint64_t first_index,
                uint64_t *d_digest, void *stream)
{
    REQUIRE(d_digest, "null device pointer");
    if (n_words == 0)
        return CSGN_OK;
    REQUIRE(d_words, "null device pointer");
    HIP_TRY(csgn::digest((const u64 *)d_words, n_words, first_index, (u64 *)d_digest, S(stream)));
    return CSGN_OK;
}

const char *csgn_mul_uniform_kernel(uint64_t n_bits, uint64_t pairs, uint64_t t1, uint64_t t2)
{
    return csgn::mul_uniform_kernel_name(n_bits, pairs, t1, t2);
}

/* ------------------------------------------------------------------ tuning ---- */

int csgn_set_tuning(const char *key, int value)
{
    REQUIRE(key, "key is null");
    if (!csgn::tune_set(key, value))
        return fail(CSGN_ERR_INVALID, "csgn_set_tuning: no knob named '%s'", key);
    return CSGN_OK;
}

int csgn_get_tuning(const char *key, int *h_value)
{
    REQUIRE(key && h_value, "null argument");
    if (!csgn::tune_get(key, h_value))
        return fail(CSGN_ERR_INVALID, "csgn_get_tuning: no knob named '%s'", key);
    return CSGN_OK;
}

void csgn_reset_tuning(void) { csgn::tune_reset(); }

const char *csgn_tuning_name(int index) { return csgn::tune_name(index); }

/* debug hook used by the CPU tests to pin the division-by-invariant helper */
uint32_t csgn_debug_fastdiv(uint32_t n, uint32_t d)
{
    return csgn_fastdiv(n, csgn_fastdiv_make(d));
}

} // extern "C"
