// sprefetch_bench.hip -- dev microbenchmark: how fast can waves pull cold HBM lines into the L2 WITHOUT the vector
// memory path, by scalar loads (s_load_dword, one per 128-byte line, results unused)?  The question behind it
// (DESIGN.md 4.4d): the ragged multiply's vector path is one in-order queue per CU that stalls behind stores and cold
// operand misses; an operand touch issued from the scalar unit would bypass it -- if the scalar cache's miss path has
// the throughput.  Compared with the vector touch (one dword per lane and 128-byte line: 64 lines per instruction).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/sprefetch_bench tools/sprefetch_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef unsigned int u32;

// every wave touches `lines` consecutive 128-byte lines starting at its own place, `inflight` scalar loads between waits
__global__ void __launch_bounds__(256) touch_scalar(const char *base, u32 lines, u32 inflight, u32 *sink)
{
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    const char *p = base + (u64)wave * lines * 128u;
    // every load writes the SAME scalar register, which stays live (read-write operand) until the last wait: a destination the
    // compiler took for dead would be handed out again while loads that write it are still in flight
    u32 v = 0;
    for (u32 i = 0; i < lines; i += inflight) {
        for (u32 j = 0; j < inflight && i + j < lines; ++j)
            asm volatile("s_load_dword %0, %1, 0x0" : "+s"(v) : "s"(p + (u64)(i + j) * 128u));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v) : : "memory");
    }
    if (v == 0x12345u)
        *sink = v;
}

// the same lines by the vector path: lane l of a wave reads one dword of line (64 * i + l)
__global__ void __launch_bounds__(256) touch_vector(const char *base, u32 lines, u32 *sink)
{
    const u32 wave = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const char *p = base + (u64)wave * lines * 128u;
    u32 acc = 0;
    for (u32 i = lane; i < lines; i += 64u)
        acc += *reinterpret_cast<const u32 *>(p + (u64)i * 128u);
    if (acc == 12345u)
        *sink = acc;
}

int main(int argc, char **argv)
{
    const u64 total = (argc > 1 ? (u64)atoll(argv[1]) : 1024ull) << 20;   // MiB to touch per launch
    char *buf[3];
    u32 *sink;
    for (int k = 0; k < 3; ++k) {
        CK(hipMalloc(&buf[k], total));
        CK(hipMemset(buf[k], k + 1, total));
    }
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (u32 lines : {16u, 128u, 1024u})
        for (u32 inflight : {1u, 4u, 16u, 64u}) {
            const u64 waves = total / (128ull * lines);
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {                       // three buffers in turn: cold lines every time (3 x total > caches)
                CK(hipEventRecord(e0));
                touch_scalar<<<(u32)(waves / 4), 256>>>(buf[it % 3], lines, inflight, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 3 && ms < best)
                    best = ms;
            }
            printf("scalar touch: %4u lines per wave, %2u in flight: %8.3f ms = %7.1f GB/s of lines\n", lines, inflight, best, total / best / 1e6);
        }
    for (u32 lines : {64u, 1024u}) {
        const u64 waves = total / (128ull * lines);
        float best = 1e9f;
        for (int it = 0; it < 6; ++it) {
            CK(hipEventRecord(e0));
            touch_vector<<<(u32)(waves / 4), 256>>>(buf[it % 3], lines, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 3 && ms < best)
                best = ms;
        }
        printf("vector touch: %4u lines per wave               : %8.3f ms = %7.1f GB/s of lines\n", lines, best, total / best / 1e6);
    }
    return 0;
}
