// wpattern_bench.hip -- dev microbenchmark: does the ORDER in which a workgroup's four waves write their 16-byte units
// matter to the memory side?  (the wave-cooperative ragged multiply's stores alone run at 5.4-5.5 TB/s,
// the flat kernels' at 7.3-7.5.)  Both kernels write the same bytes with non-temporal 16-byte stores, XCD-contiguous
// workgroup order, S units per wave, 64 units (1 KiB) per store instruction:
//   wave-contiguous : wave w of workgroup b writes units [(4b + w) S, (4b + w + 1) S) front to back      (the coop kernel)
//   interleaved     : workgroup b writes [4b S, 4(b + 1) S) in 4 KiB pieces, wave w taking KiB w of every piece
// with an optional delay of D dependent integer operations between stores (the walk).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/wpattern_bench tools/wpattern_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned int unit16;
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ inline u32 xcd_contiguous_block(u32 b, u32 nblocks)
{
    const u32 q = nblocks >> 3, r = nblocks & 7u, x = b & 7u;
    return x * q + min(x, r) + (b >> 3);
}

template <bool INTERLEAVED>
__global__ void __launch_bounds__(256) k_write(unit16 *out, u32 S, u32 delay)
{
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u32 b = xcd_contiguous_block(blockIdx.x, gridDim.x);
    unit16 v = {lane, wave, b, 0u};
    u32 x = lane * 2654435761u;
    for (u32 i = 0; i < S / 64u; ++i) {
        for (u32 d = 0; d < delay; ++d)
            x = x * 1664525u + 1013904223u;                       // dependent integer work between stores
        v.w = x;
        const u64 at = INTERLEAVED ? (u64)b * 4u * S + ((u64)i * 4u + wave) * 64u + lane
                                   : ((u64)b * 4u + wave) * S + (u64)i * 64u + lane;
        __builtin_nontemporal_store(v, out + at);
    }
}

int main(int argc, char **argv)
{
    const u64 bytes = (argc > 1 ? (u64)atoll(argv[1]) : 2400ull) << 20;
    unit16 *out;
    CK(hipMalloc(&out, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (u32 S : {1024u, 4096u})
        for (u32 delay : {0u, 16u, 64u})
            for (int inter = 0; inter < 2; ++inter) {
                const u32 wgs = (u32)(bytes / 16 / (4ull * S));
                float best = 1e9f;
                for (int it = 0; it < 5; ++it) {
                    CK(hipEventRecord(e0));
                    if (inter)
                        k_write<true><<<wgs, 256>>>(out, S, delay);
                    else
                        k_write<false><<<wgs, 256>>>(out, S, delay);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (it >= 2 && ms < best)
                        best = ms;
                }
                printf("%-16s %5u units a wave, %3u dependent ops between stores: %7.3f ms = %7.1f GB/s\n",
                       inter ? "interleaved" : "wave-contiguous", S, delay, best, (double)wgs * 4 * S * 16 / best / 1e6);
            }
    return 0;
}
