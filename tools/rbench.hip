// rbench.hip -- dev microbenchmark: read-stream and copy shapes on MI355X HBM.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/rbench tools/rbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned int unit16;
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ inline u32 fold(unit16 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// read: block owns K*256 contiguous units, K independent loads per lane; result folded, rarely stored
template <int K>
__global__ void __launch_bounds__(256) read_chunk(const unit16 *in, u64 n, u32 *sink)
{
    const u64 base = (u64)blockIdx.x * (256 * K) + threadIdx.x;
    unit16 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { u64 i = base + (u64)k * 256; v[k] = i < n ? in[i] : unit16{0, 0, 0, 0}; }
    u32 acc = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) acc ^= fold(v[k]);
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) read_stride(const unit16 *in, u64 n, u32 *sink)
{
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 s = (u64)gridDim.x * 256;
    u32 acc = 0;
    for (; i < n; i += s) acc ^= fold(in[i]);
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

// copy / 2-in-1-out
template <int K, bool NT>
__global__ void __launch_bounds__(256) and_chunk(const unit16 *a, const unit16 *b, unit16 *o, u64 n)
{
    const u64 base = (u64)blockIdx.x * (256 * K) + threadIdx.x;
    unit16 x[K], y[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { u64 i = base + (u64)k * 256; if (i < n) { x[k] = a[i]; y[k] = b[i]; } }
#pragma unroll
    for (int k = 0; k < K; ++k) { u64 i = base + (u64)k * 256; if (i < n) { unit16 r = x[k] & y[k]; if (NT) __builtin_nontemporal_store(r, o + i); else o[i] = r; } }
}

template <int K, bool NT>
__global__ void __launch_bounds__(256) copy_chunk(const unit16 *a, unit16 *o, u64 n)
{
    const u64 base = (u64)blockIdx.x * (256 * K) + threadIdx.x;
    unit16 x[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { u64 i = base + (u64)k * 256; if (i < n) x[k] = a[i]; }
#pragma unroll
    for (int k = 0; k < K; ++k) { u64 i = base + (u64)k * 256; if (i < n) { if (NT) __builtin_nontemporal_store(x[k], o + i); else o[i] = x[k]; } }
}

__device__ inline u32 xcd_contig(u32 b, u32 n) { const u32 q = n >> 3, r = n & 7u, x = b & 7u; return x * q + min(x, r) + (b >> 3); }

template <bool XCD>
__global__ void __launch_bounds__(256) read_one(const unit16 *in, u64 n, u32 *sink)
{
    const u32 b = XCD ? xcd_contig(blockIdx.x, gridDim.x) : blockIdx.x;
    const u64 i = (u64)b * 256 + threadIdx.x;
    u32 acc = i < n ? fold(in[i]) : 0;
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}
template <bool XCD, bool NT>
__global__ void __launch_bounds__(256) and_one(const unit16 *a, const unit16 *b, unit16 *o, u64 n)
{
    const u32 blk = XCD ? xcd_contig(blockIdx.x, gridDim.x) : blockIdx.x;
    const u64 i = (u64)blk * 256 + threadIdx.x;
    if (i < n) { unit16 r = a[i] & b[i]; if (NT) __builtin_nontemporal_store(r, o + i); else o[i] = r; }
}
template <bool XCD, bool NT>
__global__ void __launch_bounds__(256) copy_one(const unit16 *a, unit16 *o, u64 n)
{
    const u32 blk = XCD ? xcd_contig(blockIdx.x, gridDim.x) : blockIdx.x;
    const u64 i = (u64)blk * 256 + threadIdx.x;
    if (i < n) { unit16 r = a[i]; if (NT) __builtin_nontemporal_store(r, o + i); else o[i] = r; }
}

template <typename F>
void bench(const char *name, double bytes, int rounds, F launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    printf("%-36s median %7.1f GB/s  best %7.1f GB/s  (%.3f ms)\n", name, bytes / ts[ts.size() / 2] / 1e6, bytes / ts[0] / 1e6, ts[ts.size() / 2]);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 4.0;
    const int rounds = argc > 2 ? atoi(argv[2]) : 7;
    const u64 n = (u64)(gib * (1ull << 30)) / 16;
    unit16 *a, *b, *o; u32 *sink;
    CK(hipMalloc((void **)&a, n * 16)); CK(hipMalloc((void **)&b, n * 16)); CK(hipMalloc((void **)&o, n * 16)); CK(hipMalloc((void **)&sink, 4096));
    CK(hipMemset(a, 0x11, n * 16)); CK(hipMemset(b, 0x33, n * 16));
    const double bytes = (double)n * 16;
    printf("buffers of %.2f GiB\n", gib);
    {
        const u32 nb = (u32)((n + 255) / 256);
        bench("read_one", bytes, rounds, [&] { read_one<false><<<nb, 256>>>(a, n, sink); });
        bench("read_one xcd-contig", bytes, rounds, [&] { read_one<true><<<nb, 256>>>(a, n, sink); });
        bench("and_one NT", 3 * bytes, rounds, [&] { and_one<false, true><<<nb, 256>>>(a, b, o, n); });
        bench("and_one NT xcd-contig", 3 * bytes, rounds, [&] { and_one<true, true><<<nb, 256>>>(a, b, o, n); });
        bench("copy_one NT", 2 * bytes, rounds, [&] { copy_one<false, true><<<nb, 256>>>(a, o, n); });
        bench("copy_one NT xcd-contig", 2 * bytes, rounds, [&] { copy_one<true, true><<<nb, 256>>>(a, o, n); });
        bench("copy_one xcd-contig", 2 * bytes, rounds, [&] { copy_one<true, false><<<nb, 256>>>(a, o, n); });
    }
    if (argc > 3) return 0;
#define RC(K) bench("read_chunk K=" #K, bytes, rounds, [&] { read_chunk<K><<<(u32)((n + 256 * K - 1) / (256 * K)), 256>>>(a, n, sink); })
    RC(1); RC(2); RC(4); RC(8); RC(10); RC(16); RC(32);
    for (int g : {2048, 8192, 32768}) { char nm[64]; snprintf(nm, 64, "read_stride grid=%d", g); bench(nm, bytes, rounds, [&] { read_stride<<<g, 256>>>(a, n, sink); }); }
#define AC(K, NT) bench("and_chunk K=" #K " NT=" #NT, 3 * bytes, rounds, [&] { and_chunk<K, NT><<<(u32)((n + 256 * K - 1) / (256 * K)), 256>>>(a, b, o, n); })
    AC(1, false); AC(2, false); AC(4, false); AC(8, false); AC(1, true); AC(2, true); AC(4, true); AC(8, true);
#define CC(K, NT) bench("copy_chunk K=" #K " NT=" #NT, 2 * bytes, rounds, [&] { copy_chunk<K, NT><<<(u32)((n + 256 * K - 1) / (256 * K)), 256>>>(a, o, n); })
    CC(1, false); CC(2, false); CC(4, false); CC(8, false); CC(1, true); CC(2, true); CC(4, true); CC(8, true);
    bench("hipMemcpyDtoD", 2 * bytes, rounds, [&] { CK(hipMemcpyAsync(o, a, n * 16, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
