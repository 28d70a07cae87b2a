// wbench.hip -- dev microbenchmark: what write-stream shapes reach on MI355X HBM.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/wbench tools/wbench.hip
// Not part of the product; used to size the all-pairs kernel's store pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

typedef __attribute__((ext_vector_type(4))) unsigned int unit16;
typedef unsigned long long u64;
typedef unsigned int u32;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// A: linear fill, grid-stride
template <bool NT>
__global__ void __launch_bounds__(256) fill_stride(unit16 *o, u64 n, unit16 v)
{
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 s = (u64)gridDim.x * 256;
    for (; i < n; i += s) { if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v; }
}

// B: linear fill, each block owns a contiguous chunk of CH units per thread-column (block writes 256*K units contiguous)
template <int K, bool NT>
__global__ void __launch_bounds__(256) fill_chunk(unit16 *o, u64 n, unit16 v)
{
    u64 base = (u64)blockIdx.x * (256 * K) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        u64 i = base + (u64)k * 256;
        if (i < n) { if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v; }
    }
}

// C: the all-pairs tile pattern with constant data: block = (pair, row_tile, col_tile); BS threads,
// M columns per lane, TI rows; row stride cu units.
template <int M, bool NT>
__global__ void __launch_bounds__(512) fill_tiled(unit16 *o, u32 cu, u32 rows_total, u32 TI, u32 col_tiles, u32 row_tiles, unit16 v)
{
    const u32 BS = blockDim.x;
    const u32 tiles = col_tiles * row_tiles;
    const u32 pair = blockIdx.x / tiles;
    const u32 tile = blockIdx.x - pair * tiles;
    const u32 row_tile = tile / col_tiles, col_tile = tile - row_tile * col_tiles;
    const u32 i0 = row_tile * TI, c0 = col_tile * BS * M;
    unit16 *p = o + (u64)pair * rows_total * cu + (u64)i0 * cu + c0 + threadIdx.x;
    const u32 rows = min(TI, rows_total - i0);
    for (u32 i = 0; i < rows; ++i) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (c0 + m * BS + threadIdx.x < cu) { if (NT) __builtin_nontemporal_store(v, p + m * BS); else p[m * BS] = v; }
        }
        p += cu;
    }
}

// D: row-sequential: block owns TI full rows (contiguous TI*cu units), loops columns inner.
template <bool NT>
__global__ void __launch_bounds__(512) fill_rows(unit16 *o, u32 cu, u32 rows_total, u32 TI, u32 row_tiles, unit16 v)
{
    const u32 BS = blockDim.x;
    const u32 pair = blockIdx.x / row_tiles;
    const u32 row_tile = blockIdx.x - pair * row_tiles;
    const u32 i0 = row_tile * TI;
    const u32 rows = min(TI, rows_total - i0);
    unit16 *p = o + (u64)pair * rows_total * cu + (u64)i0 * cu;
    const u64 n = (u64)rows * cu;
    for (u64 i = threadIdx.x; i < n; i += BS) { if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
}

// F: 4 rows x 4 KiB tile per 256-thread block, but wave w owns ROW w and writes its 4 KiB as four
// consecutive 1 KiB stores (instead of every wave writing 1 KiB of each row)
template <bool NT>
__global__ void __launch_bounds__(256) fill_tile_wave_rows(unit16 *o, u32 cu, u32 rows_total, u32 col_tiles, u32 row_tiles, unit16 v)
{
    const u32 tiles = col_tiles * row_tiles;
    const u32 pair = blockIdx.x / tiles;
    const u32 tile = blockIdx.x - pair * tiles;
    const u32 row_tile = tile / col_tiles, col_tile = tile - row_tile * col_tiles;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 row = row_tile * 4 + wave;
    if (row >= rows_total) return;
    unit16 *p = o + (u64)pair * rows_total * cu + (u64)row * cu + col_tile * 256 + lane;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (col_tile * 256 + k * 64 + lane < cu) { if (NT) __builtin_nontemporal_store(v, p + k * 64); else p[k * 64] = v; }
}

// G: persistent grid-stride writers, but each wave keeps at most W+1 stores in flight
template <int W, bool XCD>
__global__ void __launch_bounds__(256) fill_stride_throttled(unit16 *o, u64 n, unit16 v)
{
    u32 b = blockIdx.x;
    if (XCD) { const u32 q = gridDim.x >> 3, r = gridDim.x & 7u, x = b & 7u; b = x * q + min(x, r) + (b >> 3); }
    // XCD variant: block b of XCD x sweeps the x-th eighth of the buffer with stride = blocks per XCD
    const u64 per = XCD ? (n / 8) : n;
    const u64 base = XCD ? (u64)(blockIdx.x & 7u) * per : 0;
    const u64 nb = XCD ? (gridDim.x >> 3) : gridDim.x;
    const u64 me = XCD ? (blockIdx.x >> 3) : blockIdx.x;
    for (u64 i = me * 256 + threadIdx.x; i < per; i += nb * 256) {
        o[base + i] = v;
        if (W == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (W == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if (W == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    }
}

// E: one store per lane, block size BS, optional XCD-contiguous remap of the chunk order
template <bool NT, int REMAP>
__global__ void __launch_bounds__(1024) fill_one(unit16 *o, u64 n, unit16 v, u32 nblocks)
{
    u32 b = blockIdx.x;
    if (REMAP == 1) { const u32 per = nblocks / 8; b = (b % 8) * per + b / 8; }        // each XCD sweeps its own eighth
    if (REMAP == 2) { const u32 grp = b / 64, r = b % 64; b = grp * 64 + (r % 8) * 8 + r / 8; }  // 8x8 transpose inside 64-block groups
    const u64 i = (u64)b * blockDim.x + threadIdx.x;
    if (i < n) { if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v; }
}

template <typename F>
double bench(const char *name, double bytes, int rounds, F launch)
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
    double med = ts[ts.size() / 2], best = ts[0];
    printf("%-44s median %7.1f GB/s  best %7.1f GB/s  (%.3f ms)\n", name, bytes / med / 1e6, bytes / best / 1e6, med);
    fflush(stdout);
    return med;
}

int main(int argc, char **argv)
{
    const int slots = argc > 1 ? atoi(argv[1]) : 64;
    const int rounds = argc > 2 ? atoi(argv[2]) : 7;
    const u32 T = 1024, U = 10;
    const u32 cu = T * U;                       // units per output row
    const u64 n = (u64)slots * T * cu;          // units in the arena
    const double bytes = (double)n * 16;
    unit16 *o;
    CK(hipMalloc((void **)&o, n * 16));
    unit16 v = {1, 2, 3, 4};
    printf("arena %.2f GiB (%d slots of %ux%u terms)\n", bytes / (1 << 30) / 1.0, slots, T, T);

    bench("hipMemsetAsync", bytes, rounds, [&] { CK(hipMemsetAsync(o, 0x5A, n * 16, 0)); });
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64]; snprintf(nm, 64, "fill_stride grid=%d", g);
        bench(nm, bytes, rounds, [&] { fill_stride<false><<<g, 256>>>(o, n, v); });
    }
    bench("fill_stride grid=4096 NT", bytes, rounds, [&] { fill_stride<true><<<4096, 256>>>(o, n, v); });
    bench("fill_chunk K=1", bytes, rounds, [&] { fill_chunk<1, false><<<(u32)((n + 255) / 256), 256>>>(o, n, v); });
    bench("fill_chunk K=4", bytes, rounds, [&] { fill_chunk<4, false><<<(u32)((n + 1023) / 1024), 256>>>(o, n, v); });
    bench("fill_chunk K=16", bytes, rounds, [&] { fill_chunk<16, false><<<(u32)((n + 4095) / 4096), 256>>>(o, n, v); });
    bench("fill_chunk K=16 NT", bytes, rounds, [&] { fill_chunk<16, true><<<(u32)((n + 4095) / 4096), 256>>>(o, n, v); });
    bench("fill_chunk K=64", bytes, rounds, [&] { fill_chunk<64, false><<<(u32)((n + 16383) / 16384), 256>>>(o, n, v); });

    {
        u32 ct = (cu + 255) / 256, rt = (T + 3) / 4;
        bench("fill_tiled bs=256 M=1 TI=4 (lane cols)", bytes, rounds, [&] { fill_tiled<1, false><<<slots * ct * rt, 256>>>(o, cu, T, 4, ct, rt, v); });
        bench("fill_tiled bs=256 M=1 TI=4 NT", bytes, rounds, [&] { fill_tiled<1, true><<<slots * ct * rt, 256>>>(o, cu, T, 4, ct, rt, v); });
        bench("fill_tile_wave_rows", bytes, rounds, [&] { fill_tile_wave_rows<false><<<slots * ct * rt, 256>>>(o, cu, T, ct, rt, v); });
        bench("fill_tile_wave_rows NT", bytes, rounds, [&] { fill_tile_wave_rows<true><<<slots * ct * rt, 256>>>(o, cu, T, ct, rt, v); });
    }
    for (int g : {2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "fill_stride_thr W=0 grid=%d", g); bench(nm, bytes, rounds, [&] { fill_stride_throttled<0, false><<<g, 256>>>(o, n, v); });
        snprintf(nm, 64, "fill_stride_thr W=1 grid=%d", g); bench(nm, bytes, rounds, [&] { fill_stride_throttled<1, false><<<g, 256>>>(o, n, v); });
        snprintf(nm, 64, "fill_stride_thr W=3 grid=%d", g); bench(nm, bytes, rounds, [&] { fill_stride_throttled<3, false><<<g, 256>>>(o, n, v); });
        snprintf(nm, 64, "fill_stride_thr W=0 xcd grid=%d", g); bench(nm, bytes, rounds, [&] { fill_stride_throttled<0, true><<<g, 256>>>(o, n, v); });
        snprintf(nm, 64, "fill_stride_thr W=1 xcd grid=%d", g); bench(nm, bytes, rounds, [&] { fill_stride_throttled<1, true><<<g, 256>>>(o, n, v); });
    }
    for (u32 bs : {256u, 512u, 1024u}) {
        const u32 nb = (u32)(n / bs);
        char nm[64];
        snprintf(nm, 64, "fill_one bs=%u", bs); bench(nm, bytes, rounds, [&] { fill_one<false, 0><<<nb, bs>>>(o, n, v, nb); });
        snprintf(nm, 64, "fill_one bs=%u NT", bs); bench(nm, bytes, rounds, [&] { fill_one<true, 0><<<nb, bs>>>(o, n, v, nb); });
        snprintf(nm, 64, "fill_one bs=%u xcd-contig", bs); bench(nm, bytes, rounds, [&] { fill_one<false, 1><<<nb, bs>>>(o, n, v, nb); });
        snprintf(nm, 64, "fill_one bs=%u NT xcd-contig", bs); bench(nm, bytes, rounds, [&] { fill_one<true, 1><<<nb, bs>>>(o, n, v, nb); });
        snprintf(nm, 64, "fill_one bs=%u 8x8-transpose", bs); bench(nm, bytes, rounds, [&] { fill_one<false, 2><<<nb, bs>>>(o, n, v, nb); });
    }
    if (argc > 3 && argv[3][0] == 'e') { CK(hipFree(o)); return 0; }
    for (u32 bs : {64u, 128u, 256u, 320u, 512u}) {
        for (u32 ti : {1u, 2u, 4u, 8u}) {
            const u32 M = 1; u32 ct = (cu + bs * M - 1) / (bs * M), rt = (T + ti - 1) / ti;
            char nm[64]; snprintf(nm, 64, "fill_tiled bs=%u M=1 TI=%u", bs, ti);
            bench(nm, bytes, rounds, [&] { fill_tiled<1, false><<<slots * ct * rt, bs>>>(o, cu, T, ti, ct, rt, v); });
        }
    }
    if (argc > 3) { CK(hipFree(o)); return 0; }
    for (u32 bs : {256u, 320u, 512u, 640u}) {
        for (u32 ti : {16u, 64u, 256u}) {
            {
                const u32 M = 4; u32 ct = (cu + bs * M - 1) / (bs * M), rt = (T + ti - 1) / ti;
                char nm[64]; snprintf(nm, 64, "fill_tiled bs=%u M=4 TI=%u", bs, ti);
                if (bs <= 512) bench(nm, bytes, rounds, [&] { fill_tiled<4, false><<<slots * ct * rt, bs>>>(o, cu, T, ti, ct, rt, v); });
            }
            {
                const u32 M = 1; u32 ct = (cu + bs * M - 1) / (bs * M), rt = (T + ti - 1) / ti;
                char nm[64]; snprintf(nm, 64, "fill_tiled bs=%u M=1 TI=%u", bs, ti);
                if (bs <= 512) bench(nm, bytes, rounds, [&] { fill_tiled<1, false><<<slots * ct * rt, bs>>>(o, cu, T, ti, ct, rt, v); });
            }
        }
    }
    for (u32 bs : {256u, 512u}) {
        for (u32 ti : {1u, 4u, 16u, 64u}) {
            u32 rt = (T + ti - 1) / ti;
            char nm[64]; snprintf(nm, 64, "fill_rows bs=%u TI=%u", bs, ti);
            bench(nm, bytes, rounds, [&] { fill_rows<false><<<slots * rt, bs>>>(o, cu, T, ti, rt, v); });
            snprintf(nm, 64, "fill_rows bs=%u TI=%u NT", bs, ti);
            bench(nm, bytes, rounds, [&] { fill_rows<true><<<slots * rt, bs>>>(o, cu, T, ti, rt, v); });
        }
    }
    CK(hipFree(o));
    return 0;
}
