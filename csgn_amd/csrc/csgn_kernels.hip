// csgn_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the certFHE/CSGN
// ciphertext-arithmetic hot path.  Everything here is bitwise integer work bound by HBM
// bandwidth: there is no MFMA anywhere.  Design notes live in DESIGN.md; reference
// citations (/root/reference/...) name the scalar loop each kernel replaces.
//
// Common shape of the data path: lanes own consecutive 16-byte units so every wave-level
// load/store is one global_{load,store}_dwordx4 covering 1 KiB of contiguous, 128-B-aligned
// HBM; anything that is re-used across a tile (the opposing operand's terms, the key mask)
// is staged once in LDS and read back with ds_read_b128.
#include "csgn_kernels.h"

#include <cstdlib>

namespace csgn {

namespace {

constexpr u32 kWave = 64;

template <typename Unit, bool NT>
__device__ inline void unit_store(Unit *p, Unit v)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// true iff every mask bit is set in x
__device__ inline bool unit_covers(unit16 x, unit16 m)
{
    unit16 d = (x & m) ^ m;
    return (d.x | d.y | d.z | d.w) == 0u;
}
__device__ inline bool unit_covers(unit8 x, unit8 m) { return (x & m) == m; }

inline u32 ceil_div_u64(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

// HIP refuses a launch whose gridDim.x * blockDim.x reaches 2^32 (hipErrorInvalidConfiguration),
// so the number of workgroups per launch is bounded by the block size, not by 2^31.
constexpr u64 kMaxBlocks256 = ((1ull << 32) - 1) / 256;    // 256-thread workgroups
constexpr u64 kMaxBlocks512 = ((1ull << 32) - 1) / 512;    // the tiled kernel's upper block size
constexpr u64 kMaxBlocks1024 = ((1ull << 32) - 1) / 1024;

// Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8; placement is
// a speed matter only, never correctness).  Remapping the block id so that XCD x owns the
// x-th contiguous eighth of the logical block range makes every XCD's L2 write back one
// sequential address stream instead of every 8th 4 KiB chunk: +5 % on a pure fill
// (tools/wbench.hip: 6.9 -> 7.3 TB/s).  Bijective for any grid size.
__device__ inline u32 xcd_contiguous_block(u32 b, u32 nblocks)
{
    const u32 q = nblocks >> 3, r = nblocks & 7u, x = b & 7u;
    return x * q + min(x, r) + (b >> 3);
}

// ---------------------------------------------------------------------------------------
// 1x1 batch: out = a & b over a flat stream of units.
// Replaces Ciphertext::defaultN_multiply (src/Ciphertext.cpp:124-131) for a whole batch of
// fresh ciphertext pairs (BASELINE configs 2 and 4): 3 x 16 B of HBM traffic per unit.
// ---------------------------------------------------------------------------------------
// One 16-byte unit per lane and one 4 KiB segment per short-lived workgroup: measured
// (tools/rbench.hip) this beats every deeper-unrolled or grid-stride form on MI355X because the
// chip-wide access front stays dense in address space (6.1 vs 5.3 TB/s).
template <typename Unit, bool NT>
__global__ void __launch_bounds__(256) k_and_stream(const Unit *__restrict__ a,
                                                    const Unit *__restrict__ b,
                                                    Unit *__restrict__ o, u64 n_units)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n_units)
        unit_store<Unit, NT>(o + i, a[i] & b[i]);
}

// ---------------------------------------------------------------------------------------
// Small uniform shapes (t1*t2*U below one tile): flat map from output unit to
// (pair, left term i, right column c).  Operands are tiny and re-read through L1/L2.
// Replaces the general path of Ciphertext::multiply (src/Ciphertext.cpp:146-163).
// ---------------------------------------------------------------------------------------
template <typename Unit, int MF, bool XCD>
__global__ void __launch_bounds__(256) k_mul_flat(const Unit *__restrict__ L,
                                                  const Unit *__restrict__ R,
                                                  Unit *__restrict__ out, u32 total_units, u32 t1,
                                                  u32 t2, u32 U, FastDiv dPU, FastDiv dCU, FastDiv dU,
                                                  u32 pf_rows, u32 total_rows)
{
    // One launch covers < 2^32 output units, so every index below is 32-bit.  Loads are
    // unconditional on clamped indices so that all 2*MF of them are in flight together.
    //
    // Left-term prefetch: every output row needs a NEW 16*U-byte left term, and with operands
    // streaming from HBM each of the row's workgroups would sit out a full HBM miss on it
    // (measured: 4.6 TB/s instead of 7.4).  The lanes that own the first U units of a row
    // therefore also touch the left term of the row `pf_rows` further down the launch (left
    // operands of consecutive pairs are contiguous, so this runs across pair boundaries); by
    // the time that row is dispatched its term sits in L2 / Infinity Cache.  The value is only
    // kept alive, never used.
    const u32 CU = t2 * U, LU = t1 * U, PU = t1 * CU;
    const u32 bid = XCD ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const u32 last = total_units - 1;
    Unit l[MF], r[MF], pf_val;
    bool pf_on = false;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
        const u32 g = min(bid * (256u * MF) + (u32)m * 256u + threadIdx.x, last);
        const u32 pair = csgn_fastdiv(g, dPU);
        const u32 rr = g - pair * PU;
        const u32 i = csgn_fastdiv(rr, dCU);
        const u32 c = rr - i * CU;
        const u32 k = c - csgn_fastdiv(c, dU) * U;
        l[m] = L[(u64)pair * LU + i * U + k];
        r[m] = R[(u64)pair * CU + c];
        if (m == 0 && pf_rows) {
            const u32 grow = pair * t1 + i + pf_rows;           // global row of this launch
            if (c < U && grow < total_rows) {
                pf_val = L[(u64)grow * U + c];
                pf_on = true;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MF; ++m) {
        const u32 g = bid * (256u * MF) + (u32)m * 256u + threadIdx.x;
        if (g <= last)
            unit_store<Unit, true>(out + g, l[m] & r[m]);
    }
    if (pf_on)
        asm volatile("" ::"v"(pf_val));
}

// ---------------------------------------------------------------------------------------
// All-pairs multiply, LDS-tiled.  Replaces Ciphertext::multiply's general path
// (src/Ciphertext.cpp:146-163):  out[(i*t2 + j)*dL + k] = L[i*dL + k] & R[j*dL + k].
//
// For a fixed left term i the output row is the WHOLE right operand masked by one
// broadcast term, so the product is a pure streaming write (reads are < 0.3 % of the
// bytes at 1024x1024).  A workgroup owns TI left terms x (BS*M) right-operand units:
//   - the TI left terms sit in LDS (TI*U units: 640 B at N=1247 with the default TI=4);
//   - each lane keeps its M right units in registers for the whole tile;
//   - per row, a lane reads the one left unit it needs (index c mod U) with ds_read_b128
//     and issues M global_store_dwordx4; a wave instruction writes 1 KiB contiguous.
// Defaults (mul_tuning): 256 threads, M = 1 for 16-byte units (2 for 8-byte units), TI = 4.
// SAMEK only matters for M > 1: when U divides the block size every one of a lane's M columns
// needs the same left unit, so there is one LDS read per row instead of M.
// ---------------------------------------------------------------------------------------
struct MulArgs {
    const void *L;
    const void *R;
    void *out;
    const u64 *offL;
    const u64 *offR;
    const u64 *offOut;
    u32 t1, t2, U, TI, col_tiles, row_tiles;
    u32 xcd_remap;
    u32 pair_base;      // ragged launches cut into chunks: first pair of this launch
};

template <typename Unit, int M, bool SAMEK, bool RAGGED, bool NT>
__global__ void __launch_bounds__(512) k_mul_tiled(MulArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Unit *lds = reinterpret_cast<Unit *>(smem_raw);

    const u32 BS = blockDim.x, tid = threadIdx.x, U = a.U;
    const u32 tiles = a.col_tiles * a.row_tiles;
    const u32 bid = a.xcd_remap ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const u32 pair_local = bid / tiles;
    const u32 tile = bid - pair_local * tiles;
    const u32 pair = a.pair_base + pair_local;
    const u32 row_tile = tile / a.col_tiles;
    const u32 col_tile = tile - row_tile * a.col_tiles;

    u32 t1, t2;
    u64 lbase, rbase, obase;   // in units
    if (RAGGED) {
        const u64 l0 = a.offL[pair], r0 = a.offR[pair];
        t1 = (u32)(a.offL[pair + 1] - l0);
        t2 = (u32)(a.offR[pair + 1] - r0);
        lbase = l0 * U;
        rbase = r0 * U;
        obase = a.offOut[pair] * U;
    } else {
        t1 = a.t1;
        t2 = a.t2;
        lbase = (u64)pair * t1 * U;
        rbase = (u64)pair * t2 * U;
        obase = (u64)pair * t1 * t2 * U;
    }
    const u32 cu = t2 * U;
    const u32 i0 = row_tile * a.TI;
    const u32 c0 = col_tile * BS * M;
    if (i0 >= t1 || c0 >= cu)
        return;                               // whole workgroup leaves together
    const u32 rows = min(a.TI, t1 - i0);

    // stage the left tile: rows*U consecutive units, coalesced (wave 0 only at the default tile)
    const Unit *Lp = reinterpret_cast<const Unit *>(a.L) + lbase + (u64)i0 * U;
    for (u32 u = tid; u < rows * U; u += BS)
        lds[u] = Lp[u];

    // this lane's right-operand units stay in registers for the whole tile.  (Measured A/B on
    // one device, bench.py batch 8192: issuing this load first and unconditionally, so that it
    // overlaps the left-tile fetch, is 4-5 % SLOWER end to end, 39.1 k vs 41.5 k mult/s; the
    // 8 resident workgroups per CU already hide the prologue latency.)
    const Unit *Rp = reinterpret_cast<const Unit *>(a.R) + rbase;
    Unit r[M];
    u32 k[M];
    bool valid[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const u32 c = c0 + (u32)m * BS + tid;
        valid[m] = c < cu;
        if (valid[m])
            r[m] = Rp[c];
        k[m] = c % U;
    }
    __syncthreads();

    Unit *orow = reinterpret_cast<Unit *>(a.out) + obase + (u64)i0 * cu + c0 + tid;
#pragma unroll 2
    for (u32 i = 0; i < rows; ++i) {
        if (SAMEK) {
            const Unit l = lds[i * U + k[0]];
#pragma unroll
            for (int m = 0; m < M; ++m)
                if (valid[m])
                    unit_store<Unit, NT>(orow + (u32)m * BS, r[m] & l);
        } else {
#pragma unroll
            for (int m = 0; m < M; ++m)
                if (valid[m])
                    unit_store<Unit, NT>(orow + (u32)m * BS, r[m] & lds[i * U + k[m]]);
        }
        orow += cu;
    }
}

// ---------------------------------------------------------------------------------------
// Ragged batches (CSR offsets), skew-proof form.  The grid covers the FLATTENED output
// (one 16-byte unit per lane, 4 KiB per workgroup, address order), so its size is the real
// output, not batch x the largest shape.  A lane finds the pair that owns its output term by
// binary search over the product offsets: first the workgroup's starting pair (identical
// addresses in every lane, so the loads broadcast), then a short per-lane search inside the
// few pairs one workgroup can span.
// ---------------------------------------------------------------------------------------
// largest p in [lo, hi) with off[p] <= term   (requires off[lo] <= term)
__device__ inline u32 csr_find(const u64 *__restrict__ off, u32 lo, u32 hi, u64 term)
{
    while (hi - lo > 1) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (off[mid] <= term)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// largest p >= lo with off[p] <= term, looking near lo first (requires off[lo] <= term): the
// common answers are lo itself (one load) or a pair a few steps on
__device__ inline u32 csr_gallop(const u64 *__restrict__ off, u32 lo, u32 batch, u64 term)
{
    u32 hi = lo + 1u, step = 1u;
    while (hi < batch && off[hi] <= term) {
        lo = hi;
        step <<= 1;
        hi = (batch - lo > step) ? lo + step : batch;
    }
    return csr_find(off, lo, hi, term);
}

// A workgroup owns C consecutive 4 KiB chunks of the flattened output: one uniform search for its
// first term, then every wave walks forward from the pair it was in (csr_gallop), so the
// log2(batch) dependent loads are paid once per C chunks instead of twice per chunk.
template <typename Unit, int C>
__global__ void __launch_bounds__(256) k_mul_ragged_flat(const Unit *__restrict__ L,
                                                         const u64 *__restrict__ offL,
                                                         const Unit *__restrict__ R,
                                                         const u64 *__restrict__ offR,
                                                         Unit *__restrict__ out,
                                                         const u64 *__restrict__ offOut, u32 batch,
                                                         u64 unit_base, u64 total_units, u32 U, FastDiv dU,
                                                         u32 pf_pairs)
{
    const u32 bid = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const u64 g_begin = unit_base + (u64)bid * (256u * C);
    if (g_begin >= total_units)
        return;
    const u64 term0 = g_begin / U;                              // workgroup-uniform
    const u32 r0blk = (u32)(g_begin - term0 * U);
    u32 pw = csr_find(offOut, 0u, batch, term0);                // uniform search: loads broadcast
    // The workgroup that holds the start of a pair pulls the operands of the pair `pf_pairs` further
    // on into the caches (one dword per 128-byte line, values unused): by the time that pair's
    // rows are written its left terms are hits instead of HBM misses under full write load.
    if (pf_pairs && term0 - offOut[pw] < (256u * C) / U + 1u) {
        const u32 pt = min(pw + pf_pairs, batch - 1u);
        const u64 lb = offL[pt] * U * sizeof(Unit), le = offL[pt + 1] * U * sizeof(Unit);
        const u64 rb = offR[pt] * U * sizeof(Unit), re = offR[pt + 1] * U * sizeof(Unit);
        const char *Lb = reinterpret_cast<const char *>(L), *Rb = reinterpret_cast<const char *>(R);
#pragma unroll 1
        for (u64 a = lb + (u64)threadIdx.x * 128u, n = 0; a < le && n < 8; a += 256u * 128u, ++n) {
            const u32 v = *reinterpret_cast<const u32 *>(Lb + a);
            asm volatile("" ::"v"(v));
        }
#pragma unroll 1
        for (u64 a = rb + (u64)threadIdx.x * 128u, n = 0; a < re && n < 8; a += 256u * 128u, ++n) {
            const u32 v = *reinterpret_cast<const u32 *>(Rb + a);
            asm volatile("" ::"v"(v));
        }
    }
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
        const u64 g = g_begin + (u32)c * 256u + threadIdx.x;
        if (g_begin + (u32)c * 256u >= total_units)
            break;
        // this lane's term: a 32-bit division of its distance from the workgroup's first term
        const u32 r = r0blk + (u32)c * 256u + threadIdx.x;
        const u32 dt = csgn_fastdiv(r, dU);
        const u64 term = term0 + dt;
        const u32 k = r - dt * U;
        u32 p = pw;
        if (g < total_units) {
            p = csr_gallop(offOut, pw, batch, term);            // runs of empty pairs are walked over
            const u64 l0 = offL[p], rr0 = offR[p];
            const u32 t2 = (u32)(offR[p + 1] - rr0);
            const u32 q = (u32)(term - offOut[p]);              // product term index inside the pair
            const u32 i = q / t2, j = q - i * t2;
            unit_store<Unit, true>(out + g, L[(l0 + i) * U + k] & R[(rr0 + j) * U + k]);
        }
        pw = (u32)__builtin_amdgcn_readfirstlane((int)p);       // later chunks start from here
    }
}

template <typename Unit, int C>
__global__ void __launch_bounds__(256) k_add_ragged_flat(const Unit *__restrict__ L,
                                                         const u64 *__restrict__ offL,
                                                         const Unit *__restrict__ R,
                                                         const u64 *__restrict__ offR,
                                                         Unit *__restrict__ out,
                                                         const u64 *__restrict__ offOut, u32 batch,
                                                         u64 unit_base, u64 total_units, u32 U, FastDiv dU)
{
    const u32 bid = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const u64 g_begin = unit_base + (u64)bid * (256u * C);
    if (g_begin >= total_units)
        return;
    const u64 term0 = g_begin / U;
    const u32 r0blk = (u32)(g_begin - term0 * U);
    u32 pw = csr_find(offOut, 0u, batch, term0);
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
        const u64 g = g_begin + (u32)c * 256u + threadIdx.x;
        if (g_begin + (u32)c * 256u >= total_units)
            break;
        const u32 r = r0blk + (u32)c * 256u + threadIdx.x;
        const u32 dt = csgn_fastdiv(r, dU);
        const u64 term = term0 + dt;
        const u32 k = r - dt * U;
        u32 p = pw;
        if (g < total_units) {
            p = csr_gallop(offOut, pw, batch, term);
            const u64 l0 = offL[p], rr0 = offR[p];
            const u64 t1 = offL[p + 1] - l0;
            const u64 q = term - offOut[p];                     // offOut[p] = l0 + rr0
            const Unit v = (q < t1) ? L[(l0 + q) * U + k] : R[(rr0 + (q - t1)) * U + k];
            unit_store<Unit, true>(out + g, v);
        }
        pw = (u32)__builtin_amdgcn_readfirstlane((int)p);
    }
}

// Product term offsets = exclusive scan of t1_b*t2_b over the batch, plus the shape maxima the
// launcher needs.  Three small kernels: per-1024-pair chunk scans, a scan of the chunk totals,
// and the fix-up -- 1M pairs plan in tens of microseconds.
__global__ void __launch_bounds__(256) k_plan_chunks(u64 batch, const u64 *__restrict__ offL,
                                                     const u64 *__restrict__ offR,
                                                     u64 *__restrict__ offOut, u64 *__restrict__ partial,
                                                     u64 *__restrict__ plan4)
{
    __shared__ u64 sums[256];
    const u32 tid = threadIdx.x;
    const u64 b0 = (u64)blockIdx.x * 1024u + (u64)tid * 4u;
    u64 c[4], m1 = 0, m2 = 0, mp = 0, mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = 0;
        if (b0 + j < batch) {
            const u64 t1 = offL[b0 + j + 1] - offL[b0 + j], t2 = offR[b0 + j + 1] - offR[b0 + j];
            c[j] = t1 * t2;
            m1 = max(m1, t1);
            m2 = max(m2, t2);
            mp = max(mp, c[j]);
        }
        mine += c[j];
    }
    sums[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        u64 run = 0;
        for (u32 t = 0; t < 256; ++t) {
            const u64 v = sums[t];
            sums[t] = run;
            run += v;
        }
        partial[blockIdx.x] = run;
    }
    __syncthreads();
    u64 run = sums[tid];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (b0 + j < batch)
            offOut[b0 + j] = run;                      // chunk-local; k_plan_fix adds the chunk base
        run += c[j];
    }
    // wave-level maxima, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        m1 = max(m1, (u64)__shfl_down(m1, off, 64));
        m2 = max(m2, (u64)__shfl_down(m2, off, 64));
        mp = max(mp, (u64)__shfl_down(mp, off, 64));
    }
    if ((tid & (kWave - 1)) == 0) {
        atomicMax(reinterpret_cast<unsigned long long *>(plan4 + 1), m1);
        atomicMax(reinterpret_cast<unsigned long long *>(plan4 + 2), m2);
        atomicMax(reinterpret_cast<unsigned long long *>(plan4 + 3), mp);
    }
}

__global__ void __launch_bounds__(1024) k_plan_scan_partials(u64 nchunks, u64 batch, u64 *__restrict__ partial,
                                                             u64 *__restrict__ offOut, u64 *__restrict__ plan4)
{
    __shared__ u64 part[1024];
    const u32 tid = threadIdx.x;
    const u64 chunk = (nchunks + 1023) / 1024;
    const u64 c0 = min(nchunks, (u64)tid * chunk), c1 = min(nchunks, c0 + chunk);
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
        offOut[batch] = run;
        plan4[0] = run;
    }
    __syncthreads();
    u64 run = part[tid];
    for (u64 c = c0; c < c1; ++c) {
        const u64 v = partial[c];
        partial[c] = run;
        run += v;
    }
}

__global__ void __launch_bounds__(256) k_plan_fix(u64 batch, const u64 *__restrict__ partial,
                                                  u64 *__restrict__ offOut)
{
    const u64 b = (u64)blockIdx.x * 256u + threadIdx.x;
    if (b < batch)
        offOut[b] += partial[b >> 10];
}

// ---------------------------------------------------------------------------------------
// add = concatenation (src/Ciphertext.cpp:107-122).  Flat map over output units.
// ---------------------------------------------------------------------------------------
template <typename Unit, bool NT>
__global__ void __launch_bounds__(256) k_add_flat(const Unit *__restrict__ L,
                                                  const Unit *__restrict__ R,
                                                  Unit *__restrict__ out, u32 total_units, u32 LU,
                                                  u32 RU, FastDiv dOU)
{
    // one unit per lane, < 2^32 units per launch (see k_and_stream for why)
    const u32 OU = LU + RU;
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    if (g < total_units) {
        const u32 pair = csgn_fastdiv(g, dOU);
        const u32 r = g - pair * OU;
        const Unit v = (r < LU) ? L[(u64)pair * LU + r] : R[(u64)pair * RU + (r - LU)];
        unit_store<Unit, NT>(out + g, v);
    }
}

__global__ void __launch_bounds__(256) k_off_sum(u64 n, const u64 *__restrict__ a,
                                                 const u64 *__restrict__ b, u64 *__restrict__ o)
{
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        o[i] = a[i] + b[i];
}

// ---------------------------------------------------------------------------------------
// decrypt, pass 1: one hit bit per term.  Replaces the unpack-everything loops of
// SecretKey::decrypt (src/SecretKey.cpp:110-137): a term "hits" iff all D secret positions
// are 1, i.e. (term & mask) == mask over its dL words.  A workgroup streams 256 consecutive
// terms with coalesced 16-B loads; a unit that misses the mask flags its term in LDS; the
// 256 verdicts leave as four __ballot words, so the bitmap needs no atomics.
// ---------------------------------------------------------------------------------------
template <typename Unit>
__global__ void __launch_bounds__(256) k_term_hits(const Unit *__restrict__ terms,
                                                   const Unit *__restrict__ mask, u64 total_terms,
                                                   u32 U, FastDiv dU, u64 *__restrict__ hits)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Unit *lmask = reinterpret_cast<Unit *>(smem_raw);                  // U units
    u32 *fail = reinterpret_cast<u32 *>(smem_raw + (size_t)U * sizeof(Unit));   // 256 flags

    const u32 tid = threadIdx.x;
    const u64 term0 = (u64)blockIdx.x * 256u;
    const u32 nterms = (u32)min((u64)256, total_terms - term0);
    const u32 nunits = nterms * U;
    for (u32 k = tid; k < U; k += 256u)
        lmask[k] = mask[k];
    fail[tid] = 0;
    __syncthreads();

    const Unit *base = terms + term0 * U;
#pragma unroll 4
    for (u32 u = tid; u < nunits; u += 256u) {
        const Unit x = base[u];
        const u32 t = csgn_fastdiv(u, dU);
        const u32 k = u - t * U;
        if (!unit_covers(x, lmask[k]))
            fail[t] = 1;                        // benign race: every writer stores 1
    }
    __syncthreads();

    const bool hit = (tid < nterms) && (fail[tid] == 0);
    const u64 b = __ballot(hit);
    if ((tid & (kWave - 1)) == 0)
        hits[(u64)blockIdx.x * 4u + (tid >> 6)] = b;
}

// decrypt, pass 1, fast form.  A 256-thread workgroup makes K passes over K consecutive
// 4 KiB segments (K*256 units = TB whole terms, TB a multiple of 8), all K loads of a lane in
// flight at once.  Every wave ballots "my unit covers the mask" per pass; the K*4 ballots form
// a bit string in LDS in which term t owns bits [t*U, t*U+U); lane t < TB tests them and the
// TB verdicts leave as TB/8 bytes of the hit bitmap.  Segments stay 4 KiB-aligned whatever U
// is (320-thread / 5 KiB workgroups measured 15 % slower at N=1247), and workgroups stay
// short-lived and in address order (see k_and_stream).
template <typename Unit, int K>
__global__ void __launch_bounds__(256) k_term_hits_seg(const Unit *__restrict__ terms,
                                                       const Unit *__restrict__ mask,
                                                       u64 total_units, u32 U, FastDiv dU, u32 TB,
                                                       unsigned char *__restrict__ hits,
                                                       uint8_t *__restrict__ direct_bits)
{
    __shared__ u64 ok_bits[K * 4 + 1];
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    // XCD-contiguous block order: +4.5 % on a pure read stream (tools/rbench.hip)
    const u32 bid = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const u64 g0 = (u64)bid * (256u * K) + tid;

    // Loads are unconditional (addresses clamped into range) so that all K of them, and the K
    // mask units, are in flight together: a load under a divergent `if` makes hipcc wait
    // vmcnt(0) right behind it.
    Unit x[K], mk[K];
    const u64 last = total_units - 1;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const u64 g = g0 + (u32)j * 256u;
        x[j] = terms[g < last ? g : last];
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const u32 local = (u32)j * 256u + tid;
        mk[j] = mask[local - csgn_fastdiv(local, dU) * U];
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const bool ok = (g0 + (u32)j * 256u <= last) && unit_covers(x[j], mk[j]);
        const u64 b = __ballot(ok);
        if (lane == 0)
            ok_bits[j * 4 + wave] = b;
    }
    if (tid == 0)
        ok_bits[K * 4] = 0;                       // pad word for the straddling shift below
    __syncthreads();

    const u64 need = (U >= 64u) ? ~0ull : ((1ull << U) - 1ull);
    for (u32 t = tid; t < TB; t += 256u) {        // TB <= 256 unless U == 1 (TB = 256*K)
        const u32 start = t * U, w = start >> 6, sh = start & 63u;
        u64 v = ok_bits[w] >> sh;
        if (sh)
            v |= ok_bits[w + 1] << (64u - sh);
        const bool hit = (v & need) == need;
        if (direct_bits) {                        // single-term ciphertexts: the verdict IS the plaintext
            const u64 term = (u64)bid * TB + t;
            if (term * U < total_units)
                direct_bits[term] = hit ? 1 : 0;
            continue;
        }
        const u64 hb = __ballot(hit);             // lanes past TB are not in this iteration
        if (lane == 0) {
            const u32 first = t;                  // first term of this wave's group
            const u32 nbits = min(64u, TB - first);
            unsigned char *dst = hits + (u64)bid * (TB >> 3) + (first >> 3);
            if (nbits == 64u)
                *reinterpret_cast<u64 *>(dst) = hb;
            else if (nbits == 32u)
                *reinterpret_cast<u32 *>(dst) = (u32)hb;
            else if (nbits == 16u)
                *reinterpret_cast<unsigned short *>(dst) = (unsigned short)hb;
            else
                *dst = (unsigned char)hb;
        }
    }
}

// decrypt, pass 2: XOR over the terms of each ciphertext = parity of the popcount of its
// bit range (src/SecretKey.cpp:139, `_dec = (dec + _dec) % 2`).  G lanes per ciphertext:
// 1 for small term counts, a whole wave (with a __ballot/__popcll fold) for large ones.
// MODE 0: every ciphertext; 1: only those of at most kLongTerms terms; 2: only the longer ones
// (ragged batches run a lane-per-ciphertext pass for the short ones and a wave-per-ciphertext
// pass for the long ones, so one huge ciphertext among many small ones costs neither).
constexpr u64 kLongTerms = 4096;

template <int G, int MODE>
__global__ void __launch_bounds__(256) k_hits_parity(const u64 *__restrict__ hits,
                                                     const u64 *__restrict__ off, u64 T, u64 batch,
                                                     uint8_t *__restrict__ bits)
{
    const u64 gid = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 b = gid / G;
    const u32 lane = (u32)(gid % G);
    if (b >= batch)
        return;
    const u64 s = off ? off[b] : b * T;
    const u64 e = off ? off[b + 1] : s + T;
    if (MODE == 1 && e - s > kLongTerms)
        return;
    if (MODE == 2 && e - s <= kLongTerms)
        return;                                     // whole wave leaves (G == 64: one ciphertext per wave)
    u32 par = 0;
    if (e > s) {
        const u64 w0 = s >> 6, w1 = (e - 1) >> 6;
        for (u64 w = w0 + lane; w <= w1; w += G) {
            u64 x = hits[w];
            if (w == w0)
                x &= ~0ull << (s & 63);
            if (w == w1 && (e & 63))
                x &= (1ull << (e & 63)) - 1;
            par ^= (u32)__popcll(x);
        }
    }
    if (G == 1) {
        bits[b] = (uint8_t)(par & 1u);
    } else {
        const u64 odd = __ballot(par & 1u);
        if (lane == 0)
            bits[b] = (uint8_t)(__popcll(odd) & 1);
    }
}

// decrypt, pass 2 for LONG uniform ciphertexts: a ciphertext's bit range is cut into chunks of
// 65536 terms (1024 bitmap words); one workgroup per (ciphertext, chunk) folds its chunk with
// __popcll / __ballot and XORs one bit into a per-ciphertext word, so a single 1M-term
// ciphertext is reduced by 16 workgroups instead of one wave.
__global__ void __launch_bounds__(256) k_hits_parity_chunked(const u64 *__restrict__ hits, u64 T,
                                                             u32 chunks, u32 *__restrict__ partial)
{
    __shared__ u32 wave_par[4];
    const u32 b = blockIdx.x / chunks, c = blockIdx.x - b * chunks;
    const u64 s = (u64)b * T, e = s + T;
    const u64 cs = s + (u64)c * 65536u;
    const u64 ce = min(e, cs + 65536u);
    u32 par = 0;
    if (ce > cs) {
        const u64 w0 = cs >> 6, w1 = (ce - 1) >> 6;
        for (u64 w = w0 + threadIdx.x; w <= w1; w += 256u) {
            u64 x = hits[w];
            if (w == w0)
                x &= ~0ull << (cs & 63);
            if (w == w1 && (ce & 63))
                x &= (1ull << (ce & 63)) - 1;
            par ^= (u32)__popcll(x);
        }
    }
    const u64 odd = __ballot(par & 1u);
    if ((threadIdx.x & (kWave - 1)) == 0)
        wave_par[threadIdx.x >> 6] = (u32)__popcll(odd) & 1u;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 p = wave_par[0] ^ wave_par[1] ^ wave_par[2] ^ wave_par[3];
        if (p)
            atomicXor(partial + b, 1u);
    }
}

__global__ void __launch_bounds__(256) k_partial_to_bits(const u32 *__restrict__ partial, u64 batch,
                                                         uint8_t *__restrict__ bits)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < batch)
        bits[i] = (uint8_t)(partial[i] & 1u);
}

// ---------------------------------------------------------------------------------------
// encrypt.  Replaces SecretKey::encrypt (bit vector src/SecretKey.cpp:35-80, packing
// :175-197) with the per-position randomness supplied packed (or generated in place).
// A workgroup builds CB ciphertexts in LDS: load/generate the random words, one lane per
// ciphertext applies the plaintext-0 rule (pick one secret slot; clear it iff every OTHER
// secret slot came out 1, else give it the spare random bit), then all lanes write the
// tile out coalesced, OR-ing the key mask into plaintext-1 ciphertexts.
// ---------------------------------------------------------------------------------------
// Device-RNG draws of one ciphertext with plaintext 0 (throughput mode only): which of the D secret
// positions is the chosen one (src/SecretKey.cpp:51) and the spare coin of :76.  One splitmix64
// word: the high half picks the position by multiply-shift range reduction (no 64-bit modulo), the
// low bit is the coin.
__device__ inline void enc_draw(u64 seed, u64 c, const u64 *__restrict__ key, u64 D, u32 &pos, u32 &spare)
{
    const u64 r = csgn_splitmix64((seed ^ 0xD1B54A32D192ED03ull) + CSGN_GOLDEN * (c + 1));
    pos = (u32)key[__umulhi((u32)(r >> 32), (u32)D)];
    spare = (u32)r & 1u;
}

template <bool DEVRNG>
__global__ void __launch_bounds__(256) k_encrypt(u64 n_bits, u32 dL, u64 D, u64 batch, u32 CB,
                                                 FastDiv ddL, const uint8_t *__restrict__ plain,
                                                 const u64 *__restrict__ rnd,
                                                 const u32 *__restrict__ chosen,
                                                 const uint8_t *__restrict__ last,
                                                 const u64 *__restrict__ key,
                                                 const u64 *__restrict__ mask, u64 seed,
                                                 u64 *__restrict__ out)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    u64 *tile = reinterpret_cast<u64 *>(smem_raw);     // CB*dL words
    u64 *lmask = tile + (size_t)CB * dL;               // dL words

    const u32 tid = threadIdx.x;
    const u64 c0 = (u64)blockIdx.x * CB;
    const u32 nc = (u32)min((u64)CB, batch - c0);
    const u32 nw = nc * dL;
    const u32 rem = (u32)(n_bits & 63);
    const u64 tail = rem ? ~0ull << (64 - rem) : ~0ull;

    for (u32 u = tid; u < nw; u += 256u) {
        const u32 c = csgn_fastdiv(u, ddL);
        const u32 k = u - c * dL;
        const u64 gw = c0 * dL + u;
        u64 w = DEVRNG ? csgn_rng_word(seed, gw) : rnd[gw];
        if (k == dL - 1)
            w &= tail;
        tile[u] = w;
    }
    for (u32 k = tid; k < dL; k += 256u)
        lmask[k] = mask[k];
    __syncthreads();

    if (tid < nc && !(plain[c0 + tid] & 1u)) {
        const u64 c = c0 + tid;
        u64 pos;
        u32 spare;
        if (DEVRNG) {
            u32 p32;
            enc_draw(seed, c, key, D, p32, spare);
            pos = p32;
        } else {
            pos = chosen[c];
            spare = last[c] & 1u;
        }
        if (pos < n_bits) {
            const u32 wsel = (u32)(pos >> 6), bsel = 63u - (u32)(pos & 63);
            u64 *mine = tile + (size_t)tid * dL;
            bool others = false, all_one = true;
            for (u32 k = 0; k < dL; ++k) {
                u64 m = lmask[k];
                if (k == wsel)
                    m &= ~(1ull << bsel);
                if (m) {
                    others = true;
                    if ((mine[k] & m) != m)
                        all_one = false;
                }
            }
            // src/SecretKey.cpp:73-76; with no other secret slot the reference's v stays 0
            const u64 newbit = (others && all_one) ? 0ull : (u64)spare;
            mine[wsel] = (mine[wsel] & ~(1ull << bsel)) | (newbit << bsel);
        }
    }
    __syncthreads();

    for (u32 u = tid; u < nw; u += 256u) {
        const u32 c = csgn_fastdiv(u, ddL);
        const u32 k = u - c * dL;
        u64 w = tile[u];
        if (plain[c0 + c] & 1u)
            w |= lmask[k];                      // src/SecretKey.cpp:44-45
        out[c0 * dL + u] = w;
    }
}

// ---------------------------------------------------------------------------------------
// Ciphertext::applyPermutation (src/Ciphertext.cpp:7-82): new bit j = old bit perm[j],
// MSB-first.  A bit permutation is a gather of single bits, so the natural wavefront form
// is one lane per OUTPUT BIT and a __ballot per output word:
//   lane l of the wave owns position j = 64*w + 63 - l of output word w, so the 64-bit
//   ballot of "source bit perm[j] is set" IS output word w (ballot bit l <-> word bit l).
// The (word, shift) of every source bit depends only on perm, so each wave decodes it once
// into registers (64 output words per pass) and then sweeps the terms of its tile; the
// source terms sit in LDS (coalesced staging), the gather is one ds_read_b64 per lane.
// The reference's O(N) byte-per-bit scratch arrays (src/Ciphertext.cpp:20-34) disappear.
// ---------------------------------------------------------------------------------------
// NW = output words decoded per pass (a compile-time count so the per-word tables stay in
// registers).  Per term and word the lane work is: one ds_read_b32 of the 32-bit half that
// holds its source bit, one AND with its bit mask, the compare that forms the ballot, and
// two v_writelane that drop the ballot into lane w -- all reads of a term are issued
// before the first ballot so LDS latency overlaps.
// encrypt, fast form: the same K-aligned-segments structure as k_term_hits_seg.  One lane per
// 16-byte (or 8-byte) unit, K passes, TB whole ciphertexts per workgroup, everything a lane
// needs stays in registers; the only cross-lane fact -- "do all OTHER secret positions of my
// ciphertext hold 1?" (src/SecretKey.cpp:60-76) -- is decided from two __ballot bit strings in
// LDS exactly like a decrypt verdict.  No staging of the words through LDS, 16-byte stores.
template <int VEC>
struct UnitWords {
    u64 w[VEC];
};
__device__ inline UnitWords<2> unit_to_words(unit16 v)
{
    UnitWords<2> r;
    r.w[0] = ((u64)v.y << 32) | v.x;
    r.w[1] = ((u64)v.w << 32) | v.z;
    return r;
}
__device__ inline UnitWords<1> unit_to_words(unit8 v)
{
    UnitWords<1> r;
    r.w[0] = v;
    return r;
}
__device__ inline void words_to_unit(const UnitWords<2> &r, unit16 &v)
{
    v.x = (u32)r.w[0];
    v.y = (u32)(r.w[0] >> 32);
    v.z = (u32)r.w[1];
    v.w = (u32)(r.w[1] >> 32);
}
__device__ inline void words_to_unit(const UnitWords<1> &r, unit8 &v) { v = r.w[0]; }

template <typename Unit, int K, bool DEVRNG>
__global__ void __launch_bounds__(256) k_encrypt_seg(u64 n_bits, u32 dL, u32 U, FastDiv dU, u64 D, u64 batch,
                                                     u32 TB, const uint8_t *__restrict__ plain,
                                                     const Unit *__restrict__ rnd,
                                                     const u32 *__restrict__ chosen,
                                                     const uint8_t *__restrict__ last,
                                                     const u64 *__restrict__ key,
                                                     const Unit *__restrict__ mask, u64 seed,
                                                     Unit *__restrict__ out)
{
    constexpr int VEC = sizeof(Unit) / 8;
    __shared__ u32 s_pos[256];                  // chosen position of ciphertext t, ~0u = none (plaintext 1)
    __shared__ unsigned char s_spare[256], s_plain[256], s_clear[256];
    __shared__ u64 others_bits[K * 4 + 1], fail_bits[K * 4 + 1];

    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const u64 c0 = (u64)blockIdx.x * TB;                       // first ciphertext of this workgroup
    const u32 nct = (u32)min((u64)TB, batch - c0);
    const u32 rem = (u32)(n_bits & 63);
    const u64 tail = rem ? ~0ull << (64 - rem) : ~0ull;

    if (tid < TB) {
        u32 pos = 0xFFFFFFFFu;
        unsigned char sp = 0, pl = 1;
        if (tid < nct) {
            const u64 c = c0 + tid;
            pl = plain[c] & 1u;
            if (!pl) {
                if (DEVRNG) {
                    u32 coin;
                    enc_draw(seed, c, key, D, pos, coin);
                    sp = (unsigned char)coin;
                } else {
                    pos = chosen[c];
                    sp = last[c] & 1u;
                }
                if (pos >= n_bits)
                    pos = 0xFFFFFFFFu;                         // invalid input: leave the words alone
            }
        }
        s_pos[tid] = pos;
        s_spare[tid] = sp;
        s_plain[tid] = pl;
    }
    if (tid == 0) {
        others_bits[K * 4] = 0;
        fail_bits[K * 4] = 0;
    }
    __syncthreads();

    const u64 last_unit = batch * (u64)U - 1;
    UnitWords<VEC> w[K], m[K];
    u32 tk[K];                                                // (term << 16) | unit-in-term
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const u32 local = (u32)j * 256u + tid;
        const u32 t = csgn_fastdiv(local, dU), k = local - t * U;
        tk[j] = (t << 16) | k;
        m[j] = unit_to_words(mask[k]);
        const u64 g = min(c0 * U + local, last_unit);          // global unit index (clamped)
        if (DEVRNG) {
#pragma unroll
            for (int q = 0; q < VEC; ++q)
                w[j].w[q] = csgn_rng_word(seed, g * VEC + q);
        } else {
            w[j] = unit_to_words(rnd[g]);
        }
        if (k == U - 1)
            w[j].w[VEC - 1] &= tail;                           // padding bits of the last word stay 0
    }
    // per unit: does it hold secret positions other than the chosen one, and are they all 1?
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const u32 t = tk[j] >> 16, k = tk[j] & 0xFFFFu;
        const u32 pos = s_pos[t];
        bool others = false, fail = false;
        if (t < nct && pos != 0xFFFFFFFFu) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                u64 mm = m[j].w[q];
                if ((pos >> 6) == k * VEC + (u32)q)
                    mm &= ~(1ull << (63u - (pos & 63u)));
                others = others || (mm != 0);
                fail = fail || ((w[j].w[q] & mm) != mm);
            }
        }
        const u64 bo = __ballot(others), bf = __ballot(fail);
        if (lane == 0) {
            others_bits[j * 4 + wave] = bo;
            fail_bits[j * 4 + wave] = bf;
        }
    }
    __syncthreads();
    if (tid < TB) {
        const u32 start = tid * U, wd = start >> 6, sh = start & 63u;
        u64 vo = others_bits[wd] >> sh, vf = fail_bits[wd] >> sh;
        if (sh) {
            vo |= others_bits[wd + 1] << (64u - sh);
            vf |= fail_bits[wd + 1] << (64u - sh);
        }
        const u64 need = (U >= 64u) ? ~0ull : ((1ull << U) - 1ull);
        // src/SecretKey.cpp:73-76: clear the chosen slot iff other secret slots exist and are all 1
        s_clear[tid] = ((vo & need) != 0 && (vf & need) == 0) ? 1 : 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const u32 t = tk[j] >> 16, k = tk[j] & 0xFFFFu;
        if (t < nct) {
            if (s_plain[t]) {
#pragma unroll
                for (int q = 0; q < VEC; ++q)
                    w[j].w[q] |= m[j].w[q];                    // src/SecretKey.cpp:44-45
            } else {
                const u32 pos = s_pos[t];
                if (pos != 0xFFFFFFFFu && (pos >> 6) / VEC == k) {
                    const u32 q = (pos >> 6) - k * VEC;
                    const u64 bit = 1ull << (63u - (pos & 63u));
                    const u64 newbit = s_clear[t] ? 0ull : (s_spare[t] ? bit : 0ull);
#pragma unroll
                    for (int qq = 0; qq < VEC; ++qq)
                        if ((u32)qq == q)
                            w[j].w[qq] = (w[j].w[qq] & ~bit) | newbit;
                }
            }
            Unit v;
            words_to_unit(w[j], v);
            unit_store<Unit, true>(out + c0 * U + (u32)j * 256u + tid, v);
        }
    }
}

// v_writelane_b32: drop a wave-uniform 64-bit value into ONE lane of a VGPR pair (hipcc 7.2
// exposes no builtin for it).  `lane` must be a compile-time constant.  The s_nop is the
// gfx940+ "VALU writes SGPR -> VALU reads that SGPR" hazard (2 wait states): the ballot is
// produced by a v_cmp immediately before, and hipcc pads nothing inside an asm statement
// (observed: without it the low word of some lanes read a stale SGPR).
__device__ inline void write_lane64(u64 uniform_value, int lane, u32 &lo, u32 &hi)
{
    asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
                 : "+v"(lo), "+v"(hi)
                 : "s"((u32)uniform_value), "s"((u32)(uniform_value >> 32)), "n"(lane));
}

template <int NW>
__global__ void __launch_bounds__(256) k_permute(u64 n_bits, u32 dL, FastDiv ddL, u64 out_terms,
                                                 u64 in_stride_words, u32 have_input, u32 TB,
                                                 const u64 *__restrict__ terms,
                                                 const u32 *__restrict__ perm, u64 *__restrict__ out)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    u64 *tile = reinterpret_cast<u64 *>(smem_raw);          // TB terms x dL words

    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const u64 t0 = (u64)blockIdx.x * TB;
    const u32 nt = (u32)min((u64)TB, out_terms - t0);

    for (u32 u = tid; u < nt * dL; u += 256u) {
        const u32 t = csgn_fastdiv(u, ddL);
        const u32 k = u - t * dL;
        tile[u] = have_input ? terms[(t0 + t) * in_stride_words + k] : 0ull;
    }
    __syncthreads();

    for (u32 c0 = 0; c0 < dL; c0 += NW) {
        const u32 nw = min((u32)NW, dL - c0);
        // this lane's source for each of the NW output words: byte offset of the 32-bit half
        // inside a term, and the bit inside that half (mask 0 = "no source": yields 0)
        u32 off[NW], msk[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            off[w] = 0;
            msk[w] = 0;
            const u64 j = (u64)(c0 + w) * 64u + (63u - lane);
            if ((u32)w < nw && j < n_bits) {
                const u32 p = perm[j];
                if (p < n_bits) {
                    const u32 b = 63u - (p & 63u);          // bit index in the uint64, LSB = 0
                    off[w] = (p >> 6) * 8u + ((b >> 5) << 2);
                    msk[w] = 1u << (b & 31u);
                }
            }
        }
        for (u32 t = wave; t < nt; t += 4u) {
            const unsigned char *src = smem_raw + (size_t)t * dL * 8u;
            u32 v[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w)
                v[w] = *reinterpret_cast<const u32 *>(src + off[w]);
            u32 lo = 0, hi = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const u64 b = __ballot((v[w] & msk[w]) != 0u);
                write_lane64(b, w, lo, hi);
            }
            if (lane < nw)
                out[(t0 + t) * dL + c0 + lane] = ((u64)hi << 32) | (u64)lo;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Permutation, bit-plane form.  One wave owns 64 terms and the permutation is the same for all of
// them, so it turns the 64 x N bit matrix on its side: a 64x64 bit transpose inside the wave
// (6 exchange stages: v_permlane32_swap / v_permlane16_swap for the two coarse ones, DPP lane
// exchange + v_alignbit + v_bfi for the four inside a 16-bit field) leaves "bit j of 64 terms"
// in one 64-bit word.  Moving bit j to bit j' is then a plain 8-byte LDS move, and a second
// transpose turns the planes back into terms.
// ~1.2 lane-operations per bit instead of ~6 for the ballot form above.
// ---------------------------------------------------------------------------------------
template <int K>
__device__ inline u32 lane_xor(u32 v)
{
    // all on the VALU (DPP): an LDS-crossbar swizzle costs a round trip the two waves a SIMD
    // holds here cannot hide
    if (K == 8)
        return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x128 /* row_ror:8 */, 0xF, 0xF, true);
    if (K == 4) {
        // lanes with bit 2 clear read lane+4 (row_shl:4, banks 0 and 2), the others lane-4
        int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x104 /* row_shl:4 */, 0xF, 0x5, false);
        return (u32)__builtin_amdgcn_update_dpp(t, (int)v, 0x114 /* row_shr:4 */, 0xF, 0xA, false);
    }
    if (K == 2)
        return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, true);
    return (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true);
}

// per-lane constants of the four in-word stages (K = 8, 4, 2, 1)
struct TrLane {
    u32 rot[4];     // rotate-right amount that lines the partner's half up with mine
    u32 keep[4];    // bits of my own word that stay
    u32 sel8;       // v_perm_b32 selector of the byte stage
};

__device__ inline TrLane tr_lane(u32 lane)
{
    const u32 M[4] = {0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
    TrLane c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32 k = 8u >> i;
        const bool upper = (lane & k) != 0;
        c.rot[i] = upper ? k : 32u - k;
        c.keep[i] = upper ? ~M[i] : M[i];
    }
    c.sel8 = (lane & 8u) ? 0x03070105u : 0x06020400u;
    return c;
}

// One in-word stage on R independent registers, written operation by operation (all lane
// exchanges, then all rotates, then all merges) and pinned with sched_barrier: a dependent VALU
// result is not available to the next instruction of the same wave without a stall, and only one
// or two waves share a SIMD here, so the independent registers have to interleave.
template <int I, int R>
__device__ inline void tr_stage_n(u32 (&h)[R], const TrLane &c)
{
    u32 y[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        y[r] = lane_xor<(8 >> I)>(h[r]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < R; ++r)
        y[r] = __builtin_amdgcn_alignbit(y[r], y[r], c.rot[I]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < R; ++r)   // (h & keep) | (y & ~keep); hipcc otherwise emits and/and/or
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(h[r]) : "v"(c.keep[I]), "v"(h[r]), "v"(y[r]));
    __builtin_amdgcn_sched_barrier(0);
}

// On return bit c of lane b's word is bit b of lane c's input (LSB = bit 0); h[2q], h[2q+1] are
// the low and high halves of word q.  Stages 32 and 16 are pure data movement between registers
// and lane groups:
//   32: lanes 0..31 hand their high words to lanes 32..63 and take those lanes' low words
//       (v_permlane32_swap);
//   16: the same between 16-bit halves and 16-lane rows -- gather the low halves of (lo,hi) in
//       one register and the high halves in another (v_perm_b32), v_permlane16_swap, scatter back.
template <int Q>
__device__ inline void wave_transpose64_n(u32 (&h)[2 * Q], const TrLane &c)
{
    u32 a[Q], b[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        auto s32 = __builtin_amdgcn_permlane32_swap(h[2 * q], h[2 * q + 1], false, false);
        a[q] = s32[0];
        b[q] = s32[1];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        h[2 * q] = __builtin_amdgcn_perm(b[q], a[q], 0x05040100u);       // low halves
        h[2 * q + 1] = __builtin_amdgcn_perm(b[q], a[q], 0x07060302u);   // high halves
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        auto s16 = __builtin_amdgcn_permlane16_swap(h[2 * q], h[2 * q + 1], false, false);
        a[q] = s16[0];
        b[q] = s16[1];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        h[2 * q] = __builtin_amdgcn_perm(b[q], a[q], 0x05040100u);
        h[2 * q + 1] = __builtin_amdgcn_perm(b[q], a[q], 0x07060302u);
    }
    __builtin_amdgcn_sched_barrier(0);
    // stage 8 moves whole bytes: one v_perm_b32 picks {own b0, partner b0, own b2, partner b2}
    // (lanes with bit 3 clear) or {partner b1, own b1, partner b3, own b3}
    {
        u32 y[2 * Q];
#pragma unroll
        for (int r = 0; r < 2 * Q; ++r)
            y[r] = lane_xor<8>(h[r]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 2 * Q; ++r)
            h[r] = __builtin_amdgcn_perm(y[r], h[r], c.sel8);
        __builtin_amdgcn_sched_barrier(0);
    }
    tr_stage_n<1, 2 * Q>(h, c);
    tr_stage_n<2, 2 * Q>(h, c);
    tr_stage_n<3, 2 * Q>(h, c);
}

constexpr int kPermUnroll = 5;

// Workgroup = ONE wave = 64 terms at a time, persistent over groups of 64 terms.  LDS: rows[64][SA]
// (SA odd: conflict-free column reads), planes[dLp*64 + 1] (the last entry stays 0: "no source"),
// psrc[dLp*64] (u16 source plane of every output bit).  The next group's terms are
// loaded into registers (LQ words per lane, all in flight) while this group is transposed.
template <int LQ, bool PIPE, int UW>
__global__ void __launch_bounds__(64) k_permute_planes(u64 n_bits, u32 dL, FastDiv dUd, u64 out_terms,
                                                       u64 in_stride_words,
                                                       const u64 *__restrict__ terms,
                                                       const u32 *__restrict__ perm,
                                                       u64 *__restrict__ out)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const u32 dLp = (dL + kPermUnroll - 1) / kPermUnroll * kPermUnroll;
    const u32 SA = dLp | 1u;
    const u32 none = dLp * 64u;                    // index of the all-zero plane
    u64 *rows = reinterpret_cast<u64 *>(smem_raw);
    u64 *planes = rows + 64u * SA;                 // none + 1 entries
    unsigned short *psrc = reinterpret_cast<unsigned short *>(planes + none + 1);   // none entries
    const u32 lane = threadIdx.x;
    const TrLane trc = tr_lane(lane);
    const u32 nb = (u32)n_bits;
    const u32 Ud = dL / UW;                        // staging units (UW words = 8 or 16 bytes) per term
    const u32 nu = 64u * Ud;                       // units in a full group
    const u64 groups = (out_terms + 63) / 64;
    typedef u64 StageUnit __attribute__((ext_vector_type(UW)));

    // this lane's LQ staging slots: unit u = q*64 + lane of the group -> (term t, word k)
    u32 rowoff[LQ];        // t*SA + k in rows[]
    u32 tq[LQ];            // t
#pragma unroll
    for (int q = 0; q < LQ; ++q) {
        const u32 u = min((u32)q * 64u + lane, nu - 1u);
        const u32 t = csgn_fastdiv(u, dUd);
        tq[q] = t;
        rowoff[q] = t * SA + (u - t * Ud) * UW;
    }
    const u64 gskip = in_stride_words - SA;        // input offset = rowoff + t*(stride - SA)
    StageUnit v[LQ];
    auto fetch = [&](u64 g) {
        const u64 t0 = g * 64;
        const u32 nt = (u32)min((u64)64, out_terms - t0);
        const u64 *base = terms + t0 * in_stride_words;
#pragma unroll
        for (int q = 0; q < LQ; ++q) {
            // a term past the end re-reads the group's first term (discarded below)
            const u64 off = tq[q] < nt ? rowoff[q] + tq[q] * gskip : 0ull;
            v[q] = *reinterpret_cast<const StageUnit *>(base + off);
        }
    };
    auto rows_put = [&](u32 off, StageUnit x, bool live) {
#pragma unroll
        for (int i = 0; i < UW; ++i)     // SA is odd: a row starts 8-byte, not 16-byte, aligned
            rows[off + i] = live ? x[i] : 0ull;
    };
    auto rows_get = [&](u32 off) {
        StageUnit x;
#pragma unroll
        for (int i = 0; i < UW; ++i)
            x[i] = rows[off + i];
        return x;
    };

    u64 g = blockIdx.x;
    if (g < groups)
        fetch(g);
    // 0. source plane of every output bit (`none` for padding bits and out-of-range entries)
    for (u32 j0 = 0; j0 < none; j0 += 64u * kPermUnroll) {
        u32 p[kPermUnroll];
#pragma unroll
        for (int q = 0; q < kPermUnroll; ++q)
            p[q] = perm[min(j0 + (u32)q * 64u + lane, nb - 1u)];
#pragma unroll
        for (int q = 0; q < kPermUnroll; ++q) {
            const u32 j = j0 + (u32)q * 64u + lane;
            psrc[j] = (unsigned short)(j < nb && p[q] < nb ? p[q] : none);
        }
    }
    if (lane == 0)
        planes[none] = 0;

    // Order of the vector-memory operations inside one turn: wait for this group's terms, issue the
    // PREVIOUS group's stores (results parked in o[] when PIPE), issue the next group's loads, then
    // compute.  Everything issued has the whole compute phase to complete, so the vmcnt(0) at the
    // top of the next turn finds it done (loads and stores share one in-order counter on gfx9).
    StageUnit o[PIPE ? LQ : 1];
    u64 gprev = ~0ull;
    auto flush = [&](u64 gp) {
        const u64 t0 = gp * 64;
        const u32 nt = (u32)min((u64)64, out_terms - t0);
        StageUnit *obase = reinterpret_cast<StageUnit *>(out + t0 * dL);
#pragma unroll
        for (int q = 0; q < LQ; ++q)
            if ((u32)q * 64u + lane < nt * Ud)
                obase[(u32)q * 64u + lane] = o[PIPE ? q : 0];
    };

    for (; g < groups; g += gridDim.x) {
        const u64 t0 = g * 64;
        const u32 nt = (u32)min((u64)64, out_terms - t0);
        // 1. rows <- the 64 terms fetched earlier (slots past the end repeat the last unit)
#pragma unroll
        for (int q = 0; q < LQ; ++q)
            rows_put(rowoff[q], v[q], tq[q] < nt);
        __syncthreads();
        if (PIPE && gprev != ~0ull)
            flush(gprev);
        if (g + gridDim.x < groups)
            fetch(g + gridDim.x);
        // 2. rows -> bit planes: lane b ends up with term-bit j = w*64 + 63 - b of all 64 terms.
        //    kPermUnroll words in flight: the six exchange stages of one transpose are a serial
        //    chain, independent words fill the gaps.  dLp is dL rounded up to kPermUnroll and every
        //    LDS array is sized for it, so the loops need no guards (tail columns hold don't-cares).
        for (u32 w0 = 0; w0 < dLp; w0 += kPermUnroll) {
            u32 h[2 * kPermUnroll];
#pragma unroll
            for (int q = 0; q < kPermUnroll; ++q) {
                const u64 x = rows[lane * SA + w0 + (u32)q];
                h[2 * q] = (u32)x;
                h[2 * q + 1] = (u32)(x >> 32);
            }
            wave_transpose64_n<kPermUnroll>(h, trc);
#pragma unroll
            for (int q = 0; q < kPermUnroll; ++q)
                planes[(w0 + (u32)q) * 64u + 63u - lane] = ((u64)h[2 * q + 1] << 32) | h[2 * q];
        }
        __syncthreads();
        // 3. new bit j <- old bit perm[j]; planes -> rows
        for (u32 w0 = 0; w0 < dLp; w0 += kPermUnroll) {
            u32 h[2 * kPermUnroll];
#pragma unroll
            for (int q = 0; q < kPermUnroll; ++q) {
                const u64 y = planes[psrc[(w0 + (u32)q) * 64u + 63u - lane]];
                h[2 * q] = (u32)y;
                h[2 * q + 1] = (u32)(y >> 32);
            }
            wave_transpose64_n<kPermUnroll>(h, trc);
#pragma unroll
            for (int q = 0; q < kPermUnroll; ++q)
                rows[lane * SA + w0 + (u32)q] = ((u64)h[2 * q + 1] << 32) | h[2 * q];
        }
        __syncthreads();
        // 4. rows -> out, coalesced: now, or parked in registers until the next turn's loads are in
        if (PIPE) {
#pragma unroll
            for (int q = 0; q < LQ; ++q)
                o[q] = rows_get(rowoff[q]);
            gprev = g;
        } else {
            StageUnit *obase = reinterpret_cast<StageUnit *>(out + t0 * dL);
#pragma unroll
            for (int q = 0; q < LQ; ++q)
                if ((u32)q * 64u + lane < nt * Ud)
                    obase[(u32)q * 64u + lane] = rows_get(rowoff[q]);
        }
        __syncthreads();
    }
    if (PIPE && gprev != ~0ull)
        flush(gprev);
}

// ---------------------------------------------------------------------------------------
// harness kernels (definitions shared with oracle/csgn_oracle.c)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_synth_fill(u64 seed, u32 dL, u64 tail, u64 first_word,
                                                    u64 n_words, u64 *__restrict__ out)
{
    u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 stride = (u64)gridDim.x * 256u;
    for (; i < n_words; i += stride) {
        const u64 idx = first_word + i;
        u64 w = csgn_splitmix64(seed + CSGN_GOLDEN * (idx + 1));
        if (idx % dL == dL - 1)
            w &= tail;
        out[i] = w;
    }
}

__global__ void __launch_bounds__(256) k_digest(const u64 *__restrict__ w, u64 n_words,
                                                u64 first_index, u64 *__restrict__ d_digest)
{
    u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 stride = (u64)gridDim.x * 256u;
    u64 acc = 0;
    for (; i < n_words; i += stride)
        acc += csgn_splitmix64(w[i] + CSGN_GOLDEN * (first_index + i + 1));
    // wave-level fold, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & (kWave - 1)) == 0 && acc != 0)
        atomicAdd(reinterpret_cast<unsigned long long *>(d_digest), acc);
}

// ------------------------------------------------------------------------- launch helpers

template <typename T>
bool aligned16(const T *p)
{
    return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// Block size for the tiled kernel: a multiple of 64 that U divides (so every column a lane
// owns needs the same left unit), at least 256 threads, at most 512.  0 = none exists.
u32 samek_block(u32 U)
{
    for (u32 bs = 256; bs <= 512; bs += 64)
        if (bs % U == 0)
            return bs;
    return 0;
}

template <typename Unit, int M, bool RAGGED>
hipError_t launch_tiled_m(const MulArgs &a, u32 bs, bool samek, bool nt, u32 blocks, size_t lds,
                          hipStream_t s)
{
    if (samek) {
        if (nt)
            k_mul_tiled<Unit, M, true, RAGGED, true><<<blocks, bs, lds, s>>>(a);
        else
            k_mul_tiled<Unit, M, true, RAGGED, false><<<blocks, bs, lds, s>>>(a);
    } else {
        if (nt)
            k_mul_tiled<Unit, M, false, RAGGED, true><<<blocks, bs, lds, s>>>(a);
        else
            k_mul_tiled<Unit, M, false, RAGGED, false><<<blocks, bs, lds, s>>>(a);
    }
    return hipGetLastError();
}

// Launch the tiled kernel for `pairs` pairs whose shapes are bounded by (t1, t2).
template <typename Unit, bool RAGGED>
hipError_t launch_tiled(MulArgs a, u64 pairs, u32 U, hipStream_t s)
{
    const MulTuning tune = mul_tuning();
    // With M > 1 a block size that U divides lets every column of a lane share one LDS read
    // (SAMEK); with the default M == 1 there is a single column per lane and 256 threads
    // (4 KiB-aligned row segments) measured fastest.
    // auto: 256 threads x 4 KiB row segments -- one 16-byte unit per lane, or two 8-byte units
    // when dL is odd (measured at N=1300: 5.3 -> 6.4 TB/s, the 2 KiB segments of M=1 lose)
    const int m_req = tune.m ? tune.m : (sizeof(Unit) == 8 ? 2 : 1);
    u32 bs = tune.bs ? (u32)tune.bs : (tune.m > 1 ? samek_block(U) : 256u);
    if (bs == 0)
        bs = 256;
    const bool samek = bs % U == 0;
    const u32 cu = a.t2 * U;
    // do not give a lane more columns than the row has
    int m = m_req;
    while (m > 1 && (u64)bs * (m / 2) >= cu)
        m /= 2;
    // left tile: TI terms, capped so the LDS image stays <= 32 KB
    u32 ti = (u32)tune.ti;
    const u32 cap = (u32)(32768u / (U * sizeof(Unit)));
    if (ti > cap)
        ti = cap ? cap : 1;
    if (ti > a.t1)
        ti = a.t1;
    a.U = U;
    a.TI = ti;
    // XCD-contiguous order helps the (linear) flat kernel but measured 7 % slower on the tiled
    // kernel's comb-shaped store pattern, so it is opt-in here (CSGN_MUL_XCD=2)
    a.xcd_remap = tune.xcd == 2 ? 1u : 0u;
    a.col_tiles = (cu + bs * m - 1) / (bs * m);
    a.row_tiles = (a.t1 + ti - 1) / ti;
    const u64 tiles = (u64)a.col_tiles * a.row_tiles;
    const size_t lds = (size_t)ti * U * sizeof(Unit);
    // at most kMaxBlocks512 workgroups per launch (gridDim.x * blockDim.x < 2^32)
    if (tiles > kMaxBlocks512 || pairs >= (1ull << 32))
        return hipErrorInvalidValue;
    const u64 max_pairs = kMaxBlocks512 / tiles;
    for (u64 p0 = 0; p0 < pairs; p0 += max_pairs) {
        const u64 np = (pairs - p0 < max_pairs) ? pairs - p0 : max_pairs;
        MulArgs b = a;
        b.pair_base = RAGGED ? (u32)p0 : 0u;
        if (!RAGGED) {
            b.L = reinterpret_cast<const Unit *>(a.L) + p0 * a.t1 * U;
            b.R = reinterpret_cast<const Unit *>(a.R) + p0 * a.t2 * U;
            b.out = reinterpret_cast<Unit *>(a.out) + p0 * a.t1 * a.t2 * U;
        }
        const u32 blocks = (u32)(np * tiles);
        hipError_t e;
        switch (m) {
        case 1: e = launch_tiled_m<Unit, 1, RAGGED>(b, bs, samek, tune.nt, blocks, lds, s); break;
        case 2: e = launch_tiled_m<Unit, 2, RAGGED>(b, bs, samek, tune.nt, blocks, lds, s); break;
        case 4: e = launch_tiled_m<Unit, 4, RAGGED>(b, bs, samek, tune.nt, blocks, lds, s); break;
        default: e = launch_tiled_m<Unit, 8, RAGGED>(b, bs, samek, tune.nt, blocks, lds, s); break;
        }
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

// Which kernel an all-pairs product takes (measured with COLD operands -- every launch reads pairs
// that are not in any cache, tools/bench_cold.py, profiles/r01/bench_cold*.log):
//   * output >= 4x the operands (t1*t2 >= 4*(t1+t2)), 16-byte units, and at least 4 MB of operands
//     in the launch (a stream, not a single product): the flat kernel (one output
//     unit per lane, linear 4 KiB per workgroup, XCD-contiguous order) AFTER a touch pass that
//     reads one dword of every operand line.  The flat kernel alone stalls on the first touch of
//     every left term (an HBM miss under full write load, 4.6 TB/s at 1024x1024); with the
//     operands already in the memory-side cache it runs at 7.2-7.5 TB/s against 6.9 for the
//     LDS-tiled kernel, 6.2 against 3.9 at 32x32, 5.7 against 4.6 at 8x8.  The touch is a second
//     read of the operands, hence the 4x condition, and is done per <= 64 MB of operands so that
//     they are still in the 256 MB cache when their pairs run.
//   * otherwise rows shorter than a workgroup (t2*U < 256 units; < 64 for the 8-byte units of an
//     odd dL, whose flat kernel only writes 2 KiB per workgroup): the flat kernel, no touch (the
//     tiled kernel leaves column lanes idle: 1.4 vs 6.0 TB/s at t2 = 1).
//   * everything else (thin products with long rows, 8-byte units): the LDS-tiled kernel.
// CSGN_MUL_FLAT (-1 tiled, k > 0 flat with k units per lane) and CSGN_MUL_TOUCH (0..3: bit 0 left,
// bit 1 right operand) override for sweeps.
struct MulPlan {
    int flat;       // 0 = LDS-tiled kernel, k > 0 = flat kernel with k units per lane
    int touch;      // operands to pull into the memory-side cache first (bit 0 left, bit 1 right)
};

static MulPlan mul_plan(size_t unit_bytes, u32 U, u64 t1, u64 t2, u64 pairs)
{
    const MulTuning tune = mul_tuning();
    const u64 PU = t1 * t2 * U;
    MulPlan p = {0, 0};
    if (PU >= (1ull << 31) || tune.flat == -1)
        return p;
    const int touch_env = env_int("CSGN_MUL_TOUCH", -1);
    if (tune.flat > 0) {
        p.flat = tune.flat;
        p.touch = touch_env > 0 ? (touch_env & 3) : 0;
        return p;
    }
    // a call with a few MB of operands is not a stream: they are usually still cached from the
    // kernel that produced them, and two extra launches triple the cost of a small product
    // (class API, 64x64: 10.9 us per multiply with the touch, 3.0-3.8 without)
    const bool streaming = pairs * (t1 + t2) * U * unit_bytes >= (4ull << 20);
    if (unit_bytes == 16 && t1 * t2 >= 4 * (t1 + t2) && (streaming || touch_env > 0)) {
        p.flat = 1;
        p.touch = touch_env >= 0 ? (touch_env & 3) : 3;
    } else if (t2 * U < (unit_bytes == 16 ? 256u : 64u)) {
        p.flat = 1;
        p.touch = touch_env > 0 ? (touch_env & 3) : 0;
    }
    return p;
}

// Read one dword of every 128-byte line of [p, p+bytes): pulls an operand into the memory-side
// cache ahead of a kernel whose first touch of it would otherwise be a serialising HBM miss.
__global__ void __launch_bounds__(256) k_touch(const u32 *__restrict__ p, u64 lines, u32 dwords_per_line)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < lines) {
        const u32 v = p[i * dwords_per_line];
        asm volatile("" ::"v"(v));
    }
}

// One uniform chunk (pairs are contiguous in L, R and out).
template <typename Unit>
hipError_t mul_uniform_chunk(u32 U, u64 pairs, u32 t1, u32 t2, const u64 *L, const u64 *R, u64 *out,
                             hipStream_t s)
{
    const Unit *Lu = reinterpret_cast<const Unit *>(L);
    const Unit *Ru = reinterpret_cast<const Unit *>(R);
    Unit *Ou = reinterpret_cast<Unit *>(out);
    const u64 PU = (u64)t1 * t2 * U;
    const u64 total = pairs * PU;
    if (total == 0)
        return hipSuccess;
    const MulTuning tune = mul_tuning();
    if (t1 == 1 && t2 == 1) {
        // at most 2^31-1 workgroups of 256 units per launch
        const u64 per_launch = kMaxBlocks256 * 256u;
        for (u64 u0 = 0; u0 < total; u0 += per_launch) {
            const u64 nu = (total - u0 < per_launch) ? total - u0 : per_launch;
            k_and_stream<Unit, true><<<ceil_div_u64(nu, 256u), 256, 0, s>>>(Lu + u0, Ru + u0, Ou + u0, nu);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess)
                return e;
        }
        return hipSuccess;
    }
    const MulPlan plan = mul_plan(sizeof(Unit), U, t1, t2, pairs);
    if (plan.flat) {
        const int mf = plan.flat;
        // left-term prefetch from inside the kernel (only without the touch pass, only when a row
        // fills a workgroup): ~6 MB of output ahead per XCD stream
        u32 pfr = 0;
        if (!plan.touch && (u64)t2 * U >= 256u) {
            const u64 ahead = (u64)env_int("CSGN_MUL_PF_KB", tune.xcd ? 6144 : 49152) << 10;
            const u64 row_bytes = (u64)t2 * U * sizeof(Unit);
            pfr = (u32)((ahead + row_bytes - 1) / row_bytes);
        }
        u64 pairs_per = (0xFFFFFF00ull / PU) ? (0xFFFFFF00ull / PU) : 1;   // units (= threads) per launch < 2^32
        if (plan.touch) {
            // touched operands must still be in the 256 MB memory-side cache when their pair runs:
            // at most 64 MB of them per touch + launch
            const u64 op_bytes = (u64)(t1 + t2) * U * sizeof(Unit);
            pairs_per = std::min<u64>(pairs_per, std::max<u64>(1, (64ull << 20) / op_bytes));
        }
        const FastDiv dPU = csgn_fastdiv_make((u32)PU), dCU = csgn_fastdiv_make(t2 * U),
                      dU = csgn_fastdiv_make(U);
        for (u64 p0 = 0; p0 < pairs; p0 += pairs_per) {
            const u64 np = (pairs - p0 < pairs_per) ? pairs - p0 : pairs_per;
            const u32 tot = (u32)(np * PU);
            const u32 trows = (u32)(np * t1);
            const u32 blocks = ceil_div_u64(tot, 256u * (u64)mf);
            const Unit *Lc = Lu + p0 * t1 * U, *Rc = Ru + p0 * t2 * U;
            Unit *Oc = Ou + p0 * PU;
            if (plan.touch) {
                const u32 lb = 128;       // the L2 fills whole 128-byte lines (a 256-byte stride loses the gain)
                const u64 ll = (np * t1 * U * sizeof(Unit) + lb - 1) / lb, rl = (np * t2 * U * sizeof(Unit) + lb - 1) / lb;
                if (plan.touch & 1)
                    k_touch<<<ceil_div_u64(ll, 256u), 256, 0, s>>>(reinterpret_cast<const u32 *>(Lc), ll, lb / 4);
                if (plan.touch & 2)
                    k_touch<<<ceil_div_u64(rl, 256u), 256, 0, s>>>(reinterpret_cast<const u32 *>(Rc), rl, lb / 4);
            }
#define CSGN_FLAT(MF)                                                                                  \
    do {                                                                                               \
        if (tune.xcd)                                                                                  \
            k_mul_flat<Unit, MF, true><<<blocks, 256, 0, s>>>(Lc, Rc, Oc, tot, t1, t2, U, dPU, dCU, dU, pfr, trows);  \
        else                                                                                           \
            k_mul_flat<Unit, MF, false><<<blocks, 256, 0, s>>>(Lc, Rc, Oc, tot, t1, t2, U, dPU, dCU, dU, pfr, trows); \
    } while (0)
            switch (mf) {
            case 1: CSGN_FLAT(1); break;
            case 2: CSGN_FLAT(2); break;
            case 4: CSGN_FLAT(4); break;
            default: CSGN_FLAT(8); break;
            }
#undef CSGN_FLAT
            hipError_t e = hipGetLastError();
            if (e != hipSuccess)
                return e;
        }
        return hipSuccess;
    }
    MulArgs a = {};
    a.L = L;
    a.R = R;
    a.out = out;
    a.t1 = t1;
    a.t2 = t2;
    return launch_tiled<Unit, false>(a, pairs, U, s);
}

} // namespace

// ------------------------------------------------------------------------------ public

MulTuning mul_tuning()
{
    MulTuning t;
    // Defaults from the MI355X sweeps recorded in DESIGN.md / profiles/: 256-thread workgroups
    // (4 KiB row segments), one column unit per lane, 4 left terms per tile, non-temporal
    // stores.  Short-lived workgroups keep the chip-wide write front dense in address space,
    // which is what HBM rewards; long-lived tiles (TI=64) lose ~20 % to the scattered store pattern.
    t.m = env_int("CSGN_MUL_M", 0);          // 0 = auto: 1 column unit per lane (2 for 8-byte units)
    if (t.m != 1 && t.m != 2 && t.m != 4 && t.m != 8)
        t.m = 0;
    t.ti = env_int("CSGN_MUL_TI", 4);
    if (t.ti < 1)
        t.ti = 1;
    t.nt = env_int("CSGN_MUL_NT", 1) ? 1 : 0;
    t.flat = env_int("CSGN_MUL_FLAT", 0);   // 0 = auto (mul_plan); >0 = flat with that unroll; -1 = always tiled
    if (t.flat != -1 && t.flat != 1 && t.flat != 2 && t.flat != 4 && t.flat != 8)
        t.flat = 0;
    t.xcd = env_int("CSGN_MUL_XCD", 1);      // 0 = dispatch order, 1 = remap flat kernel, 2 = remap both
    if (t.xcd < 0 || t.xcd > 2)
        t.xcd = 1;
    t.bs = env_int("CSGN_MUL_BS", 0);
    if (t.bs % 64 != 0 || t.bs < 64 || t.bs > 512)
        t.bs = 0;
    return t;
}

// 4 KiB chunks per workgroup of the flat ragged kernels: as many as 8 (the workgroup's first search
// is paid once per C chunks) while the grid still has >= 8192 workgroups to fill the chip with.
// CSGN_RAGGED_C = 1, 2, 4, 8, 16 overrides.
static int ragged_chunks(u64 total_units)
{
    const int forced = env_int("CSGN_RAGGED_C", 0);
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16)
        return forced;
    int c = 1;
    while (c < 8 && total_units / (256u * 2u * (u64)c) >= 8192u)
        c *= 2;
    return c;
}

const char *mul_uniform_kernel_name(u64 n_bits, u64 pairs, u64 t1, u64 t2)
{
    const u64 dL = (n_bits + 63) / 64;
    const bool wide = dL % 2 == 0;
    const u32 U = (u32)(wide ? dL / 2 : dL);
    if (t1 == 1 && t2 == 1)
        return "k_and_stream";
    const MulPlan p = mul_plan(wide ? 16 : 8, U, t1, t2, pairs);
    return p.flat ? (p.touch ? "k_touch+k_mul_flat" : "k_mul_flat") : "k_mul_tiled";
}

hipError_t mul_uniform(u64 n_bits, u64 batch, u64 t1, u64 t2, const u64 *L, const u64 *R, u64 *out,
                       u64 out_slots, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0 || t1 == 0 || t2 == 0)
        return hipSuccess;
    const bool wide = (dL % 2 == 0) && aligned16(L) && aligned16(R) && aligned16(out);
    const u32 U = (u32)(wide ? dL / 2 : dL);
    const u64 slots = (out_slots == 0 || out_slots > batch) ? batch : out_slots;
    for (u64 p0 = 0; p0 < batch; p0 += slots) {
        const u64 np = (batch - p0 < slots) ? batch - p0 : slots;
        const u64 *Lc = L + p0 * t1 * dL;
        const u64 *Rc = R + p0 * t2 * dL;
        hipError_t e = wide ? mul_uniform_chunk<unit16>(U, np, (u32)t1, (u32)t2, Lc, Rc, out, s)
                            : mul_uniform_chunk<unit8>(U, np, (u32)t1, (u32)t2, Lc, Rc, out, s);
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

u64 mul_ragged_plan_scratch_words(u64 batch) { return 4 + (batch + 1023) / 1024 + 1; }

hipError_t mul_ragged_plan(u64 batch, const u64 *offL, const u64 *offR, u64 *offOut, u64 *d_work,
                           hipStream_t s)
{
    // d_work: [plan4 (total, max t1, max t2, max t1*t2)][one partial per 1024-pair chunk]
    u64 *plan4 = d_work, *partial = d_work + 4;
    const u64 nchunks = (batch + 1023) / 1024;
    hipError_t e = hipMemsetAsync(d_work, 0, mul_ragged_plan_scratch_words(batch) * 8, s);
    if (e != hipSuccess)
        return e;
    if (nchunks > kMaxBlocks256)
        return hipErrorInvalidValue;
    if (batch)
        k_plan_chunks<<<(u32)nchunks, 256, 0, s>>>(batch, offL, offR, offOut, partial, plan4);
    k_plan_scan_partials<<<1, 1024, 0, s>>>(nchunks, batch, partial, offOut, plan4);
    if (batch)
        k_plan_fix<<<ceil_div_u64(batch, 256), 256, 0, s>>>(batch, partial, offOut);
    return hipGetLastError();
}

hipError_t mul_ragged(u64 n_bits, u64 batch, const u64 *L, const u64 *offL, const u64 *R,
                      const u64 *offR, u64 *out, const u64 *offOut, u64 max_t1, u64 max_t2,
                      u64 total_out_terms, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0 || max_t1 == 0 || max_t2 == 0 || total_out_terms == 0)
        return hipSuccess;
    if (batch >= (1ull << 32))
        return hipErrorInvalidValue;
    const bool wide = (dL % 2 == 0) && aligned16(L) && aligned16(R) && aligned16(out);
    const u32 U = (u32)(wide ? dL / 2 : dL);
    // Nearly uniform batches of large products keep the LDS-tiled kernel (one grid sized for the
    // largest shape); anything skewed or small goes through the flat ragged kernel, whose grid
    // is the real output.
    const bool tiled = env_int("CSGN_RAGGED_FLAT", 0) == 0 && max_t1 * max_t2 * U > 8192 &&
                       batch * max_t1 * max_t2 <= 2 * total_out_terms;
    if (tiled) {
        MulArgs a = {};
        a.L = L;
        a.R = R;
        a.out = out;
        a.offL = offL;
        a.offR = offR;
        a.offOut = offOut;
        a.t1 = (u32)max_t1;
        a.t2 = (u32)max_t2;
        return wide ? launch_tiled<unit16, true>(a, batch, U, s) : launch_tiled<unit8, true>(a, batch, U, s);
    }
    const u64 total_units = total_out_terms * U;
    const FastDiv dU = csgn_fastdiv_make(U);
    const int chunks = ragged_chunks(total_units);
    const u32 pf_pairs = (u32)env_int("CSGN_RAGGED_PF", 32);    // operand prefetch distance in pairs, 0 = off
    const u64 per_launch = kMaxBlocks256 * 256u;         // units: a multiple of every 256*C
    for (u64 u0 = 0; u0 < total_units; u0 += per_launch) {
        const u64 nu = (total_units - u0 < per_launch) ? total_units - u0 : per_launch;
        const u32 blocks = ceil_div_u64(nu, 256u * (u32)chunks);
#define CSGN_RAGGED_LAUNCH(CH)                                                                      \
    do {                                                                                            \
        if (wide)                                                                                   \
            k_mul_ragged_flat<unit16, CH><<<blocks, 256, 0, s>>>(                                  \
                reinterpret_cast<const unit16 *>(L), offL, reinterpret_cast<const unit16 *>(R), offR, \
                reinterpret_cast<unit16 *>(out), offOut, (u32)batch, u0, u0 + nu, U, dU, pf_pairs); \
        else                                                                                        \
            k_mul_ragged_flat<unit8, CH><<<blocks, 256, 0, s>>>(L, offL, R, offR, out, offOut,     \
                                                                (u32)batch, u0, u0 + nu, U, dU,     \
                                                                pf_pairs);                          \
    } while (0)
        switch (chunks) {
        case 1: CSGN_RAGGED_LAUNCH(1); break;
        case 2: CSGN_RAGGED_LAUNCH(2); break;
        case 4: CSGN_RAGGED_LAUNCH(4); break;
        case 16: CSGN_RAGGED_LAUNCH(16); break;
        default: CSGN_RAGGED_LAUNCH(8); break;
        }
#undef CSGN_RAGGED_LAUNCH
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess)
            return le;
    }
    return hipSuccess;
}

hipError_t add_uniform(u64 n_bits, u64 batch, u64 t1, u64 t2, const u64 *L, const u64 *R, u64 *out,
                       hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0 || t1 + t2 == 0)
        return hipSuccess;
    const bool wide = (dL % 2 == 0) && aligned16(L) && aligned16(R) && aligned16(out);
    const u32 U = (u32)(wide ? dL / 2 : dL);
    const u64 OU = (t1 + t2) * U;
    const u64 pairs_per = (0xFFFFFF00ull / OU) ? (0xFFFFFF00ull / OU) : 1;       // units (= threads) per launch < 2^32
    const FastDiv d = csgn_fastdiv_make((u32)OU);
    for (u64 p0 = 0; p0 < batch; p0 += pairs_per) {
        const u64 np = (batch - p0 < pairs_per) ? batch - p0 : pairs_per;
        const u32 tot = (u32)(np * OU);
        const u32 blocks = ceil_div_u64(tot, 256u);
        if (wide)
            k_add_flat<unit16, true><<<blocks, 256, 0, s>>>(
                reinterpret_cast<const unit16 *>(L) + p0 * t1 * U,
                reinterpret_cast<const unit16 *>(R) + p0 * t2 * U,
                reinterpret_cast<unit16 *>(out) + p0 * OU, tot, (u32)(t1 * U), (u32)(t2 * U), d);
        else
            k_add_flat<unit8, true><<<blocks, 256, 0, s>>>(L + p0 * t1 * U, R + p0 * t2 * U,
                                                           out + p0 * OU, tot, (u32)(t1 * U),
                                                           (u32)(t2 * U), d);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

hipError_t add_ragged(u64 n_bits, u64 batch, const u64 *L, const u64 *offL, const u64 *R,
                      const u64 *offR, u64 *out, u64 *offOut, u64 total_terms_out, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch >= (1ull << 32))
        return hipErrorInvalidValue;
    k_off_sum<<<ceil_div_u64(batch + 1, 256), 256, 0, s>>>(batch + 1, offL, offR, offOut);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || batch == 0 || total_terms_out == 0)
        return e;
    const bool wide = (dL % 2 == 0) && aligned16(L) && aligned16(R) && aligned16(out);
    const u32 U = (u32)(wide ? dL / 2 : dL);
    const u64 total_units = total_terms_out * U;
    const FastDiv dU = csgn_fastdiv_make(U);
    const int chunks = ragged_chunks(total_units);
    const u64 per_launch = kMaxBlocks256 * 256u;         // units: a multiple of every 256*C
    for (u64 u0 = 0; u0 < total_units; u0 += per_launch) {
        const u64 nu = (total_units - u0 < per_launch) ? total_units - u0 : per_launch;
        const u32 blocks = ceil_div_u64(nu, 256u * (u32)chunks);
#define CSGN_RAGGED_LAUNCH(CH)                                                                      \
    do {                                                                                            \
        if (wide)                                                                                   \
            k_add_ragged_flat<unit16, CH><<<blocks, 256, 0, s>>>(                                  \
                reinterpret_cast<const unit16 *>(L), offL, reinterpret_cast<const unit16 *>(R), offR, \
                reinterpret_cast<unit16 *>(out), offOut, (u32)batch, u0, u0 + nu, U, dU);           \
        else                                                                                        \
            k_add_ragged_flat<unit8, CH><<<blocks, 256, 0, s>>>(L, offL, R, offR, out, offOut,     \
                                                                (u32)batch, u0, u0 + nu, U, dU);    \
    } while (0)
        switch (chunks) {
        case 1: CSGN_RAGGED_LAUNCH(1); break;
        case 2: CSGN_RAGGED_LAUNCH(2); break;
        case 4: CSGN_RAGGED_LAUNCH(4); break;
        case 16: CSGN_RAGGED_LAUNCH(16); break;
        default: CSGN_RAGGED_LAUNCH(8); break;
        }
#undef CSGN_RAGGED_LAUNCH
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess)
            return le;
    }
    return hipSuccess;
}

static size_t decrypt_bitmap_bytes(u64 total_terms)
{
    return (size_t)((total_terms + 255) / 256) * 32u + 64u;
}

size_t decrypt_scratch_bytes(u64 batch, u64 total_terms)
{
    // [hit bitmap, one bit per term | pad][one u32 partial parity per ciphertext]
    return decrypt_bitmap_bytes(total_terms) + (size_t)batch * 4u + 16u;
}

hipError_t decrypt(u64 n_bits, u64 batch, u64 terms_uniform, u64 total_terms, const u64 *terms,
                   const u64 *off, const u64 *mask, uint8_t *bits, void *scratch, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0)
        return hipSuccess;
    u64 *hits = reinterpret_cast<u64 *>(scratch);
    if (total_terms) {
        const bool wide = (dL % 2 == 0) && aligned16(terms) && aligned16(mask);
        const u32 U = (u32)(wide ? dL / 2 : dL);
        const u64 blocks64 = (total_terms + 255) / 256;
        if (blocks64 > kMaxBlocks256)
            return hipErrorInvalidValue;
        const u32 blocks = (u32)blocks64;
        const FastDiv dU = csgn_fastdiv_make(U);
        // fast form: K 4-KiB segments per workgroup holding TB whole terms (TB % 8 == 0);
        // term sizes that need K > 8 (or U > 64) use the looping form
        int k_seg = 0;
        if (U <= 64u)
            for (int k = 1; k <= 8; ++k)
                if ((256u * k) % U == 0 && ((256u * k) / U) % 8u == 0) {
                    k_seg = k;
                    break;
                }
        if (k_seg && env_int("CSGN_DEC_LOOP", 0) == 0) {
            const u32 tb = 256u * k_seg / U;
            const u64 nblk = (total_terms + tb - 1) / tb;
            if (nblk > kMaxBlocks256)
                return hipErrorInvalidValue;
            unsigned char *hb = reinterpret_cast<unsigned char *>(scratch);
            const u64 tu = total_terms * U;
            // fresh (single-term) ciphertexts in a uniform batch: pass 1 writes the plaintext
            // bytes itself and pass 2 is skipped
            uint8_t *direct = (!off && terms_uniform == 1) ? bits : nullptr;
#define CSGN_HITS_SEG(K)                                                                              \
    do {                                                                                              \
        if (wide)                                                                                     \
            k_term_hits_seg<unit16, K><<<(u32)nblk, 256, 0, s>>>(                                     \
                reinterpret_cast<const unit16 *>(terms), reinterpret_cast<const unit16 *>(mask), tu, U, \
                dU, tb, hb, direct);                                                                  \
        else                                                                                          \
            k_term_hits_seg<unit8, K><<<(u32)nblk, 256, 0, s>>>(terms, mask, tu, U, dU, tb, hb, direct); \
    } while (0)
            switch (k_seg) {
            case 1: CSGN_HITS_SEG(1); break;
            case 2: CSGN_HITS_SEG(2); break;
            case 3: CSGN_HITS_SEG(3); break;
            case 4: CSGN_HITS_SEG(4); break;
            case 5: CSGN_HITS_SEG(5); break;
            case 6: CSGN_HITS_SEG(6); break;
            case 7: CSGN_HITS_SEG(7); break;
            default: CSGN_HITS_SEG(8); break;
            }
#undef CSGN_HITS_SEG
            if (direct)
                return hipGetLastError();
        } else if (wide)
            k_term_hits<unit16><<<blocks, 256, (size_t)U * 16 + 1024, s>>>(
                reinterpret_cast<const unit16 *>(terms), reinterpret_cast<const unit16 *>(mask),
                total_terms, U, dU, hits);
        else
            k_term_hits<unit8><<<blocks, 256, (size_t)U * 8 + 1024, s>>>(terms, mask, total_terms, U,
                                                                        dU, hits);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return e;
    }
    if (off) {
        // ragged: short ciphertexts one lane each, long ones one wave each
        if (batch * 64 > kMaxBlocks256 * 256u)
            return hipErrorInvalidValue;
        k_hits_parity<1, 1><<<ceil_div_u64(batch, 256), 256, 0, s>>>(hits, off, 0, batch, bits);
        if (total_terms > kLongTerms)
            k_hits_parity<64, 2><<<ceil_div_u64(batch * 64, 256), 256, 0, s>>>(hits, off, 0, batch, bits);
    } else if (terms_uniform <= kLongTerms) {
        k_hits_parity<1, 0><<<ceil_div_u64(batch, 256), 256, 0, s>>>(hits, nullptr, terms_uniform, batch, bits);
    } else if (batch * ((terms_uniform + 65535) / 65536) <= kMaxBlocks256) {
        // long uniform ciphertexts: chunked fold + one atomicXor per (ciphertext, chunk)
        u32 *partial = reinterpret_cast<u32 *>(reinterpret_cast<unsigned char *>(scratch) +
                                               decrypt_bitmap_bytes(total_terms));
        hipError_t e = hipMemsetAsync(partial, 0, (size_t)batch * 4u, s);
        if (e != hipSuccess)
            return e;
        const u32 chunks = (u32)((terms_uniform + 65535) / 65536);
        k_hits_parity_chunked<<<(u32)(batch * chunks), 256, 0, s>>>(hits, terms_uniform, chunks, partial);
        k_partial_to_bits<<<ceil_div_u64(batch, 256), 256, 0, s>>>(partial, batch, bits);
    } else {
        if (batch * 64 > kMaxBlocks256 * 256u)
            return hipErrorInvalidValue;
        k_hits_parity<64, 0><<<ceil_div_u64(batch * 64, 256), 256, 0, s>>>(hits, nullptr, terms_uniform, batch, bits);
    }
    return hipGetLastError();
}

// dec(a*b) = dec(a) & dec(b) and dec(a+b) = dec(a) ^ dec(b): a product term L_i & R_j covers
// the key mask iff both factors do, so the number of hitting product terms is
// hits(L)*hits(R) and its parity the AND of the parities; concatenation adds the counts.
// The 168 MB product of a 1024x1024 pair is therefore never materialised when only its
// plaintext is wanted: 2 x 160 KB are read instead.
__global__ void __launch_bounds__(256) k_combine_bits(const uint8_t *__restrict__ a,
                                                      const uint8_t *__restrict__ b, u64 n, int is_product,
                                                      uint8_t *__restrict__ out)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n)
        out[i] = is_product ? (a[i] & b[i] & 1u) : ((a[i] ^ b[i]) & 1u);
}

size_t decrypt_combined_scratch_bytes(u64 batch, u64 t1, u64 t2)
{
    const size_t pad = 256;
    return decrypt_scratch_bytes(batch, batch * t1) + decrypt_scratch_bytes(batch, batch * t2) +
           2 * (((size_t)batch + pad - 1) / pad * pad) + 4 * pad;
}

hipError_t decrypt_combined(u64 n_bits, u64 batch, u64 t1, u64 t2, const u64 *L, const u64 *R,
                            const u64 *mask, bool is_product, uint8_t *bits, void *scratch, hipStream_t s)
{
    if (batch == 0)
        return hipSuccess;
    const size_t pad = 256;
    auto up = [&](size_t x) { return (x + pad - 1) / pad * pad; };
    unsigned char *base = reinterpret_cast<unsigned char *>(scratch);
    base = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(base)));
    unsigned char *s1 = base;
    unsigned char *s2 = s1 + up(decrypt_scratch_bytes(batch, batch * t1));
    uint8_t *b1 = s2 + up(decrypt_scratch_bytes(batch, batch * t2));
    uint8_t *b2 = b1 + up(batch);
    hipError_t e = decrypt(n_bits, batch, t1, batch * t1, L, nullptr, mask, b1, s1, s);
    if (e != hipSuccess)
        return e;
    e = decrypt(n_bits, batch, t2, batch * t2, R, nullptr, mask, b2, s2, s);
    if (e != hipSuccess)
        return e;
    k_combine_bits<<<ceil_div_u64(batch, 256), 256, 0, s>>>(b1, b2, batch, is_product ? 1 : 0, bits);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// EXTENSION (not reference behaviour): mod-2 compaction of term lists.
// Decryption XORs over terms, so a term occurring an even number of times contributes
// nothing and one occurring an odd number of times contributes once: each ciphertext is
// rewritten as its distinct odd-multiplicity terms, in order of first occurrence.  The
// reference never does this (add is pure concatenation), so it is opt-in and never runs on a
// parity path.  Method: per-ciphertext open-addressing hash table in HBM keyed by a 64-bit
// hash of the term; representative = smallest index with that key; every term is compared
// in full against its representative (a hash collision between different terms just keeps
// the colliding term unmerged, it can never merge unequal terms); parity by atomicXor;
// survivors are compacted with a prefix sum.  Deterministic output.
// ---------------------------------------------------------------------------------------
struct CompactView {
    u64 *keys;      // 2 slots per term
    u32 *rep;       // per slot: smallest term index holding the key
    u32 *parity;    // per slot: multiplicity mod 2 of the representative's value
    u32 *slot_of;   // per term: its slot (global slot index)
    u32 *keep;      // per term: 1 = survives
    u64 *scan;      // per term: exclusive prefix sum of keep (+1 total at [total])
};

__device__ inline u64 term_hash(const u64 *t, u32 dL)
{
    u64 h = 0x243F6A8885A308D3ull;
    for (u32 k = 0; k < dL; ++k)
        h = csgn_splitmix64(h ^ t[k]);
    return h | 1ull;                                 // 0 is the empty-slot marker
}

// which ciphertext owns global term index g (CSR offsets, batch >= 1)
__device__ inline u32 owner_of_term(const u64 *off, u32 batch, u64 g)
{
    u32 lo = 0, hi = batch;                          // off[lo] <= g < off[hi]
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (off[mid] <= g)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256) k_compact_insert(const u64 *__restrict__ terms,
                                                        const u64 *__restrict__ off, u32 batch,
                                                        u64 total, u32 dL, CompactView v)
{
    const u64 g = (u64)blockIdx.x * 256u + threadIdx.x;
    if (g >= total)
        return;
    const u32 b = owner_of_term(off, batch, g);
    const u64 base = 2 * off[b], nslots = 2 * (off[b + 1] - off[b]);
    const u64 key = term_hash(terms + g * dL, dL);
    u64 slot = key % nslots;
    for (;;) {
        const u64 old = atomicCAS(reinterpret_cast<unsigned long long *>(v.keys + base + slot), 0ull, key);
        if (old == 0ull || old == key)
            break;
        slot = slot + 1 == nslots ? 0 : slot + 1;    // table is at most half full: terminates
    }
    v.slot_of[g] = (u32)(base + slot);
    atomicMin(v.rep + base + slot, (u32)(g - off[b]));
}

__global__ void __launch_bounds__(256) k_compact_match(const u64 *__restrict__ terms,
                                                       const u64 *__restrict__ off, u32 batch,
                                                       u64 total, u32 dL, CompactView v)
{
    const u64 g = (u64)blockIdx.x * 256u + threadIdx.x;
    if (g >= total)
        return;
    const u32 b = owner_of_term(off, batch, g);
    const u32 slot = v.slot_of[g];
    const u64 r = off[b] + v.rep[slot];              // global index of the representative
    bool same = true;
    if (r != g) {
        const u64 *x = terms + g * dL, *y = terms + r * dL;
        for (u32 k = 0; k < dL; ++k)
            same = same && (x[k] == y[k]);
    }
    if (same)
        atomicXor(v.parity + slot, 1u);
    v.keep[g] = same ? 2u : 1u;                      // 2 = decided by the slot parity, 1 = collision survivor
}

__global__ void __launch_bounds__(256) k_compact_decide(const u64 *__restrict__ off, u32 batch, u64 total,
                                                        CompactView v)
{
    const u64 g = (u64)blockIdx.x * 256u + threadIdx.x;
    if (g >= total)
        return;
    const u32 b = owner_of_term(off, batch, g);
    const u32 slot = v.slot_of[g];
    u32 k = v.keep[g];
    if (k == 2u)
        k = (off[b] + v.rep[slot] == g && (v.parity[slot] & 1u)) ? 1u : 0u;
    v.keep[g] = k;
}

// exclusive scan of keep[] (one workgroup, chunked) + compacted CSR offsets
__global__ void __launch_bounds__(1024) k_compact_scan(u64 total, CompactView v)
{
    __shared__ u64 part[1024];
    const u32 tid = threadIdx.x;
    const u64 chunk = (total + 1023) / 1024;
    const u64 g0 = min(total, (u64)tid * chunk), g1 = min(total, g0 + chunk);
    u64 sum = 0;
    for (u64 g = g0; g < g1; ++g)
        sum += v.keep[g];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        u64 run = 0;
        for (u32 t = 0; t < 1024; ++t) {
            const u64 x = part[t];
            part[t] = run;
            run += x;
        }
        v.scan[total] = run;
    }
    __syncthreads();
    u64 run = part[tid];
    for (u64 g = g0; g < g1; ++g) {
        v.scan[g] = run;
        run += v.keep[g];
    }
}

// compacted CSR offsets (separate launch: reads scan[] written by the whole scan workgroup)
__global__ void __launch_bounds__(256) k_compact_offsets(const u64 *__restrict__ off, u32 batch,
                                                         CompactView v, u64 *__restrict__ off_out)
{
    const u32 b = blockIdx.x * 256u + threadIdx.x;
    if (b <= batch)
        off_out[b] = v.scan[off[b]];                 // off[batch] == total
}

__global__ void __launch_bounds__(256) k_compact_scatter(const u64 *__restrict__ terms, u64 total_words,
                                                         u32 dL, CompactView v, u64 *__restrict__ out)
{
    const u64 w = (u64)blockIdx.x * 256u + threadIdx.x;
    if (w >= total_words)
        return;
    const u64 g = w / dL;
    if (v.keep[g])
        out[v.scan[g] * dL + (w - g * dL)] = terms[w];
}

size_t compact_scratch_bytes(u64 total_terms)
{
    // keys 16 B + rep 8 B + parity 8 B (2 slots per term) + slot_of 4 + keep 4 + scan 8 (+1)
    return (size_t)total_terms * (16 + 8 + 8 + 4 + 4 + 8) + 8 + 6 * 256;
}

hipError_t compact(u64 n_bits, u64 batch, u64 total_terms, const u64 *terms, const u64 *off, u64 *out,
                   u64 *off_out, void *scratch, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0)
        return hipSuccess;
    if (batch >= (1ull << 31) || total_terms > kMaxBlocks256 * 256u - 256u)
        return hipErrorInvalidValue;
    auto up = [](uintptr_t x) { return (x + 255) & ~(uintptr_t)255; };
    unsigned char *p = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(scratch)));
    CompactView v;
    v.keys = reinterpret_cast<u64 *>(p);
    p = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(p + total_terms * 16)));
    v.rep = reinterpret_cast<u32 *>(p);
    p = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(p + total_terms * 8)));
    v.parity = reinterpret_cast<u32 *>(p);
    p = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(p + total_terms * 8)));
    v.slot_of = reinterpret_cast<u32 *>(p);
    p = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(p + total_terms * 4)));
    v.keep = reinterpret_cast<u32 *>(p);
    p = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(p + total_terms * 4)));
    v.scan = reinterpret_cast<u64 *>(p);
    hipError_t e;
    if (total_terms) {
        if ((e = hipMemsetAsync(v.keys, 0, total_terms * 16, s)) != hipSuccess)
            return e;
        if ((e = hipMemsetAsync(v.rep, 0xFF, total_terms * 8, s)) != hipSuccess)
            return e;
        if ((e = hipMemsetAsync(v.parity, 0, total_terms * 8, s)) != hipSuccess)
            return e;
        const u32 blocks = ceil_div_u64(total_terms, 256);
        k_compact_insert<<<blocks, 256, 0, s>>>(terms, off, (u32)batch, total_terms, (u32)dL, v);
        k_compact_match<<<blocks, 256, 0, s>>>(terms, off, (u32)batch, total_terms, (u32)dL, v);
        k_compact_decide<<<blocks, 256, 0, s>>>(off, (u32)batch, total_terms, v);
    }
    k_compact_scan<<<1, 1024, 0, s>>>(total_terms, v);
    k_compact_offsets<<<ceil_div_u64(batch + 1, 256), 256, 0, s>>>(off, (u32)batch, v, off_out);
    if (total_terms) {
        const u64 words = total_terms * dL;
        const u64 blocks64 = (words + 255) / 256;
        if (blocks64 > kMaxBlocks256)
            return hipErrorInvalidValue;
        k_compact_scatter<<<(u32)blocks64, 256, 0, s>>>(terms, words, (u32)dL, v, out);
    }
    return hipGetLastError();
}

hipError_t encrypt(u64 n_bits, u64 d, u64 batch, const uint8_t *plain, const u64 *rnd,
                   const u32 *chosen, const uint8_t *last, const u64 *key, const u64 *mask, u64 seed,
                   bool device_rng, u64 *out, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0)
        return hipSuccess;
    // fast form: K aligned 4 KiB segments = TB whole ciphertexts per workgroup, one lane per unit
    // (a wave-local variant -- whole ciphertexts per wave, ballots only, no LDS or barrier -- measured
    // slower: 2.6 vs 3.3 TB/s device-RNG at N=1247; its 960-byte wave stores lose more than the
    // barriers cost)
    {
        const bool wide = (dL % 2 == 0) && aligned16(out) && aligned16(mask) && (device_rng || aligned16(rnd));
        const u32 U = (u32)(wide ? dL / 2 : dL);
        int k_seg = 0;
        if (U <= 64u)
            for (int k = 1; k <= 8; ++k)
                if ((256u * k) % U == 0 && (256u * k) / U <= 256u) {
                    k_seg = k;
                    break;
                }
        if (k_seg && env_int("CSGN_ENC_LDS", 0) == 0) {
            const u32 tb = 256u * k_seg / U;
            const u64 nblk = (batch + tb - 1) / tb;
            if (nblk > kMaxBlocks256)
                return hipErrorInvalidValue;
            const FastDiv dU = csgn_fastdiv_make(U);
#define CSGN_ENC_SEG(UNIT, K)                                                                           \
    do {                                                                                                \
        if (device_rng)                                                                                 \
            k_encrypt_seg<UNIT, K, true><<<(u32)nblk, 256, 0, s>>>(                                      \
                n_bits, (u32)dL, U, dU, d, batch, tb, plain, reinterpret_cast<const UNIT *>(rnd), chosen, \
                last, key, reinterpret_cast<const UNIT *>(mask), seed, reinterpret_cast<UNIT *>(out));   \
        else                                                                                            \
            k_encrypt_seg<UNIT, K, false><<<(u32)nblk, 256, 0, s>>>(                                     \
                n_bits, (u32)dL, U, dU, d, batch, tb, plain, reinterpret_cast<const UNIT *>(rnd), chosen, \
                last, key, reinterpret_cast<const UNIT *>(mask), seed, reinterpret_cast<UNIT *>(out));   \
    } while (0)
#define CSGN_ENC_SEG_K(UNIT)                                       \
    switch (k_seg) {                                               \
    case 1: CSGN_ENC_SEG(UNIT, 1); break;                          \
    case 2: CSGN_ENC_SEG(UNIT, 2); break;                          \
    case 3: CSGN_ENC_SEG(UNIT, 3); break;                          \
    case 4: CSGN_ENC_SEG(UNIT, 4); break;                          \
    case 5: CSGN_ENC_SEG(UNIT, 5); break;                          \
    case 6: CSGN_ENC_SEG(UNIT, 6); break;                          \
    case 7: CSGN_ENC_SEG(UNIT, 7); break;                          \
    default: CSGN_ENC_SEG(UNIT, 8); break;                         \
    }
            if (wide) {
                CSGN_ENC_SEG_K(unit16)
            } else {
                CSGN_ENC_SEG_K(unit8)
            }
#undef CSGN_ENC_SEG_K
#undef CSGN_ENC_SEG
            return hipGetLastError();
        }
    }
    // general form (term sizes that do not pack): ciphertexts staged in LDS
    u32 cb = 64;
    while (cb > 1 && (u64)cb * dL * 8 > 32768)
        cb /= 2;
    const size_t lds = ((size_t)cb * dL + dL) * 8;
    const u64 blocks64 = (batch + cb - 1) / cb;
    if (blocks64 > kMaxBlocks256)
        return hipErrorInvalidValue;
    const FastDiv ddL = csgn_fastdiv_make((u32)dL);
    if (device_rng)
        k_encrypt<true><<<(u32)blocks64, 256, lds, s>>>(n_bits, (u32)dL, d, batch, cb, ddL, plain, rnd,
                                                        chosen, last, key, mask, seed, out);
    else
        k_encrypt<false><<<(u32)blocks64, 256, lds, s>>>(n_bits, (u32)dL, d, batch, cb, ddL, plain, rnd,
                                                         chosen, last, key, mask, seed, out);
    return hipGetLastError();
}

hipError_t permute(u64 n_bits, u64 batch, u64 terms_in, bool per_term, const u64 *terms,
                   const u32 *perm, u64 *out, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    const u64 out_terms = per_term ? batch * terms_in : batch;
    if (out_terms == 0)
        return hipSuccess;
    const u64 stride = per_term ? dL : terms_in * dL;
    // bit-plane form (64 terms per wave) unless the batch is too small to fill a wave or the LDS
    // image (rows + planes) would not fit; CSGN_PERM_BALLOT=1 forces the ballot form
    {
        const u64 dLp = (dL + kPermUnroll - 1) / kPermUnroll * kPermUnroll;
        const size_t lds = ((size_t)64 * (dLp | 1) + dLp * 64 + 1) * 8 + dLp * 64 * 2;
        if (terms_in != 0 && out_terms >= 16 && dL <= 64 && lds <= 160 * 1024 && !env_int("CSGN_PERM_BALLOT", 0)) {
            const u64 groups = (out_terms + 63) / 64;
            // persistent waves: as many as the chip holds at once, equal group counts per wave
            int cus = 256;
            {
                int dev = 0;
                if (hipGetDevice(&dev) == hipSuccess)
                    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            }
            // 16-byte staging accesses when every term starts 16-byte aligned
            const bool wide = dL % 2 == 0 && stride % 2 == 0 && (((uintptr_t)terms | (uintptr_t)out) & 15) == 0 &&
                              !env_int("CSGN_PERM_NARROW", 0);
            const u32 Ud = (u32)(wide ? dL / 2 : dL);
            const FastDiv dUd = csgn_fastdiv_make(Ud);
#define CSGN_PLANES_LAUNCH(LQ, PIPE, UW)                                                            \
    do {                                                                                            \
        if (lds > 65536) {      /* beyond the default dynamic-LDS window (N > ~3500 bits) */        \
            hipError_t e = hipFuncSetAttribute(                                                     \
                reinterpret_cast<const void *>(&k_permute_planes<LQ, PIPE, UW>),                    \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
            if (e != hipSuccess)                                                                    \
                return e;                                                                           \
        }                                                                                           \
        /* how many of these waves a CU really holds: the runtime's answer, and LDS handed out in  \
           granules (measured: 7 x 23.3 KB is reported to fit 160 KB but the seventh wave runs after \
           the other six).  A wave beyond that number would start when the rest have finished. */   \
        int per_cu = 0;                                                                             \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_permute_planes<LQ, PIPE, UW>, 64, \
                                                         lds) != hipSuccess || per_cu < 1)          \
            per_cu = 1;                                                                             \
        per_cu = std::max(1, std::min(per_cu, (int)(160 * 1024 / ((lds + 1023) / 1024 * 1024))));   \
        if (const int cap = env_int("CSGN_PERM_WAVES", 0))                                          \
            per_cu = std::min(per_cu, cap);                                                         \
        const u64 resident = (u64)cus * (u64)per_cu;                                                \
        const u64 rounds = (groups + resident - 1) / resident;                                      \
        const u32 grid = (u32)((groups + rounds - 1) / rounds);                                     \
        k_permute_planes<LQ, PIPE, UW><<<grid, 64, lds, s>>>(n_bits, (u32)dL, dUd, out_terms,       \
                                                             stride, terms, perm, out);             \
    } while (0)
            if (wide) {
                if (Ud <= 4)
                    CSGN_PLANES_LAUNCH(4, true, 2);
                else if (Ud <= 10)
                    CSGN_PLANES_LAUNCH(10, true, 2);
                else if (Ud <= 16)
                    CSGN_PLANES_LAUNCH(16, true, 2);
                else
                    CSGN_PLANES_LAUNCH(32, false, 2);
            } else {
                if (Ud <= 4)
                    CSGN_PLANES_LAUNCH(4, true, 1);
                else if (Ud <= 8)
                    CSGN_PLANES_LAUNCH(8, true, 1);
                else if (Ud <= 20)
                    CSGN_PLANES_LAUNCH(20, true, 1);
                else if (Ud <= 32)
                    CSGN_PLANES_LAUNCH(32, true, 1);
                else
                    CSGN_PLANES_LAUNCH(64, false, 1);
            }
#undef CSGN_PLANES_LAUNCH
            return hipGetLastError();
        }
    }
    // terms per workgroup: a multiple of the 4 waves, LDS image <= 32 KB
    u32 tb = 64;
    while (tb > 4 && (u64)tb * dL * 8 > 32768)
        tb /= 2;
    const u64 blocks64 = (out_terms + tb - 1) / tb;
    if (blocks64 > kMaxBlocks256)
        return hipErrorInvalidValue;
    const FastDiv ddL = csgn_fastdiv_make((u32)dL);
    const u32 have = terms_in != 0 ? 1u : 0u;
    const size_t lds = (size_t)tb * dL * 8;
#define CSGN_PERMUTE_LAUNCH(NW)                                                                     \
    k_permute<NW><<<(u32)blocks64, 256, lds, s>>>(n_bits, (u32)dL, ddL, out_terms, stride, have, tb, \
                                                  terms, perm, out)
    if (dL <= 4)
        CSGN_PERMUTE_LAUNCH(4);
    else if (dL <= 8)
        CSGN_PERMUTE_LAUNCH(8);
    else if (dL <= 16)
        CSGN_PERMUTE_LAUNCH(16);
    else if (dL <= 20)
        CSGN_PERMUTE_LAUNCH(20);
    else if (dL <= 32 || dL % 32 == 0 && dL % 64 != 0)
        CSGN_PERMUTE_LAUNCH(32);
    else
        CSGN_PERMUTE_LAUNCH(64);
#undef CSGN_PERMUTE_LAUNCH
    return hipGetLastError();
}

hipError_t synth_fill(u64 seed, u64 n_bits, u64 first_word, u64 n_words, u64 *out, hipStream_t s)
{
    if (n_words == 0)
        return hipSuccess;
    const u64 dL = (n_bits + 63) / 64;
    const u32 rem = (u32)(n_bits & 63);
    const u64 tail = rem ? ~0ull << (64 - rem) : ~0ull;
    const u64 want = (n_words + 255) / 256;
    const u32 blocks = (u32)(want < 16384 ? want : 16384);
    k_synth_fill<<<blocks, 256, 0, s>>>(seed, (u32)dL, tail, first_word, n_words, out);
    return hipGetLastError();
}

hipError_t digest(const u64 *w, u64 n_words, u64 first_index, u64 *d_digest, hipStream_t s)
{
    if (n_words == 0)
        return hipSuccess;
    const u64 want = (n_words + 255) / 256;
    const u32 blocks = (u32)(want < 8192 ? want : 8192);
    k_digest<<<blocks, 256, 0, s>>>(w, n_words, first_index, d_digest);
    return hipGetLastError();
}

} // namespace csgn
