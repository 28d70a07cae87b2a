// csgn_device.h -- helpers shared by the kernel translation units (csgn_mul.hip, csgn_add.hip,
// csgn_decrypt.hip, csgn_encrypt.hip, csgn_permute.hip, csgn_compact.hip, csgn_harness.hip):
// 16-/8-byte unit access, the XCD-contiguous block order, CSR pair search, launch limits and the
// knob-dependent launch choices (csgn_tuning.h).  Everything has internal linkage (one copy per translation unit).
//
// Common shape of the data path: lanes own consecutive 16-byte units so every wave-level
// load/store is one global_{load,store}_dwordx4 covering 1 KiB of contiguous, 128-B-aligned
// HBM.  All of it is bitwise integer work bound by HBM bandwidth: there is no MFMA anywhere.
// Design notes live in DESIGN.md; reference citations (/root/reference/...) name the scalar loop
// each kernel replaces.
#pragma once

#include "csgn_kernels.h"
#include "csgn_tuning.h"

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

// The same with the XCDs taking GROUPS of `group` consecutive logical blocks in turn (XCD x owns groups x, x + 8, ...)
// instead of one contiguous eighth each: every XCD still writes runs of group x 4-16 KiB, but a stretch of the output
// whose blocks are slow (the single-term pairs behind one huge pair of a skewed ragged batch: latency-bound workgroups)
// is spread over all eight XCDs instead of being the tail of ONE (round 5: 640 such workgroups, 6 % of the output,
// were 2.5 rounds of 32 CUs at the end of a launch the other 224 CUs had left).  group = 0: the contiguous eighths.
// Bijective for any grid size: the blocks past the last whole round of 8 groups keep their own index.
__device__ inline u32 xcd_grouped_block(u32 b, u32 nblocks, u32 group)
{
    if (group == 0u)
        return xcd_contiguous_block(b, nblocks);
    const u32 round = 8u * group, full = nblocks - nblocks % round;
    if (b >= full)
        return b;
    const u32 x = b & 7u, q = b >> 3;                            // XCD, index inside the XCD
    return ((q / group) * 8u + x) * group + q % group;
}

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

// The same answer as csr_find(off, lo, hi, term), computed by a whole WAVE: a 64-ary search, every
// step one load per lane (64 probes spread over the range) and one __ballot -- log64(batch) dependent
// round trips (3 for 65 536 pairs, 4 for 1 M) where the binary search makes 17 to 20.  The start-up
// search of a ragged workgroup was ~10 us of a lifetime in which it writes 16 KiB (~5 us at full
// rate): with few workgroup rounds (a 178 MB output) nothing hid it (round 2: 2.7 TB/s on one
// 1024x1024 pair among 65 535 singles).  Requires off[lo] <= term; every lane returns the result.
__device__ inline u32 wave_find(const u64 *__restrict__ off, u32 lo, u32 hi, u64 term)
{
    const u32 lane = threadIdx.x & (kWave - 1);
    while (hi - lo > 1) {
        const u32 n = hi - lo, s = (n + kWave - 1) / kWave;
        const u64 idx = (u64)lo + (u64)lane * s;
        const bool ok = idx < hi && off[idx] <= term;          // monotone in the lane number; lane 0 holds by the invariant
        const u32 c = (u32)__popcll(__ballot(ok));             // >= 1
        lo += (c - 1u) * s;
        hi = min(lo + s, hi);
    }
    return lo;
}

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

// ---------------------------------------------------------------------- chained scan across workgroups
constexpr u64 kFlagAggregate = 1ull << 62, kFlagPrefix = 2ull << 62, kValueMask = (1ull << 62) - 1;

// Decoupled look-back (one wave): publish this group's count, add up the counts of the groups before
// it back to the nearest one whose inclusive prefix is known, publish the own inclusive prefix.  Every
// granule is ONE 8-byte agent-scope store/load carrying flag and value together, so no ordering
// between data and flag is needed.  Returns the exclusive prefix in every lane.
__device__ inline u64 lookback(u64 *status, u32 gid, u64 count)
{
    const u32 lane = threadIdx.x & (kWave - 1);
    if (lane == 0)
        __hip_atomic_store(status + gid, kFlagAggregate | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u64 excl = 0;
    long long base = (long long)gid - 1;
    while (base >= 0) {
        const long long idx = base - (long long)lane;
        const u64 v = idx >= 0 ? __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : kFlagPrefix;              // before group 0: prefix 0
        const u32 flag = (u32)(v >> 62);
        const u64 pending = __ballot(flag == 0u), known = __ballot(flag == 2u);
        const u32 first = known ? (u32)__builtin_ctzll(known) : kWave;     // nearest lane holding a prefix
        const u64 need = first < kWave - 1 ? (2ull << first) - 1ull : ~0ull;
        if (pending & need) {
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        u64 mine = lane <= first ? (v & kValueMask) : 0ull;
        for (u32 d = 32; d > 0; d >>= 1)
            mine += __shfl_xor(mine, d, kWave);
        excl += mine;
        if (first < kWave)
            break;
        base -= kWave;
    }
    if (lane == 0)
        __hip_atomic_store(status + gid, kFlagPrefix | (excl + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// The same with one MARK bit riding along (value: 61 bits): a group may mark itself, and every group learns whether
// any group before it did.  k_plan marks chunks that hold a pair of more than one product term, so the LAST chunk
// knows from its look-back alone -- no counter of finished workgroups, no fence -- the batch total and whether every
// pair of the batch is 1 x 1.
constexpr u64 kMarkBit = 1ull << 61, kMarkedValueMask = kMarkBit - 1;
__device__ inline u64 lookback_marked(u64 *status, u32 gid, u64 count, bool mark, bool &marked_before)
{
    const u32 lane = threadIdx.x & (kWave - 1);
    const u64 own = mark ? kMarkBit : 0ull;
    if (lane == 0)
        __hip_atomic_store(status + gid, kFlagAggregate | own | (count & kMarkedValueMask), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    u64 excl = 0;
    bool before = false;
    long long base = (long long)gid - 1;
    while (base >= 0) {
        const long long idx = base - (long long)lane;
        const u64 v = idx >= 0 ? __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : kFlagPrefix;              // before group 0: prefix 0, no mark
        const u32 flag = (u32)(v >> 62);
        const u64 pending = __ballot(flag == 0u), known = __ballot(flag == 2u);
        const u32 first = known ? (u32)__builtin_ctzll(known) : kWave;     // nearest lane holding a prefix
        const u64 need = first < kWave - 1 ? (2ull << first) - 1ull : ~0ull;
        if (pending & need) {
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        u64 mine = lane <= first ? (v & kMarkedValueMask) : 0ull;
        before |= __ballot(lane <= first && (v & kMarkBit) != 0ull) != 0ull;
        for (u32 d = 32; d > 0; d >>= 1)
            mine += __shfl_xor(mine, d, kWave);
        excl += mine;
        if (first < kWave)
            break;
        base -= kWave;
    }
    if (lane == 0)
        __hip_atomic_store(status + gid, kFlagPrefix | ((before || mark) ? kMarkBit : 0ull) | ((excl + count) & kMarkedValueMask),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    marked_before = before;
    return excl;
}

// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global load, store and
// atomic the wave has in flight (s_waitcnt vmcnt(0)); where a wave has issued a global request whose result it wants
// AFTER the barrier, this one lets it stay in flight.
__device__ inline void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Zero-fill as a KERNEL: no compute path of this library calls hipMemsetAsync, so a circuit's graph
// (csgn_circuit_*) holds kernel nodes only.  Why (profiles/r05/graph_memset_*.log): circuits whose
// zero fills were captured as memset NODES have twice returned wrong data inside the long-lived test process
// (round 4: the inputs of a circuit with a compaction node read as zeros; round 5: one wrong bit in the second
// of five runs of a long-uniform decrypt), never with kernel nodes.  The cause is NOT pinned: a minimal HIP
// program (tools/graph_memset_probe.hip: memset node against uploads, fills and atomics on every stream kind,
// under the ROCm 7.2.0 runtime and under the 7.0.2 one torch bundles) orders memset nodes correctly, and the
// failing circuit alone ran 240 times clean (tools/graph_memset_case.py).  Dev knob zero_memset = 1 brings the
// memset form back for such experiments.
__global__ void __launch_bounds__(256) k_zero_words(u64 *__restrict__ p, u64 n)
{
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < n; i += (u64)gridDim.x * 256u)
        p[i] = 0ull;
}
inline hipError_t zero_words(u64 *p, u64 n, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    if (tune(TUNE_ZERO_MEMSET))                      // dev only (see above)
        return hipMemsetAsync(p, 0, n * 8, s);
    const u32 blocks = (u32)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    k_zero_words<<<blocks, 256, 0, s>>>(p, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- launch helpers
template <typename T>
inline bool aligned16(const T *p)
{
    return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

// Block order of the one-unit-per-lane stream kernels (1x1 multiply, uniform add): XCD-contiguous
// once a launch is large (measured +3-5 % from 32 M units = 512 MB per stream up, -1-2 % at
// 10 M units).  Knob stream_xcd = 0 / 1 forces it.
inline u32 stream_xcd(u64 units)
{
    const int forced = tune(TUNE_STREAM_XCD);
    if (forced == 0 || forced == 1)
        return (u32)forced;
    return units >= (1ull << 25) ? 1u : 0u;
}

// 4 KiB chunks per workgroup of the flat ragged kernels: as many as 8 (the workgroup's first search
// is paid once per C chunks) while the grid still has >= 8192 workgroups to fill the chip with.
// Knob ragged_c = 1, 2, 4, 8, 16 overrides.
inline int ragged_chunks(u64 total_units)
{
    const int forced = tune(TUNE_RAGGED_C);
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16)
        return forced;
    int c = 1;
    while (c < 8 && total_units / (256u * 2u * (u64)c) >= 8192u)
        c *= 2;
    return c;
}

} // namespace

} // namespace csgn
