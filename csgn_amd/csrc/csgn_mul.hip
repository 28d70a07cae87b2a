// csgn_mul.hip -- all-pairs multiply: 1x1 stream, flat and LDS-tiled kernels, the operand touch pass, ragged (CSR) form and its planner.
// Hand-written CDNA4 (gfx950) HIP; shared helpers in csgn_device.h, design notes in DESIGN.md.
#include "csgn_device.h"

namespace csgn {

namespace {

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
                                                    Unit *__restrict__ o, u64 n_units, u32 xcd)
{
    const u32 bid = xcd ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const u64 i = (u64)bid * 256u + threadIdx.x;
    if (i < n_units)
        unit_store<Unit, NT>(o + i, a[i] & b[i]);
}

// ---------------------------------------------------------------------------------------
// Small uniform shapes (t1*t2*U below one tile): flat map from output unit to
// (pair, left term i, right column c).  Operands are tiny and re-read through L1/L2.
// Replaces the general path of Ciphertext::multiply (src/Ciphertext.cpp:146-163).
// ---------------------------------------------------------------------------------------
// PITCH (circuit placement, csgn_circuit.hip): pair p's product starts at out + p * opitch units instead of
// p * t1*t2*U -- the producer of a sum's operand writes straight into its slice of the sum.
template <typename Unit, int MF, bool XCD, bool PITCH = false>
__global__ void __launch_bounds__(256) k_mul_flat(const Unit *__restrict__ L,
                                                  const Unit *__restrict__ R,
                                                  Unit *__restrict__ out, u32 total_units, u32 t1,
                                                  u32 t2, u32 U, FastDiv dPU, FastDiv dCU, FastDiv dU,
                                                  u32 pf_rows, u32 total_rows, u32 opitch = 0)
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
    u64 oidx[MF];
    bool pf_on = false;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
        const u32 g = min(bid * (256u * MF) + (u32)m * 256u + threadIdx.x, last);
        const u32 pair = csgn_fastdiv(g, dPU);
        const u32 rr = g - pair * PU;
        oidx[m] = PITCH ? (u64)pair * opitch + rr : 0;
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
            unit_store<Unit, true>(out + (PITCH ? oidx[m] : (u64)g), l[m] & r[m]);
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
    const u32 *pair_list;   // ragged, size classes: the launch's pairs by index into this list (nullptr: consecutive pairs)
    u32 opitch;         // uniform only: units from one pair's product to the next in `out` (0 = dense, t1*t2*U)
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
    const u32 pair = (RAGGED && a.pair_list) ? a.pair_list[a.pair_base + pair_local] : a.pair_base + pair_local;
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
        obase = a.opitch ? (u64)pair * a.opitch : (u64)pair * t1 * t2 * U;
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
        // unconditional, of a clamped column (c0 < cu here): `if (valid) r = Rp[c]` compiles to a branch, a load and a
        // vmcnt(0) per column -- the M loads of a lane one round trip after the other (tools/isa_waits.py)
        r[m] = Rp[min(c, cu - 1u)];
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

// A workgroup owns C consecutive 4 KiB chunks of the flattened output and takes them M at a time (a
// "turn").  Start: the pair of its first term by a 64-ary wave search (wave_find).  Per turn:
//   * bet on the pair the workgroup was in: its offsets are uniform addresses (scalar loads, one round
//     trip); if that pair owns the whole turn -- any turn inside a large product -- nothing else is
//     looked up;
//   * otherwise the offsets of the next 256 pairs come into LDS in ONE coalesced round trip (three
//     loads per thread) and every lane finds its pair there by an 8-step LDS binary search: no
//     per-lane chain of dependent global loads whatever the pair sizes (round 2 walked the offsets
//     lane by lane: 3.4 TB/s on a million 1x1 pairs);
//   * the 2M operand loads of a lane travel together, then M non-temporal 16-byte stores.
constexpr u32 kWin = 256;                                       // pairs per offset window
constexpr u64 kHugeTerms = 65536;        // a product of this many terms is recorded by the plan (10 MB at N=1247)
constexpr u64 kHugeRecords = 32;            // = the rows of MulPlanNotes::rec

// csgn_mul_ragged_async's huge pairs.  The planned multiply gives a pair of kHugeTerms product terms and more the uniform
// kernels on its own sub-buffers -- no lookup inside what is usually nearly all of a skewed batch's output -- but only the
// HOST can launch those, and the async call never sees the plan.  So the records the plan kernel left on the device
// ({pair, offL, offR, t1, t2, offOut}, up to kHugeRecords of them) are multiplied by the FIRST workgroups of the CSR
// kernel's own launch (mul_huge_tiles; a launch of its own in front of the CSR kernel was 29 + 18 us on the skewed batch where
// the CSR kernel alone took 40: the singles' latency-bound tail has to run BESIDE the huge pair): persistent workgroups over
// 32 KiB tiles of the records' outputs, a lane's (left term, right term, place) from two multiply-high divisions, eight
// operand pairs in flight per lane, non-temporal stores -- and the CSR kernel skips every turn that lies inside a recorded
// pair (huge_record_valid: the same test on both sides).  A record is taken if its product stays under 2^32 units and
// inside the real output (the gate).
constexpr u32 kHugeTile = 2048;                                 // units per tile
__device__ inline bool huge_record_valid(u64 t1, u64 t2, u64 o0, u32 U, u64 real_terms)
{
    const u64 c = t1 * t2;
    return c >= kHugeTerms && t1 < (1ull << 31) && t2 < (1ull << 31) && c * U < (1ull << 32) && o0 + c <= real_terms;
}
// is pair `pw` (shape t1 x t2, output at o0) one of the recorded huge pairs that mul_huge_tiles writes?
__device__ inline bool huge_pair_recorded(const u64 *__restrict__ huge, u32 pw, u64 t1, u64 t2, u64 o0, u32 U, u64 real_terms)
{
    if (!huge_record_valid(t1, t2, o0, U, real_terms))
        return false;
    const u32 n = (u32)min(huge[0], kHugeRecords);
    bool found = false;
    for (u32 i = 0; i < n; ++i)
        found = found || huge[1 + i * 6] == (u64)pw;
    return found;
}
template <typename Unit>
__device__ inline void mul_huge_tiles(const Unit *__restrict__ L, const Unit *__restrict__ R, Unit *__restrict__ out,
                                      const u64 *__restrict__ huge, const u64 *__restrict__ gate, u32 U, FastDiv dU,
                                      u32 first_tile, u32 tile_stride)
{
    constexpr int kPer = kHugeTile / 256;
    __shared__ u64 s_first[kHugeRecords + 1];
    const u64 real_terms = gate[0];
    const u32 n = real_terms ? (u32)min(huge[0], kHugeRecords) : 0u;
    if (n == 0u)
        return;
    if (threadIdx.x == 0) {
        u64 run = 0;
        for (u32 i = 0; i < n; ++i) {
            const u64 *rec = huge + 1 + i * 6;
            s_first[i] = run;
            if (huge_record_valid(rec[3], rec[4], rec[5], U, real_terms))
                run += (rec[3] * rec[4] * U + kHugeTile - 1) / kHugeTile;
        }
        s_first[n] = run;
    }
    __syncthreads();
    const u64 tiles = s_first[n];
    for (u64 tile = first_tile; tile < tiles; tile += tile_stride) {
        u32 i = 0;
        while (i + 1u < n && s_first[i + 1u] <= tile)                // (workgroup-uniform; records without tiles are passed over)
            ++i;
        const u64 *rec = huge + 1 + i * 6;
        const u64 l0 = rec[1], r0 = rec[2], t2 = rec[4], o0 = rec[5];
        const u32 cU = (u32)(rec[3] * t2 * U);
        const FastDiv d2 = csgn_fastdiv_make((u32)t2);
        const u32 rel0 = (u32)(tile - s_first[i]) * kHugeTile + threadIdx.x;
        Unit a[kPer], b[kPer];
#pragma unroll
        for (int m = 0; m < kPer; ++m) {
            const u32 rel = min(rel0 + (u32)m * 256u, cU - 1u);     // (lanes past the end redo the last unit: loads unconditional)
            const u32 term = csgn_fastdiv(rel, dU), k = rel - term * U;
            const u32 ii = csgn_fastdiv(term, d2), jj = term - ii * (u32)t2;
            a[m] = L[(l0 + ii) * U + k];
            b[m] = R[(r0 + jj) * U + k];
        }
#pragma unroll
        for (int m = 0; m < kPer; ++m)
            asm volatile("" : "+v"(a[m]), "+v"(b[m]));               // (all loads issued before the first store)
#pragma unroll
        for (int m = 0; m < kPer; ++m) {
            const u32 rel = rel0 + (u32)m * 256u;
            if (rel < cU)
                unit_store<Unit, true>(out + o0 * U + rel, a[m] & b[m]);
        }
    }
}

template <typename Unit, int C, int M>
__global__ void __launch_bounds__(256) k_mul_ragged_flat(const Unit *__restrict__ L,
                                                         const u64 *__restrict__ offL,
                                                         const Unit *__restrict__ R,
                                                         const u64 *__restrict__ offR,
                                                         Unit *__restrict__ out,
                                                         const u64 *__restrict__ offOut, u32 batch,
                                                         u64 unit_base, u64 total_units, u32 U, FastDiv dU,
                                                         u32 pf_pairs, const u64 *__restrict__ d_gate,
                                                         u32 skip_t1, u32 skip_t2, u32 xcd_group = 0,
                                                         const u64 *__restrict__ d_huge = nullptr, u32 huge_blocks = 0)
{
    static_assert(C % M == 0, "chunks per workgroup must be a multiple of the chunks per turn");
    // csgn_mul_ragged_async: the grid was sized for the caller's bound; the real end of the output is
    // what the plan kernels left in d_gate[0] (0 when the 1x1 stream kernel takes the batch, or when
    // the products do not fit)
    if (d_gate)
        total_units = min(total_units, d_gate[0] * U);
    __shared__ u64 w_out[kWin + 1], w_l[kWin + 1], w_r[kWin + 2];
    __shared__ u32 s_next;
    // (csgn_mul_ragged_async) the launch's first huge_blocks workgroups multiply the recorded huge pairs, the rest is the CSR
    // kernel proper
    if (huge_blocks != 0u && blockIdx.x < huge_blocks) {
        mul_huge_tiles<Unit>(L, R, out, d_huge, d_gate, U, dU, blockIdx.x, huge_blocks);
        return;
    }
    const u32 bid = xcd_grouped_block(blockIdx.x - huge_blocks, gridDim.x - huge_blocks, xcd_group);
    const u64 g_begin = unit_base + (u64)bid * (256u * C);
    if (g_begin >= total_units)
        return;
    // (csgn_mul_ragged_async) a workgroup whose whole span lies inside a recorded huge pair has nothing to do -- and learns
    // so from the records, before the search for its first pair
    if (d_huge) {
        const u32 nrec = (u32)min(d_huge[0], kHugeRecords);
        const u64 span_end = min(g_begin + 256u * C, total_units);
        for (u32 i = 0; i < nrec; ++i) {
            const u64 *rec = d_huge + 1 + i * 6;
            const u64 t1 = rec[3], t2 = rec[4], o0 = rec[5];
            if (o0 * U <= g_begin && span_end <= (o0 + t1 * t2) * U && huge_record_valid(t1, t2, o0, U, d_gate[0]))
                return;
        }
    }
    const u64 term0 = g_begin / U;                              // workgroup-uniform
    const u32 r0blk = (u32)(g_begin - term0 * U);
    u32 pw = wave_find(offOut, 0u, batch, term0);               // the same answer in every wave
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
    for (int c0 = 0; c0 < C; c0 += M) {
        if (g_begin + (u32)c0 * 256u >= total_units)
            break;
        // the bet: pair pw and the one after it (uniform addresses: scalar loads, one round trip)
        const u32 pw2 = min(pw + 2u, batch);
        const u64 s_o0 = offOut[pw], s_o1 = offOut[pw + 1], s_o2 = offOut[pw2];
        const u64 s_l0 = offL[pw], s_l1 = offL[pw + 1], s_r0 = offR[pw], s_r1 = offR[pw + 1], s_r2 = offR[pw2];
        // last term of the turn (workgroup-uniform): does the bet pair own all of it, or the two together?
        const u64 turn_end = min(g_begin + (u64)(c0 + M) * 256u, total_units);          // one past the last unit
        const u64 last_term = term0 + csgn_fastdiv(r0blk + (u32)(turn_end - g_begin) - 1u, dU);
        const bool whole = last_term < s_o1;
        const bool two = !whole && last_term < s_o2;            // a turn that crosses ONE pair boundary: no window
        // (csgn_mul_ragged_async) a turn inside a huge pair that mul_huge_tiles has written: nothing to do here
        if (d_huge && whole && huge_pair_recorded(d_huge, pw, s_l1 - s_l0, s_r1 - s_r0, s_o0, U, d_gate[0]))
            continue;
        // pairs of the size classes (t1 <= skip_t1, t2 <= skip_t2) were written by their own launches
        // (mul_ragged, "size classes"): a turn inside such pairs has nothing to do here
        if (skip_t1 && (whole || two)) {
            const bool b0 = (u32)(s_l1 - s_l0) <= skip_t1 && (u32)(s_r1 - s_r0) <= skip_t2;
            const bool b1 = (u32)(offL[pw2] - s_l1) <= skip_t1 && (u32)(s_r2 - s_r1) <= skip_t2;
            if (b0 && (whole || b1)) {
                if (two)
                    pw += 1u;
                continue;
            }
        }
        if (!whole && !two) {
            const u32 i = threadIdx.x;
            const u32 pi = min(pw + i, batch);                   // the offset arrays have batch + 1 entries
            w_out[i] = offOut[pi];
            w_l[i] = offL[pi];
            w_r[i] = offR[pi];
            if (i == 0) {
                const u32 pe = min(pw + kWin, batch), pe1 = min(pw + kWin + 1u, batch);
                w_out[kWin] = offOut[pe];
                w_l[kWin] = offL[pe];
                w_r[kWin] = offR[pe];
                w_r[kWin + 1] = offR[pe1];
            }
            __syncthreads();
        }
        u32 p[M];
        u64 la[M], ra[M];                                       // operand unit indices
        bool live[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int c = c0 + m;
            const u64 g = g_begin + (u32)c * 256u + threadIdx.x;
            live[m] = g < total_units;
            // this lane's term: a 32-bit division of its distance from the workgroup's first term
            const u32 r = r0blk + (u32)c * 256u + threadIdx.x;
            const u32 dt = csgn_fastdiv(r, dU);
            const u64 term = term0 + dt;
            const u32 k = r - dt * U;
            p[m] = pw;
            la[m] = ra[m] = 0;
            if (live[m]) {
                u64 o0 = s_o0, l0 = s_l0, rr0 = s_r0;
                u32 t2 = (u32)(s_r1 - s_r0), t1 = (u32)(s_l1 - s_l0);
                if (two && term >= s_o1) {                      // the second pair of the bet
                    p[m] = pw + 1u;
                    o0 = s_o1;
                    l0 = s_l1;
                    rr0 = s_r1;
                    t2 = (u32)(s_r2 - s_r1);
                    t1 = skip_t1 ? (u32)(offL[pw2] - s_l1) : 0u;
                } else if (!whole && term >= s_o1) {            // past the end of pair pw: look in the window
                    // largest j in [0, kWin] with w_out[j] <= term (w_out[0] = s_o0 <= term)
                    u32 lo = 0, hi = kWin + 1u;
#pragma unroll
                    for (int step = 0; step < 9; ++step) {      // 257 entries
                        const u32 mid = (lo + hi) >> 1;
                        const bool le = w_out[mid] <= term;
                        lo = le ? mid : lo;
                        hi = le ? hi : mid;
                    }
                    if (lo == kWin && pw + kWin < batch) {      // beyond the window (long runs of empty pairs): walk
                        p[m] = csr_gallop(offOut, pw + kWin, batch, term);
                        o0 = offOut[p[m]];
                        l0 = offL[p[m]];
                        rr0 = offR[p[m]];
                        t2 = (u32)(offR[p[m] + 1] - rr0);
                        t1 = (u32)(offL[p[m] + 1] - l0);
                    } else {
                        p[m] = pw + lo;
                        o0 = w_out[lo];
                        l0 = w_l[lo];
                        rr0 = w_r[lo];
                        t2 = (u32)(w_r[lo + 1] - rr0);
                        t1 = (u32)(w_l[lo + 1] - l0);           // lo < kWin here: w_l[kWin] is staged
                    }
                }
                const u32 q = (u32)(term - o0);                 // product term index inside the pair
                const u32 i = q / t2, j = q - i * t2;
                la[m] = (l0 + i) * U + k;
                ra[m] = (rr0 + j) * U + k;
                if (skip_t1 && t1 <= skip_t1 && t2 <= skip_t2) {   // a size-class launch wrote this pair
                    live[m] = false;
                    la[m] = ra[m] = 0;
                }
            }
        }
        Unit lv[M], rv[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {                           // unconditional (index 0 for dead lanes): in flight together
            lv[m] = L[la[m]];
            rv[m] = R[ra[m]];
        }
        // (... and issued HERE, all 2M of them: left to itself the compiler sinks the loads of one unit into the `if (live)`
        // of its store, behind the other stores, with a vmcnt(0) of their own)
#pragma unroll
        for (int m = 0; m < M; ++m)
            asm volatile("" : "+v"(lv[m]), "+v"(rv[m]));
#pragma unroll
        for (int m = 0; m < M; ++m)
            if (live[m])
                unit_store<Unit, true>(out + g_begin + (u32)(c0 + m) * 256u + threadIdx.x, lv[m] & rv[m]);
        if (two) {
            pw += 1u;                                           // last_term >= s_o1: the turn ended in the second pair
        } else if (!whole) {                                    // the next turn starts from the last lane's pair
            if (threadIdx.x == 255u)
                s_next = p[M - 1];
            __syncthreads();
            pw = s_next;
        }
    }
}

// Size classes of small pairs (round 4): t1 in (0, 8], (8, 16], (16, 32], (32, 64] x t2 in (0, 2], (2, 4], ... (64, 128]:
// 28 classes; the plan counts each class's pairs and product terms and lists its pairs, so that the multiply can
// give every class ONE launch of the LDS-tiled kernel shaped for it (pair = list[block / tiles]: no lookup of any
// kind) and leave only pairs with a long side to the CSR kernel.
constexpr u32 kClsRows = 4, kClsCols = 7, kNumClasses = kClsRows * kClsCols;
constexpr u32 kClsMaxT1 = 64, kClsMaxT2 = 128;
__host__ __device__ inline int class_of(u64 t1, u64 t2)
{
    if (t1 == 0 || t2 == 0 || t1 > kClsMaxT1 || t2 > kClsMaxT2)
        return -1;
    const int r = t1 <= 8 ? 0 : t1 <= 16 ? 1 : t1 <= 32 ? 2 : 3;
    int c = 0;
    while ((2ull << c) < t2)
        ++c;                                         // t2 <= 2 << c
    return r * (int)kClsCols + c;
}
inline u32 class_max_t1(int cls) { return 8u << (cls / (int)kClsCols); }
inline u32 class_max_t2(int cls) { return 2u << (cls % (int)kClsCols); }

// the checksum of the offset arrays is accumulated in kSumSlots words (workgroup b adds to slot b % kSumSlots: one
// word would take every workgroup's atomic in turn, ~11 ns each) and summed on the host
constexpr u32 kSumSlots = 64;
constexpr u64 kPlanHeadWords = 4 + 1 + kHugeRecords * 6 + 2 + kSumSlots + 2 * kNumClasses;  // [plan4][huge count][records][left terms, right terms of the batch][checksum slots][class pairs][class terms]
constexpr u64 kSumAt = 4 + 1 + kHugeRecords * 6 + 2;       // first checksum slot of the head
constexpr u64 kClassAt = kSumAt + kSumSlots;                // first class word of the head

// ---- ragged multiply, wave-cooperative form (round 4) -----------------------------------------------------------
// The CSR kernel above gives every LANE a 16-byte unit of the flattened output and lets it find out on its own which
// pair, row and column that is: a division by the unit count, a search of the offset window, a division by t2 and
// 64-bit address arithmetic -- ~100 vector and ~70 scalar instructions per 64 units (rocprofv3 --pmc SQ_INSTS_VALU
// on a log-normal batch of mean-8 pairs: 235 M wave instructions for 2.35 M blocks), which is what bounds it on small
// pairs (3.7 TB/s where the same bytes stream at 6).  Here a WAVE owns a contiguous stretch of the output and walks
// its pairs together: the offsets of the next 63 pairs sit in the lanes of three registers (one coalesced load each,
// read back with v_readlane), every pair's stretch is written in blocks of 64 units, and a lane keeps (row, column,
// unit in term) of its unit by ADDING per block -- one division per lane and pair, none per block; 32-bit offsets from
// wave-uniform pair bases.  The price: the last block of a pair's stretch is partly idle (a 1 x 1 pair of 160-byte
// terms uses 10 lanes of 64), ~5 % of the blocks on the batches this is for.
__device__ inline u64 readlane64(u64 v, u32 j)
{
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, (int)j);
    const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), (int)j);
    return ((u64)hi << 32) | lo;
}

// The walk is ONE loop over blocks, kCoopBlocks of them in flight, whatever pairs they belong to: a block that finds its
// pair used up moves the wave on to the next one (a wave-uniform branch), so a run of 1 x 1 pairs keeps as many loads
// and stores in flight as the middle of a large pair does.  (The first form finished a pair before it started the
// next: 0.5 TB/s on 65 535 singles.)
// A wave's work is blocks, not bytes -- a single-term pair costs it a block as 64 units of a large pair do -- so the
// output is dealt out along a VIRTUAL axis on which pair p starts at f(p) = offOut[p] * U + vw * p: its units plus vw
// units of padding per pair before it (vw = 64 for a launch over the whole output; 0, the plain unit axis, for the
// slices of a large output, whose ends the host knows in units only).  Stretches of equal virtual length hold at most
// twice the blocks of one another, and nothing but offOut is needed to find them.
// Two things about waiting (round 4, second form; rocprofv3 --pmc on the first: waves parked 65 % of their cycles):
//   * the offset window lives in registers across the loop; left alone, the compiler cannot tell that its three loads
//     have landed and puts s_waitcnt vmcnt(0) in front of every v_readlane of them -- a full drain of the wave's
//     loads and stores at EVERY pair.  The window is waited for once, where it is loaded (once per 63 pairs).
//   * gfx9 counts loads AND stores in one in-order counter (vmcnt): a wave that waits for loads it issued after a
//     batch of stores also waits for those stores to be acknowledged by L2, microseconds under write load.  The walk is
//     therefore software-pipelined over two groups of K blocks -- the loads of group B are issued BEFORE group A is
//     stored, so the wait in front of A's stores can count PAST B's loads and the stores before them: a wave never
//     waits for a store.  The compiler's own counting cannot express this (a store it may skip when no lane is live
//     makes it count the shorter path, and it waits for the stores after all; its register reuse put full drains in
//     front of the loads besides), so in the PIPE form the operand loads and the waits are inline assembly and the
//     counts are ours:
//         fill(X) issues exactly 2K loads; put(X) of a group filled before the end of the stretch issues exactly K
//         stores (every block generated before `done` has its first lane live, so no store is skipped);
//         in front of put(A):  ... L(A) | S(B') L(B)  -> vmcnt(3K) leaves L(A) landed
//     What the compiler adds for the loads it does know (the window, the start-up search) only waits for MORE.
//     The one thing that must not happen is a copy of a loaded register between its load and its wait (the compiler
//     thinks the value is there): tests/test_capi_symbols.py::test_coop_kernel_isa_keeps_loaded_registers_untouched
//     disassembles the kernel and checks.
// Everything the walk branches on is wave-uniform, and the wave index is read through readfirstlane so that the
// compiler knows it too (scalar branches and SGPR bases instead of exec masks and VGPR copies).
template <typename Unit, int K>
struct CoopGroup {
    Unit lv[K], rv[K];
    char *obase[K];
    u32 oa[K];
    bool live[K];
};

// global_load of one unit outside the compiler's books.  The address is a VGPR pair: an SGPR base would have to be
// five wait states old if a VALU instruction wrote it (v_readlane, a reload of a spilled SGPR), and the hazard
// recogniser does not look inside inline assembly.  The stores stay the compiler's (it never waits for a store, and it
// knows the wait state a 16-byte store's data registers need).
__device__ inline void coop_load_asm(unit16 &d, const char *addr)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(addr));
}
__device__ inline void coop_load_asm(unit8 &d, const char *addr)
{
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(addr));
}
template <int N>
__device__ inline void coop_wait()
{
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// largest p in [0, batch) with f(p) <= target (f(0) = 0 <= target), by a whole wave: wave_find on the virtual axis
__device__ inline u32 wave_find_virtual(const u64 *__restrict__ offOut, u32 batch, u32 U, u32 vw, u64 target)
{
    const u32 lane = threadIdx.x & (kWave - 1);
    u32 lo = 0, hi = batch;
    while (hi - lo > 1) {
        const u32 n = hi - lo, s = (n + kWave - 1) / kWave;
        const u64 idx = (u64)lo + (u64)lane * s;
        const bool ok = idx < hi && offOut[idx] * U + (u64)vw * idx <= target;
        const u32 c = (u32)__popcll(__ballot(ok));             // >= 1: lane 0 holds by the invariant
        lo += (c - 1u) * s;
        hi = min(lo + s, hi);
    }
    return (u32)__builtin_amdgcn_readfirstlane((int)lo);
}

#ifdef CSGN_COOP_STAMPS     // dev (tools/coop_probe.hip): every wave leaves {cycles: whole, search, waits; blocks, pairs, windows} behind
__device__ u64 *g_coop_stamps;
#define CSGN_CSTAMP(...) __VA_ARGS__
#else
#define CSGN_CSTAMP(...)
#endif

// The operand touches' destination: v127.  The kernel is compiled for 127 vector registers (amdgpu_num_vgpr: v0 .. v126), so
// the register allocator never hands v127 out -- a destination it took for dead would be given to another value while
// touches that write it are still in flight, and a value carried in a variable gets copied between registers (the first
// build did exactly that; tools/check_coop_isa.py found it) -- and every touch names it as clobbered, so the wave's
// allocation covers it (128 registers: four waves a SIMD, what the pipelined form has anyway).
#define CSGN_COOP_TOUCH(ptr) asm volatile("global_load_dword v127, %0, off" : : "v"(ptr) : "v127")
template <typename Unit, int K, bool PIPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(127))) k_mul_ragged_coop(const Unit *__restrict__ L, const u64 *__restrict__ offL,
                                                         const Unit *__restrict__ R, const u64 *__restrict__ offR,
                                                         Unit *__restrict__ out, const u64 *__restrict__ offOut,
                                                         u32 batch, u64 v_begin, u64 v_end, u32 U, FastDiv dU,
                                                         u32 span, u32 vw, const u64 *__restrict__ d_gate, u32 touch_terms,
                                                         u32 xcd_group = 0)
{
    // csgn_mul_ragged_async: the grid was sized for the caller's bound, the real end is in the gate (0: nothing is to be
    // written -- the products do not fit the bound, or the 1x1 stream kernel has the batch)
    if (d_gate) {
        const u64 real_terms = d_gate[0];
        v_end = real_terms ? min(v_end, real_terms * U + (u64)vw * batch) : 0ull;
    }
    const u32 lane = threadIdx.x & (kWave - 1);
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 bid = xcd_grouped_block(blockIdx.x, gridDim.x, xcd_group);
    const u64 v0 = v_begin + ((u64)bid * 4u + wv) * span;       // this wave's stretch of the virtual axis
    if (v0 >= v_end)
        return;
    const u64 v1 = min(v0 + span, v_end);
    CSGN_CSTAMP(const u64 st_begin = __builtin_readcyclecounter(); u64 st_wait = 0; u32 st_blocks = 0, st_pairs = 0, st_windows = 0;)
    u32 p = wave_find_virtual(offOut, batch, U, vw, v0);        // the pair v0 falls in (its units or its padding)
    CSGN_CSTAMP(const u64 st_found = __builtin_readcyclecounter();)
    // From here on every place is a BYTE offset from a wave-uniform base, 32 bits wide (a pair's product stays under
    // 4 GiB: the launcher's condition) -- an SGPR base plus one VGPR per address instead of a 64-bit VGPR pair each.
    constexpr u32 kUB = (u32)sizeof(Unit), kBlockBytes = kWave * kUB;
    const u32 termB = U * kUB;                                   // bytes of a term
    const u32 kstepB = (kWave - csgn_fastdiv(kWave, dU) * U) * kUB;   // a block on, the place inside the term moves by 64 mod U units
    // the window: offsets of pairs p .. p+62 (and the entry that closes the last of them) in the lanes
    u64 wo = 0, wl = 0, wr = 0, todo = 0;
    bool fresh = true;                                          // the window has to be loaded, p stays
    // the pair in hand (wave-uniform) ...
    const char *Lp = reinterpret_cast<const char *>(L), *Rp = reinterpret_cast<const char *>(R);
    char *Op = reinterpret_cast<char *>(out);
    u32 xeB = 0, xbB = 0, rowB = kUB, aB = 0, bB = kBlockBytes;   // xbB >= xeB: used up
    // ... and this lane's place in it: its unit x, the place y inside the row (= inside the right operand), the start
    // lu of its left term, the place k inside the term
    u32 xB = 0, yB = 0, luB = 0, kB = 0;
    bool done = false;
    // The operand touch (touch_terms != 0).  Operands of a batch of small pairs are a quarter of its products and cold.  A
    // line's first load is an HBM round trip, and every other load of that line issued meanwhile -- the next row's block
    // re-reading the right operand, the next block of the row re-reading the left term -- waits for it inside the L1,
    // whose pipe is in order: the stores of sixteen waves queue behind (profiles/r04/NOTES_ragged_kernels.md: the L1 stalled on its pending queue
    // 69 % of the time; the same kernel on operands that sit in the memory-side cache needs half the cycles per block).
    // So a wave that has just loaded an offset window reads ONE dword of every 128-byte line of the operands of the
    // window's pairs that begin inside its stretch -- 64 lines per instruction, every line once, nothing re-read while it
    // is on its way -- and only then walks them.  Loads return in order, so the first wait after a window waits for the
    // touch as well: one HBM round trip per 63 pairs instead of one per cold line.  Left terms of the first pair: from the
    // row the stretch starts in; of the last pair: to the row it ends in; the first pair's right operand only if the
    // stretch starts in its first row (else the waves before this one have it in the L2); at most touch_terms terms a side.
    // The loads are inline assembly and all write v127, which nothing else uses (CSGN_COOP_TOUCH, above the kernel).
    auto touch_window = [&]() {
        const u64 fl = wo * U + (u64)vw * (p + lane);
        const u64 f0 = readlane64(fl, 0);
        if (f0 >= v1)
            return;                                             // the walk ends at this pair
        const u64 inside = __ballot(lane == 0u || (lane < kWave - 1u && p + lane < batch && fl < v1));
        const u32 last = 63u - (u32)__builtin_clzll(inside);     // pairs 0 .. last of the window begin in front of v1
        const u64 o0 = readlane64(wo, 0), o1 = readlane64(wo, 1), l0 = readlane64(wl, 0), r0 = readlane64(wr, 0), r1 = readlane64(wr, 1);
        u64 lo = l0, rlo = r0;
        const u32 rowlen0 = (u32)(r1 - r0) * U;
        if (v0 > f0 && rowlen0) {
            const u32 row0 = (u32)__builtin_amdgcn_readfirstlane((int)((u32)min(v0 - f0, (o1 - o0) * U) / rowlen0));
            lo += row0;
            if (row0)
                rlo = r1;
        }
        const u64 lL = readlane64(wl, last), rL = readlane64(wr, last), rL1 = readlane64(wr, last + 1u);
        u64 hi = readlane64(wl, last + 1u);
        const u32 rowlenL = (u32)(rL1 - rL) * U;
        if (rowlenL) {
            const u64 fL = readlane64(fl, last);
            const u32 rows = (u32)__builtin_amdgcn_readfirstlane((int)((u32)min(v1 - fL, (u64)0xffffffffu) / rowlenL));
            hi = min(hi, lL + rows + 1u);
        }
        hi = min(hi, lo + touch_terms);
        const u64 rhi = min(rL1, rlo + touch_terms);
        const char *Lb = reinterpret_cast<const char *>(L), *Rb = reinterpret_cast<const char *>(R);
        for (size_t a = (size_t)lo * termB + lane * 128u, e = (size_t)hi * termB; a < e; a += kWave * 128u)
            CSGN_COOP_TOUCH(Lb + a);
        for (size_t a = (size_t)rlo * termB + lane * 128u, e = (size_t)rhi * termB; a < e; a += kWave * 128u)
            CSGN_COOP_TOUCH(Rb + a);
    };
    // the places of the next K blocks, their operand loads issued
    auto fill = [&](CoopGroup<Unit, K> &g) {
        u32 la[K], ra[K];
        const char *lbase[K], *rbase[K];
#pragma unroll
        for (int m = 0; m < K; ++m) {
            while (xbB >= xeB && !done) {                       // (wave-uniform) the next pair with product terms
                if (todo == 0ull) {
                    if (!fresh)
                        p += kWave - 1u;
                    if (p >= batch) {
                        done = true;
                        break;
                    }
                    const u32 pi = min(p + lane, batch);
                    wo = offOut[pi];
                    wl = offL[pi];
                    wr = offR[pi];
                    asm volatile("" : "+v"(wo), "+v"(wl), "+v"(wr));   // all three landed HERE (see above)
                    CSGN_CSTAMP(++st_windows;)
                    const u64 wo_next = (u64)__shfl_down(wo, 1, kWave);
                    todo = __ballot(lane < kWave - 1u && p + lane < batch && wo_next > wo);
                    fresh = false;
                    if (touch_terms)
                        touch_window();
                    continue;
                }
                const u32 j = (u32)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                const u64 o0 = readlane64(wo, j), o1 = readlane64(wo, j + 1u);
                const u64 fv = o0 * U + (u64)vw * (p + j);      // where the pair starts on the virtual axis
                if (fv >= v1) {
                    done = true;
                    break;
                }
                const u64 pair_units = (o1 - o0) * U;
                const u64 xs = v0 > fv ? min(v0 - fv, pair_units) : 0ull;
                const u64 end = min(pair_units, v1 - fv);
                if (xs >= end)
                    continue;                                   // only the pair's padding is in this stretch
                const u64 l0 = readlane64(wl, j), r0 = readlane64(wr, j), r1 = readlane64(wr, j + 1u);
                Lp = reinterpret_cast<const char *>(L + l0 * U);
                Rp = reinterpret_cast<const char *>(R + r0 * U);
                Op = reinterpret_cast<char *>(out + o0 * U);
                const u32 rowlen = (u32)(r1 - r0) * U;
                const u32 x = (u32)xs + lane;
                u32 row = 0, y = x;
                if ((u32)end > rowlen) {                        // more than the first row: one division per lane
                    row = x / rowlen;
                    y = x - row * rowlen;
                }
                xbB = (u32)xs * kUB;
                xeB = (u32)end * kUB;
                rowB = rowlen * kUB;
                xB = x * kUB;
                yB = y * kUB;
                luB = row * termB;
                kB = (x - csgn_fastdiv(x, dU) * U) * kUB;       // rowlen is a multiple of U
                const u32 a = rowlen <= kWave ? kWave / rowlen : 0u;     // one block on: a rows and bB bytes
                bB = kBlockBytes - a * rowB;
                aB = a * termB;
                CSGN_CSTAMP(++st_pairs;)
            }
            CSGN_CSTAMP(st_blocks += done ? 0u : 1u;)
            g.live[m] = !done && xB < xeB;
            lbase[m] = Lp;
            rbase[m] = Rp;
            g.obase[m] = Op;
            la[m] = g.live[m] ? luB + kB : 0u;
            ra[m] = g.live[m] ? yB : 0u;
            g.oa[m] = xB;
            xbB += kBlockBytes;
            xB += kBlockBytes;
            yB += bB;
            luB += aB;
            if (yB >= rowB) {
                yB -= rowB;
                luB += termB;
            }
            kB += kstepB;
            kB = kB >= termB ? kB - termB : kB;
        }
#pragma unroll
        for (int m = 0; m < K; ++m) {                           // unconditional (a pair's first unit for idle lanes)
#if defined(CSGN_COOP_STAMPS) && defined(COOP_PROBE_NO_LOAD)       // tools/coop_probe.hip: the walk and the stores alone
            g.lv[m] = g.rv[m] = Unit{} + (Unit)(la[m] + ra[m]);
            continue;
#endif
            if (PIPE) {
                coop_load_asm(g.lv[m], lbase[m] + la[m]);
                coop_load_asm(g.rv[m], rbase[m] + ra[m]);
            } else {
                g.lv[m] = *reinterpret_cast<const Unit *>(lbase[m] + la[m]);
                g.rv[m] = *reinterpret_cast<const Unit *>(rbase[m] + ra[m]);
            }
        }
    };
    auto put = [&](CoopGroup<Unit, K> &g) {
        // (not PIPE) every load of the group is waited for HERE, on all paths: a wait left inside `if (live)` leaves
        // the loads of an idle block pending in the compiler's books, and the next fill's first write to those
        // registers then costs a vmcnt(0).  (PIPE) the values pass through here on their way from the wait to the
        // stores, so nothing that uses them can be scheduled in front of it.
#pragma unroll
        for (int m = 0; m < K; ++m)
            asm volatile("" : "+v"(g.lv[m]), "+v"(g.rv[m]));
#if defined(CSGN_COOP_STAMPS) && defined(COOP_PROBE_NO_STORE)      // tools/coop_probe.hip: the walk and the loads alone
#pragma unroll
        for (int m = 0; m < K; ++m) {
            Unit v = g.lv[m] & g.rv[m];
            asm volatile("" : : "v"(v));
        }
        return;
#endif
#pragma unroll
        for (int m = 0; m < K; ++m)
            if (g.live[m]) {
#if defined(CSGN_COOP_STAMPS) && defined(COOP_PROBE_PLAIN_STORE)   // tools/coop_probe.hip: ordinary stores instead of non-temporal ones
                unit_store<Unit, false>(reinterpret_cast<Unit *>(g.obase[m] + g.oa[m]), g.lv[m] & g.rv[m]);
#else
                unit_store<Unit, true>(reinterpret_cast<Unit *>(g.obase[m] + g.oa[m]), g.lv[m] & g.rv[m]);
#endif
            }
    };
    CoopGroup<Unit, K> A, B;
#ifdef CSGN_COOP_STAMPS
#define CSGN_COOP_WAIT(N) do { const u64 w0 = __builtin_readcyclecounter(); coop_wait<N>(); asm volatile("" ::: "memory"); st_wait += __builtin_readcyclecounter() - w0; } while (0)
#define CSGN_COOP_LEAVE() do { if (lane == 0) { u64 *st = g_coop_stamps + ((u64)bid * 4u + wv) * 8u; st[0] = __builtin_readcyclecounter() - st_begin; st[1] = st_found - st_begin; st[2] = st_wait; st[3] = st_blocks; st[4] = st_pairs; st[5] = st_windows; st[6] = st_begin; } } while (0)
#else
#define CSGN_COOP_WAIT(N) coop_wait<N>()
#define CSGN_COOP_LEAVE() do {} while (0)
#endif
    if (!PIPE) {                                                // loads, wait, stores; the compiler counts
        do {
            fill(A);
            put(A);
        } while (!done);
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");          // (the touches have landed before the wave ends)
        CSGN_COOP_LEAVE();
        return;
    }
    // Every wait in the loop is the same vmcnt(3K), on every path (no flag to follow in the generated code:
    // tools/check_coop_isa.py walks it).  The first turn has no group A yet: put(A) stores nothing (no lane live), and K
    // one-dword loads of nothing stand in for its K stores so that the count in front of put(B) holds.  Their
    // destinations are kept alive (an empty asm per turn) until the wait at the top of the second turn has retired them:
    // registers the compiler took for dead would be handed out while the loads are still in flight.
#pragma unroll
    for (int m = 0; m < K; ++m)
        A.live[m] = false;
    u32 stand_in[K];
#pragma unroll
    for (int m = 0; m < K; ++m)
        stand_in[m] = 0;
    bool first = true;
    for (;;) {
        fill(B);                                                // 2K loads
        CSGN_COOP_WAIT(3 * K);                                  // ... L(A) | S(B') L(B): L(A) has landed
#pragma unroll
        for (int m = 0; m < K; ++m)
            asm volatile("" : "+v"(stand_in[m]));
        put(A);                                                 // K stores
        if (first) {
#pragma unroll
            for (int m = 0; m < K; ++m)
                asm volatile("global_load_dword %0, %1, off" : "=v"(stand_in[m]) : "v"(L));
            first = false;
        }
        if (done) {
            CSGN_COOP_WAIT(0);
            put(B);
            break;
        }
        fill(A);
        CSGN_COOP_WAIT(3 * K);                                  // L(B) | S(A) L(A'): L(B) has landed
        put(B);
        if (done) {
            CSGN_COOP_WAIT(0);
            put(A);
            break;
        }
    }
    CSGN_COOP_LEAVE();
#undef CSGN_COOP_WAIT
#undef CSGN_COOP_LEAVE
}

// one pair's share of the checksum of the offset arrays: one splitmix64 of the three entries folded together and
// salted with the position (round 4; three splitmix64 per pair until then -- 64-bit multiplies run at a quarter of
// the rate, and hashing four pairs a thread that way took the plan kernel 3.3 us of its 19)
__device__ inline u64 offsets_mix(u64 b, u64 l, u64 r, u64 o)
{
    return csgn_splitmix64((l ^ ((r << 21) | (r >> 43)) ^ ((o << 42) | (o >> 22))) + CSGN_GOLDEN * (b + 1));
}

// Product term offsets = exclusive scan of t1_b*t2_b over the batch, plus what the launcher wants to know: shape
// maxima, the HUGE pairs (products of kHugeTerms terms and more: {pair, offL, offR, t1, t2, offOut}, up to
// kHugeRecords of them -- csgn_mul_planned gives such a pair a uniform launch of its own), operand totals, a
// checksum of the three offset arrays, the size-class histogram, and for csgn_mul_ragged_async the gate.
constexpr u32 kPlanThreads = 1024, kPlanChunk = kPlanThreads * 4;       // pairs per workgroup of k_plan

#ifdef CSGN_PLAN_STAMPS     // dev (tools/prof_plan_phases.py): every 4th workgroup leaves 100 MHz wall-clock stamps behind the scan block
#define CSGN_PSTAMP(k) do { if (tid == 0 && (chunk & 3u) == 0u) scan[nchunks + 3 + (chunk >> 2) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define CSGN_PSTAMP(k) do {} while (0)
#endif

// ONE kernel (round 4; three until then: chunk scans, a scan of the chunk totals, a fix-up pass that read every
// offset a second time -- 27-45 us at a million pairs): a workgroup takes a ticket, scans its 4096 pairs,
// gets the sum of all chunks before it by a decoupled look-back over one {flag, sum} granule per chunk (csgn_device.h),
// and writes final offsets at once.  `scan` = [nchunks status granules][ticket],
// zeroed by the caller.  One workgroup per CU (a 64-VGPR build for two spilled 270 bytes), so a million pairs are
// exactly ONE wave of 256 workgroups: the closing entry of the arrays is written by the thread that owns the last
// pair and gets no workgroup of its own (as a 257th it ran after the others, 9 us more).
// `vec`: all three arrays 16-byte aligned -> 16-byte loads and stores.  `resident`: the CUs of the device = the
// workgroups of this kernel it holds at once.
__global__ void __launch_bounds__(kPlanThreads) k_plan(u64 batch, u32 nchunks, const u64 *__restrict__ offL,
                                                          const u64 *__restrict__ offR, u64 *__restrict__ offOut,
                                                          u64 *__restrict__ head, u64 *__restrict__ scan, u32 classes,
                                                          u64 *__restrict__ gate, u64 capacity_terms, u32 can_stream, u32 vec,
                                                          u32 resident)
{
    constexpr u32 kWaves = kPlanThreads / kWave;
    __shared__ u64 wtot[kWaves], wmax[3][kWaves], s_prefix;
    __shared__ u32 s_chunk;
    __shared__ u32 h_pairs[kNumClasses];
    __shared__ u64 h_terms[kNumClasses];
    u64 *plan4 = head, *huge = head + 4;
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;
#ifdef CSGN_PLAN_STAMPS
    const u64 st0 = wall_clock64();
#endif
    if (tid < kNumClasses) {
        h_pairs[tid] = 0u;
        h_terms[tid] = 0ull;
    }
    // A chunk waits for the chunks before it, so those must be running: in general a workgroup takes its chunk from a
    // ticket counter (whoever holds an older ticket started earlier).  A grid that fits the chip at once needs no
    // ticket -- every workgroup gets a CU without any other having to finish -- and 256 atomics on one word are 3-4 us
    // for the last in line.
    u32 chunk = blockIdx.x;
    if (nchunks > resident) {
        if (tid == 0)
            s_chunk = atomicAdd(reinterpret_cast<u32 *>(scan + nchunks), 1u);
        __syncthreads();
        chunk = s_chunk;
    }
#ifdef CSGN_PLAN_STAMPS
    if (tid == 0 && (chunk & 3u) == 0u)
        scan[nchunks + 3 + (chunk >> 2) * 8] = st0;
#endif
    CSGN_PSTAMP(1);
    const u64 b0 = (u64)chunk * kPlanChunk + (u64)tid * 4u;
    // entries b0 .. b0+4 of both operand arrays (past the closing entry: the closing entry again, so such pairs are 0 x 0)
    u64 l[5], r[5];
    const bool whole = b0 + 4 <= batch;
    if (vec && whole) {
        const ulonglong2 la = *reinterpret_cast<const ulonglong2 *>(offL + b0), lb = *reinterpret_cast<const ulonglong2 *>(offL + b0 + 2);
        const ulonglong2 ra = *reinterpret_cast<const ulonglong2 *>(offR + b0), rb = *reinterpret_cast<const ulonglong2 *>(offR + b0 + 2);
        l[0] = la.x; l[1] = la.y; l[2] = lb.x; l[3] = lb.y; l[4] = offL[b0 + 4];
        r[0] = ra.x; r[1] = ra.y; r[2] = rb.x; r[3] = rb.y; r[4] = offR[b0 + 4];
    } else {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const u64 at = b0 + j < batch ? b0 + j : batch;
            l[j] = offL[at];
            r[j] = offR[at];
        }
    }
    u64 m1 = 0, m2 = 0, mp = 0, mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u64 t1 = l[j + 1] - l[j], t2 = r[j + 1] - r[j], c = t1 * t2;
        m1 = max(m1, t1);
        m2 = max(m2, t2);
        mp = max(mp, c);
        mine += c;
    }
    // exclusive scan of the per-thread sums
    u64 incl = mine;
    for (u32 d = 1; d < kWave; d <<= 1) {
        const u64 nb = (u64)__shfl_up(incl, d, kWave);
        if (lane >= d)
            incl += nb;
    }
    if (lane == kWave - 1)
        wtot[wv] = incl;
    for (int off = 32; off > 0; off >>= 1) {
        m1 = max(m1, (u64)__shfl_down(m1, off, 64));
        m2 = max(m2, (u64)__shfl_down(m2, off, 64));
        mp = max(mp, (u64)__shfl_down(mp, off, 64));
    }
    if (lane == 0) {
        wmax[0][wv] = m1;
        wmax[1][wv] = m2;
        wmax[2][wv] = mp;
    }
    __syncthreads();
    CSGN_PSTAMP(2);
    u64 wbase = 0, all = 0;
    for (u32 w = 0; w < kWaves; ++w) {
        wbase += w < wv ? wtot[w] : 0ull;
        all += wtot[w];
    }
    if (wv == 0) {
        // a chunk with a pair of more than one product term marks itself: the LAST chunk then has, from its look-back
        // alone, what csgn_mul_ragged_async's gate says to the kernels behind this one -- real output terms / does not
        // fit / every pair 1 x 1 (the plain AND stream; total == batch with no pair above one term).  (Round-4 notes:
        // a counter of finished workgroups cost 3 us of same-address atomics plus, with the release fence that makes
        // the total visible, 5-8 us of L2 write-back with every workgroup's offsets in it.)
        const bool mark = __ballot(lane < kWaves && wmax[2][lane < kWaves ? lane : 0] > 1ull) != 0ull;
        bool marked_before;
        const u64 excl = lookback_marked(scan, chunk, all, mark, marked_before);
        if (lane == 0) {
            s_prefix = excl;
            if (gate && chunk == nchunks - 1u) {
                const u64 total = excl + all;
                const bool fits = total <= capacity_terms;
                const bool ones = can_stream && fits && total == batch && !mark && !marked_before;
                gate[0] = (fits && !ones) ? total : 0ull;
                gate[1] = fits ? 0ull : 1ull;
                gate[2] = ones ? total : 0ull;
            }
        }
    } else if (tid < kWave + 3u) {
        // (wave 1) shape maxima: an atomic only where the workgroup would RAISE the running maximum -- atomics on one
        // word complete at ~11 ns apiece chip-wide, and nearly every workgroup sees that it has nothing to add
        const u32 k = tid - kWave;
        u64 m = 0;
        for (u32 w = 0; w < kWaves; ++w)
            m = max(m, wmax[k][w]);
        if (m > __hip_atomic_load(plan4 + 1 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(reinterpret_cast<unsigned long long *>(plan4 + 1 + k), (unsigned long long)m);
    }
    __syncthreads();
    CSGN_PSTAMP(3);
    u64 run = s_prefix + wbase + incl - mine, mix = 0;
    const bool wide = vec && whole;
    if (wide) {
        const u64 o1 = run + (l[1] - l[0]) * (r[1] - r[0]), o2 = o1 + (l[2] - l[1]) * (r[2] - r[1]);
        const u64 o3 = o2 + (l[3] - l[2]) * (r[3] - r[2]);
        *reinterpret_cast<ulonglong2 *>(offOut + b0) = make_ulonglong2(run, o1);
        *reinterpret_cast<ulonglong2 *>(offOut + b0 + 2) = make_ulonglong2(o2, o3);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u64 b = b0 + j;
        if (b >= batch)
            break;
        const u64 t1 = l[j + 1] - l[j], t2 = r[j + 1] - r[j], c = t1 * t2;
        if (!wide)
            offOut[b] = run;
        mix += offsets_mix(b, l[j], r[j], run);
        const int cls = classes ? class_of(t1, t2) : -1;
        if (cls >= 0) {
            atomicAdd(h_pairs + cls, 1u);
            atomicAdd(reinterpret_cast<unsigned long long *>(h_terms + cls), (unsigned long long)c);
        }
        if (c >= kHugeTerms) {
            const u64 slot = atomicAdd(reinterpret_cast<unsigned long long *>(huge), 1ull);
            if (slot < kHugeRecords) {
                u64 *rec = huge + 1 + slot * 6;
                rec[0] = b; rec[1] = l[j]; rec[2] = r[j]; rec[3] = t1; rec[4] = t2; rec[5] = run;
            }
        }
        run += c;
        if (b + 1 == batch) {                       // the owner of the last pair: the closing entries of the three arrays
            offOut[batch] = run;
            plan4[0] = run;
            mix += offsets_mix(batch, l[j + 1], r[j + 1], run);
            huge[1 + kHugeRecords * 6] = l[j + 1] - offL[0];         // operand totals: what sizes the slices of a large product
            huge[2 + kHugeRecords * 6] = r[j + 1] - offR[0];
        }
    }
    if (batch == 0 && tid == 0) {                   // an empty batch has closing entries only
        offOut[0] = 0;
        plan4[0] = 0;
        mix += offsets_mix(0, l[0], r[0], 0);
    }
    // checksum of the three offset arrays as planned (csgn_mul_plan_validate recomputes it later): every wave adds
    // its share to one of the kSumSlots words
    for (int off = 32; off > 0; off >>= 1)
        mix += (u64)__shfl_down(mix, off, 64);
    if (lane == 0)
        atomicAdd(reinterpret_cast<unsigned long long *>(head + kSumAt + ((chunk * kWaves + wv) % kSumSlots)), (unsigned long long)mix);
    CSGN_PSTAMP(4);
    if (classes) {                                  // the workgroup's class histogram: one global atomic per class it met
        __syncthreads();
        if (tid < kNumClasses && h_pairs[tid]) {
            atomicAdd(reinterpret_cast<unsigned long long *>(head + kClassAt + tid), (unsigned long long)h_pairs[tid]);
            atomicAdd(reinterpret_cast<unsigned long long *>(head + kClassAt + kNumClasses + tid), (unsigned long long)h_terms[tid]);
        }
    }
    if (tid != kWave)
        return;
#ifdef CSGN_PLAN_STAMPS
    if ((chunk & 3u) == 0u)
        scan[nchunks + 3 + (chunk >> 2) * 8 + 5] = wall_clock64();
#endif
}

// class bases (exclusive scan of the class pair counts) and zeroed cursors, then the lists
__global__ void k_class_bases(const u64 *__restrict__ head, u64 *__restrict__ bases)
{
    u64 run = 0;
    for (u32 c = 0; c < kNumClasses; ++c) {
        bases[c] = run;
        bases[kNumClasses + 1 + c] = 0;                         // cursor
        run += head[kClassAt + c];
    }
    bases[kNumClasses] = run;
}

__global__ void __launch_bounds__(256) k_class_lists(u64 batch, const u64 *__restrict__ offL, const u64 *__restrict__ offR,
                                                     u64 *__restrict__ bases, u32 *__restrict__ list)
{
    __shared__ u32 h_count[kNumClasses], h_base[kNumClasses];
    if (threadIdx.x < kNumClasses)
        h_count[threadIdx.x] = 0u;
    __syncthreads();
    const u64 b = (u64)blockIdx.x * 256u + threadIdx.x;
    int cls = -1;
    u32 rank = 0;
    if (b < batch) {
        cls = class_of(offL[b + 1] - offL[b], offR[b + 1] - offR[b]);
        if (cls >= 0)
            rank = atomicAdd(h_count + cls, 1u);
    }
    __syncthreads();
    if (threadIdx.x < kNumClasses && h_count[threadIdx.x])      // the workgroup's stretch of the class's list
        h_base[threadIdx.x] = (u32)atomicAdd(reinterpret_cast<unsigned long long *>(bases + kNumClasses + 1 + threadIdx.x),
                                             (unsigned long long)h_count[threadIdx.x]);
    __syncthreads();
    if (cls >= 0)
        list[bases[cls] + h_base[cls] + rank] = (u32)b;
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
// bs_hint / ti_hint: block size and left terms per tile chosen by mul_plan for short rows (0 = the defaults
// below); the knobs mul_bs (non-zero) and mul_ti (other than its default 4) still override.
template <typename Unit, bool RAGGED>
hipError_t launch_tiled(MulArgs a, u64 pairs, u32 U, hipStream_t s, u32 bs_hint = 0, u32 ti_hint = 0)
{
    const MulTuning tune = mul_tuning();
    // With M > 1 a block size that U divides lets every column of a lane share one LDS read
    // (SAMEK); with the default M == 1 there is a single column per lane and 256 threads
    // (4 KiB-aligned row segments) measured fastest.
    // auto: 256 threads x 4 KiB row segments -- one 16-byte unit per lane, or two 8-byte units
    // when dL is odd (measured at N=1300: 5.3 -> 6.4 TB/s, the 2 KiB segments of M=1 lose)
    const int m_req = tune.m ? tune.m : (sizeof(Unit) == 8 ? 2 : 1);
    u32 bs = tune.bs ? (u32)tune.bs : (tune.m > 1 ? samek_block(U) : (bs_hint ? bs_hint : 256u));
    if (bs == 0)
        bs = 256;
    const bool samek = bs % U == 0;
    const u32 cu = a.t2 * U;
    // do not give a lane more columns than the row has
    int m = m_req;
    while (m > 1 && (u64)bs * (m / 2) >= cu)
        m /= 2;
    // left tile: TI terms, capped so the LDS image stays <= 32 KB
    u32 ti = (ti_hint && tune.ti == 4) ? ti_hint : (u32)tune.ti;
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
            b.out = reinterpret_cast<Unit *>(a.out) + p0 * (a.opitch ? (u64)a.opitch : (u64)a.t1 * a.t2 * U);
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
//     unit per lane -- two when a row is shorter than a workgroup --, linear 4 KiB per workgroup,
//     XCD-contiguous order) AFTER a touch pass that reads one dword of every operand line.  The flat kernel alone stalls on the first touch of
//     every left term (an HBM miss under full write load, 4.6 TB/s at 1024x1024); with the
//     operands already in the memory-side cache it runs at 7.2-7.5 TB/s against 6.9 for the
//     LDS-tiled kernel, 6.2 against 3.9 at 32x32, 5.7 against 4.6 at 8x8.  The touch is a second
//     read of the operands, hence the 4x condition, and is done per <= 64 MB of operands so that
//     they are still in the 256 MB cache when their pairs run.
//   * otherwise rows shorter than half a workgroup (t2*U < 128 units; < 64 for the 8-byte units of an
//     odd dL, whose flat kernel only writes 2 KiB per workgroup): the flat kernel, no touch (the
//     tiled kernel leaves column lanes idle: 1.4 vs 6.0 TB/s at t2 = 1), two units per lane when
//     the product is tall (t1 >= 4).
// Round 3 re-measured the small and thin shapes with operands that the PREVIOUS KERNEL HAD JUST WRITTEN
// (tools/ab_fresh_operands.py): such lines are not in the memory-side cache, so a multiply chain is
// the cold case and the touch stays; the thresholds below come from those runs.
//   * everything else (thin products with long rows, 8-byte units): the LDS-tiled kernel.
//   * and, ahead of all of these, small pairs and thin products with rows of <= 128 units (16-byte units):
//     the LDS-tiled kernel with a 64- or 128-thread workgroup and 8 left terms per tile (see mul_plan).
// CSGN_MUL_FLAT (-1 tiled, k > 0 flat with k units per lane) and CSGN_MUL_TOUCH (0..3: bit 0 left,
// bit 1 right operand) override for sweeps.
struct MulPlan {
    int flat;       // 0 = LDS-tiled kernel, k > 0 = flat kernel with k units per lane
    int touch;      // operands to pull into the memory-side cache first (bit 0 left, bit 1 right)
    u32 bs, ti;     // tiled kernel: threads per workgroup and left terms per tile, 0 = the launcher's defaults
};

static MulPlan mul_plan(size_t unit_bytes, u32 U, u64 t1, u64 t2, u64 pairs)
{
    const MulTuning tune = mul_tuning();
    const u64 PU = t1 * t2 * U;
    MulPlan p = {0, 0, 0, 0};
    if (PU >= (1ull << 31) || tune.flat == -1)
        return p;
    const int touch_env = csgn::tune(TUNE_MUL_TOUCH);
    if (tune.flat > 0) {
        p.flat = tune.flat;
        p.touch = touch_env > 0 ? (touch_env & 3) : 0;
        return p;
    }
    // a call with a few MB of operands is not a stream: they are usually still cached from the
    // kernel that produced them, and two extra launches triple the cost of a small product
    // (class API, 64x64: 10.9 us per multiply with the touch, 3.0-3.8 without)
    const bool streaming = pairs * (t1 + t2) * U * unit_bytes >= (4ull << 20);
    // Knob shared_gpu = 1: the caller is NOT alone on the GPU.  The touch + flat pair counts on its
    // operands staying in the 256 MiB memory-side cache between the touch and the rows that read
    // them; a co-tenant that streams through HBM (a second stream of 1 GiB copies, profiles/r03/
    // cotenant_ab.json) evicts them and the pair falls to 3.5 TB/s, where the LDS-tiled kernel, which
    // fetches each left term once per 4 rows x 4 KiB whatever the cache holds, keeps 4.8.
    const bool shared = csgn::tune(TUNE_SHARED_GPU) != 0;
    // Short rows and small pairs (round 3, operands fresh from the previous kernel, profiles/r03/ab_short_rows*.log):
    // the tiled kernel with a workgroup no wider than the row (64 or 128 threads) and 8 left terms per tile
    // turns a whole small pair, or 8 rows of a thin one, into ONE workgroup -- operands staged once, coalesced,
    // no per-row miss.  8x8 at N=1247 4.8-5.0 -> 5.9 TB/s; 4x4 5.5 -> 6.1; 8x4 / 16x4 / 64x4 5.3 / 5.1 / 4.8 ->
    // 6.0 / 5.9 / 5.9; N=4096 16x4 / 64x4 5.5 -> 6.2.  Taller or wider products (16x8 and up,
    // 10x10, 12x12) stay with the flat kernel, which leads there.
    const u64 row_units = t2 * U;
    const bool touch_regime = unit_bytes == 16 && t1 * t2 >= 4 * (t1 + t2) && ((streaming && !shared) || touch_env > 0);
    if (unit_bytes == 16 && touch_env <= 0 && row_units <= 128u &&
        ((touch_regime && t1 <= 8u) ||                                   // a whole small pair per workgroup
         (!touch_regime && t1 >= 4u && t2 >= 4u && row_units >= 32u && row_units < 128u) ||   // thin and small products
                                                                         // (two-term rows stay flat: config 5's tall x 2
                                                                         // products at N=4096 ran 2-20 % slower tiled)
         (!touch_regime && t1 >= 8u && row_units == 128u))) {            // 128-unit rows: half the default block
        p.bs = row_units <= 64u ? 64u : 128u;
        p.ti = 8u;
        return p;
    }
    if (unit_bytes == 16 && t1 * t2 >= 4 * (t1 + t2) && ((streaming && !shared) || touch_env > 0)) {
        // rows shorter than a workgroup's 256 units: a wave crosses row boundaries, every crossing is a new
        // left term, and two units per lane keep twice the loads in flight -- N=1247 16x16 4.7 -> 5.4 TB/s,
        // 16x8 / 8x16 +5 %, 8x8 +10 % at 32 768 pairs and -2 % at 131 072; with rows of 256 units and more
        // (every N=4096 shape from 8x8 up, the bench shape) one unit per lane stays 1-3 % ahead
        // (profiles/r03/ab_fresh_operands_mid.log)
        p.flat = t2 * U < 256u ? 2 : 1;
        p.touch = touch_env >= 0 ? (touch_env & 3) : 3;
    } else if (t2 * U < (unit_bytes == 16 ? 128u : 64u)) {
        // rows shorter than 2 KiB: the tiled kernel's workgroups would be mostly idle lanes.  From 128 units on it
        // wins on thin products whose operands the previous kernel has just written (round 3, profiles/r03/
        // ab_fresh_operands_thin.log: 4x16 at N=1247 4.65 -> 6.0 TB/s, 4x4 / 16x4 / 64x4 at N=4096 +10-20 %;
        // at 64 units -- 64x2 at N=4096 -- the flat kernel still leads 5.7 to 4.6).
        // Tall products (every few lanes start a new row, i.e. a new left term that nobody has loaded yet) take
        // two output units per lane: twice the loads in flight per wave against the miss latency -- 16x4 at
        // N=1247 4.33 -> 5.03 TB/s, 64x2 / 1024x2 / 16x2 / 4x4 +7-10 % at both N, 2x2 unchanged
        // (profiles/r03/ab_fresh_operands_tall.log)
        p.flat = t1 >= 4 ? 2 : 1;
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

// Both operands in ONE launch (the usual case): a mid-shape stream is cut into <= 64 MB of operands per touch +
// multiply, so a launch saved per cut is several per cent of its time (8x8 at N=1247: 4 touch launches + 2
// multiplies in 76 us).
__global__ void __launch_bounds__(256) k_touch2(const u32 *__restrict__ a, u64 lines_a, const u32 *__restrict__ b,
                                                u64 lines_b, u32 dwords_per_line)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    u32 v = 0;
    if (i < lines_a)
        v = a[i * dwords_per_line];
    else if (i - lines_a < lines_b)
        v = b[(i - lines_a) * dwords_per_line];
    asm volatile("" ::"v"(v));
}

// The same for a slice of a ragged launch: the operands of the pairs that own output terms
// [term_lo, term_hi) are looked up on the device (the host never sees the offsets) and touched,
// unless they are too large a share of the slice's traffic (> 1/4 of its output) or more than the
// memory-side cache will keep (96 MB).
__global__ void __launch_bounds__(256) k_touch_ragged(const u32 *__restrict__ L, const u64 *__restrict__ offL,
                                                      const u32 *__restrict__ R, const u64 *__restrict__ offR,
                                                      const u64 *__restrict__ offOut, u32 batch, u64 term_lo,
                                                      u64 term_hi, u64 term_bytes, const u64 *__restrict__ d_gate)
{
    if (d_gate) {                                    // csgn_mul_ragged_async: slices are cut from the caller's bound
        term_hi = min(term_hi, d_gate[0]);
        if (term_lo >= term_hi)
            return;
    }
    // the first and the last pair of the slice: two 64-ary wave searches side by side (waves 0 and 1), three or
    // four round trips in all -- the two binary searches that stood here were ~28 dependent loads, most of the
    // kernel's 20 us (PMC round 3: 60 VALU instructions per wave, parked 94 % of the time)
    __shared__ u32 s_p[2];
    const u32 wave = threadIdx.x >> 6;
    if (wave < 2u) {
        const u32 f = wave_find(offOut, 0u, batch, wave == 0u ? term_lo : term_hi - 1);
        if ((threadIdx.x & (kWave - 1)) == 0u)
            s_p[wave] = f;
    }
    __syncthreads();
    const u32 p_lo = s_p[0], p_hi = s_p[1];
    const u64 lb = offL[p_lo] * term_bytes, le = offL[p_hi + 1] * term_bytes;
    const u64 rb = offR[p_lo] * term_bytes, re = offR[p_hi + 1] * term_bytes;
    const u64 op = (le - lb) + (re - rb);
    if (op > (96ull << 20) || op * 4 > (term_hi - term_lo) * term_bytes)
        return;
    const u64 stride = (u64)gridDim.x * 256u * 128u;
    const u64 first = ((u64)blockIdx.x * 256u + threadIdx.x) * 128u;
    // four lines per thread in flight at a time (the values are not used; the asm only keeps the loads)
    auto touch = [&](const u32 *__restrict__ base, u64 b, u64 e) {
        for (u64 a = b + first; a < e; a += 4u * stride) {
            u32 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                v[i] = base[min(a + (u64)i * stride, e - 4u) >> 2];
            asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
        }
    };
    if (le - lb >= 4u)
        touch(L, lb, le);
    if (re - rb >= 4u)
        touch(R, rb, re);
}

// One uniform chunk (pairs are contiguous in L, R and out).
template <typename Unit>
hipError_t mul_uniform_chunk(u32 U, u64 pairs, u64 call_pairs, u32 t1, u32 t2, const u64 *L, const u64 *R,
                             u64 *out, hipStream_t s, u32 opitch = 0)
{
    const Unit *Lu = reinterpret_cast<const Unit *>(L);
    const Unit *Ru = reinterpret_cast<const Unit *>(R);
    Unit *Ou = reinterpret_cast<Unit *>(out);
    const u64 PU = (u64)t1 * t2 * U;
    const u64 total = pairs * PU;
    if (total == 0)
        return hipSuccess;
    const MulTuning tune = mul_tuning();
    if (t1 == 1 && t2 == 1 && !opitch) {
        // at most 2^31-1 workgroups of 256 units per launch
        const u64 per_launch = kMaxBlocks256 * 256u;
        for (u64 u0 = 0; u0 < total; u0 += per_launch) {
            const u64 nu = (total - u0 < per_launch) ? total - u0 : per_launch;
            k_and_stream<Unit, true><<<ceil_div_u64(nu, 256u), 256, 0, s>>>(Lu + u0, Ru + u0, Ou + u0, nu,
                                                                            stream_xcd(nu));
            hipError_t e = hipGetLastError();
            if (e != hipSuccess)
                return e;
        }
        return hipSuccess;
    }
    // the kernel choice looks at the whole call (a batch streamed through a small arena is still a stream)
    const MulPlan plan = mul_plan(sizeof(Unit), U, t1, t2, call_pairs);
    if (plan.flat) {
        const int mf = plan.flat;
        // left-term prefetch from inside the kernel (only without the touch pass, only when a row
        // fills a workgroup): ~6 MB of output ahead per XCD stream
        u32 pfr = 0;
        if (!plan.touch && (u64)t2 * U >= 256u) {
            const int pf_kb = csgn::tune(TUNE_MUL_PF_KB);
            const u64 ahead = (u64)(pf_kb >= 0 ? pf_kb : (tune.xcd ? 6144 : 49152)) << 10;
            const u64 row_bytes = (u64)t2 * U * sizeof(Unit);
            pfr = (u32)((ahead + row_bytes - 1) / row_bytes);
        }
        u64 pairs_per = (0xFFFFFF00ull / PU) ? (0xFFFFFF00ull / PU) : 1;   // units (= threads) per launch < 2^32
        if (plan.touch) {
            // touched operands must still be in the 256 MB memory-side cache when their pair runs:
            // at most 64 MB of them per touch + launch (96-200 MB lose 10-25 %, round 3), and 32 MB for
            // products of 1 Ki to 64 Ki terms when the call has to be cut anyway -- 32x32 and 64x64 gain 5-9 %
            // at both N with the smaller cut, 8x8 and 16x16 lose 1-8 % (their launches are too short to
            // halve), 256x256 and up do not care; a call that fits one 64 MB cut is left whole (cutting a
            // 42 MB call in two cost it 5 %)  (profiles/r03/ab_touch_chunk*_experiment.log)
            const u64 op_bytes = (u64)(t1 + t2) * U * sizeof(Unit);
            const bool mid = (u64)t1 * t2 >= 1024u && (u64)t1 * t2 < 65536u && pairs * op_bytes > (64ull << 20);
            const u64 cut = mid ? (32ull << 20) : (64ull << 20);
            pairs_per = std::min<u64>(pairs_per, std::max<u64>(1, cut / op_bytes));
        }
        const FastDiv dPU = csgn_fastdiv_make((u32)PU), dCU = csgn_fastdiv_make(t2 * U),
                      dU = csgn_fastdiv_make(U);
        for (u64 p0 = 0; p0 < pairs; p0 += pairs_per) {
            const u64 np = (pairs - p0 < pairs_per) ? pairs - p0 : pairs_per;
            const u32 tot = (u32)(np * PU);
            const u32 trows = (u32)(np * t1);
            const u32 blocks = ceil_div_u64(tot, 256u * (u64)mf);
            const Unit *Lc = Lu + p0 * t1 * U, *Rc = Ru + p0 * t2 * U;
            Unit *Oc = Ou + p0 * (opitch ? (u64)opitch : PU);
            if (plan.touch) {
                const u32 lb = 128;       // the L2 fills whole 128-byte lines (a 256-byte stride loses the gain)
                const u64 ll = (np * t1 * U * sizeof(Unit) + lb - 1) / lb, rl = (np * t2 * U * sizeof(Unit) + lb - 1) / lb;
                if ((plan.touch & 3) == 3)
                    k_touch2<<<ceil_div_u64(ll + rl, 256u), 256, 0, s>>>(reinterpret_cast<const u32 *>(Lc), ll,
                                                                         reinterpret_cast<const u32 *>(Rc), rl, lb / 4);
                else if (plan.touch & 1)
                    k_touch<<<ceil_div_u64(ll, 256u), 256, 0, s>>>(reinterpret_cast<const u32 *>(Lc), ll, lb / 4);
                else if (plan.touch & 2)
                    k_touch<<<ceil_div_u64(rl, 256u), 256, 0, s>>>(reinterpret_cast<const u32 *>(Rc), rl, lb / 4);
            }
#define CSGN_FLAT(MF)                                                                                  \
    do {                                                                                               \
        if (opitch)                                                                                    \
            k_mul_flat<Unit, MF, true, true><<<blocks, 256, 0, s>>>(Lc, Rc, Oc, tot, t1, t2, U, dPU, dCU, dU, pfr, trows, opitch);  \
        else if (tune.xcd)                                                                             \
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
    a.opitch = opitch;
    return launch_tiled<Unit, false>(a, pairs, U, s, plan.bs, plan.ti);
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
    t.m = tune(TUNE_MUL_M);          // 0 = auto: 1 column unit per lane (2 for 8-byte units)
    if (t.m != 1 && t.m != 2 && t.m != 4 && t.m != 8)
        t.m = 0;
    t.ti = tune(TUNE_MUL_TI);
    if (t.ti < 1)
        t.ti = 1;
    t.nt = tune(TUNE_MUL_NT) ? 1 : 0;
    t.flat = tune(TUNE_MUL_FLAT);   // 0 = auto (mul_plan); >0 = flat with that unroll; -1 = always tiled
    if (t.flat != -1 && t.flat != 1 && t.flat != 2 && t.flat != 4 && t.flat != 8)
        t.flat = 0;
    t.xcd = tune(TUNE_MUL_XCD);      // 0 = dispatch order, 1 = remap flat kernel, 2 = remap both
    if (t.xcd < 0 || t.xcd > 2)
        t.xcd = 1;
    t.bs = tune(TUNE_MUL_BS);
    if (t.bs % 64 != 0 || t.bs < 64 || t.bs > 512)
        t.bs = 0;
    return t;
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
                       u64 out_slots, hipStream_t s, u64 out_pitch_words)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0 || t1 == 0 || t2 == 0)
        return hipSuccess;
    const bool wide = (dL % 2 == 0) && aligned16(L) && aligned16(R) && aligned16(out) && out_pitch_words % 2 == 0;
    const u32 U = (u32)(wide ? dL / 2 : dL);
    if (out_pitch_words) {
        // placed output (csgn_circuit.hip): element e's product at out + e * out_pitch_words, no arena
        if (out_pitch_words < t1 * t2 * dL || out_pitch_words >= (1ull << 32))
            return hipErrorInvalidValue;
        const u32 opitch = (u32)(wide ? out_pitch_words / 2 : out_pitch_words);
        return wide ? mul_uniform_chunk<unit16>(U, batch, batch, (u32)t1, (u32)t2, L, R, out, s, opitch)
                    : mul_uniform_chunk<unit8>(U, batch, batch, (u32)t1, (u32)t2, L, R, out, s, opitch);
    }
    const u64 slots = (out_slots == 0 || out_slots > batch) ? batch : out_slots;
    for (u64 p0 = 0; p0 < batch; p0 += slots) {
        const u64 np = (batch - p0 < slots) ? batch - p0 : slots;
        const u64 *Lc = L + p0 * t1 * dL;
        const u64 *Rc = R + p0 * t2 * dL;
        hipError_t e = wide ? mul_uniform_chunk<unit16>(U, np, batch, (u32)t1, (u32)t2, Lc, Rc, out, s)
                            : mul_uniform_chunk<unit8>(U, np, batch, (u32)t1, (u32)t2, Lc, Rc, out, s);
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

// [head][scan block: one granule per chunk of pairs, ticket][class bases + cursors][class lists: one u32 per pair]
static u64 plan_bases_at(u64 batch) { return kPlanHeadWords + (batch + 1 + 1023) / 1024 + 4; }   // (room for more granules than k_plan uses)
static u64 plan_lists_at(u64 batch) { return plan_bases_at(batch) + 2 * kNumClasses + 2; }
u64 mul_ragged_plan_scratch_words(u64 batch) { return plan_lists_at(batch) + (batch + 1) / 2 + 1; }
u64 mul_ragged_plan_head_words() { return kPlanHeadWords; }

hipError_t mul_ragged_plan(u64 batch, const u64 *offL, const u64 *offR, u64 *offOut, u64 *d_work,
                           hipStream_t s, u64 *gate, u64 capacity_terms, bool can_stream)
{
    // d_work: [head: plan4 (total, max t1, max t2, max t1*t2), huge count + records, operand totals, checksum slots,
    //          class histogram][scan: one granule per 4096-pair chunk, ticket][class bases][class lists]
    u64 *scan = d_work + kPlanHeadWords;
    const u64 nchunks = batch ? (batch + kPlanChunk - 1) / kPlanChunk : 1;
    hipError_t e = zero_words(d_work, plan_lists_at(batch), s);       // head, scan block, class bases (a kernel: see zero_words)
    if (e != hipSuccess)
        return e;
    if (nchunks > kMaxBlocks1024)
        return hipErrorInvalidValue;
    // the size-class lists (and the histogram's global atomics) only when the multiply is going to use them
    const bool classes = csgn::tune(TUNE_RAGGED_CLASSES) == 1;
    const u32 vec = ((reinterpret_cast<uintptr_t>(offL) | reinterpret_cast<uintptr_t>(offR) | reinterpret_cast<uintptr_t>(offOut)) & 15u) == 0u;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 0;                                                   // unknown: tickets always
    k_plan<<<(u32)nchunks, kPlanThreads, 0, s>>>(batch, (u32)nchunks, offL, offR, offOut, d_work, scan, classes ? 1u : 0u, gate,
                                                 capacity_terms, can_stream ? 1u : 0u, vec, (u32)cus);
    if (classes && batch) {
        u64 *bases = d_work + plan_bases_at(batch);
        k_class_bases<<<1, 1, 0, s>>>(d_work, bases);
        k_class_lists<<<ceil_div_u64(batch, 256), 256, 0, s>>>(batch, offL, offR, bases,
                                                               reinterpret_cast<u32 *>(d_work + plan_lists_at(batch)));
    }
    return hipGetLastError();
}

// What a plan learned beyond its four numbers: the batch's huge pairs and operand size.  It belongs to
// an explicit csgn_mul_plan object of the caller (round 4; round 3 kept it in hidden per-thread state
// matched by array addresses, which a caller that rewrote offsets in place could fool -- ADVICE r3).
void mul_plan_notes_from_head(MulPlanNotes &r, const u64 *offL, const u64 *offR, const u64 *offOut, u64 batch,
                              const u64 *h_head, const u64 *d_work)
{
    static_assert(kNumClasses == 28, "MulPlanNotes is sized for 28 classes");
    r.lists = d_work ? reinterpret_cast<const u32 *>(d_work + plan_lists_at(batch)) : nullptr;
    r.cls_base[0] = 0;
    for (u32 c = 0; c < kNumClasses; ++c) {
        r.cls_pairs[c] = h_head[kClassAt + c];
        r.cls_terms[c] = h_head[kClassAt + kNumClasses + c];
        r.cls_base[c + 1] = r.cls_base[c] + r.cls_pairs[c];
    }
    r.offL = offL;
    r.offR = offR;
    r.offOut = offOut;
    r.batch = batch;
    r.total = h_head[0];
    r.max_t1 = h_head[1];
    r.max_t2 = h_head[2];
    r.operand_terms = h_head[5 + kHugeRecords * 6] + h_head[6 + kHugeRecords * 6];
    r.checksum = 0;
    for (u32 i = 0; i < kSumSlots; ++i)
        r.checksum += h_head[kSumAt + i];
    const u64 count = h_head[4];
    r.n = (u32)std::min<u64>(count, kHugeRecords);
    for (u32 i = 0; i < r.n; ++i)
        for (int k = 0; k < 6; ++k)
            r.rec[i][k] = h_head[5 + i * 6 + k];
    // the slots were handed out by an atomic counter: order the records by pair (insertion sort, <= 32)
    for (u32 i = 1; i < r.n; ++i)
        for (u32 j = i; j > 0 && r.rec[j][0] < r.rec[j - 1][0]; --j)
            for (int k = 0; k < 6; ++k)
                std::swap(r.rec[j][k], r.rec[j - 1][k]);
}

// checksum of the three offset arrays as they are NOW (the same sum k_plan_fix accumulates): one word
__global__ void __launch_bounds__(256) k_offsets_checksum(u64 batch, const u64 *__restrict__ offL,
                                                          const u64 *__restrict__ offR,
                                                          const u64 *__restrict__ offOut, u64 *__restrict__ sum)
{
    u64 mine = 0;
    for (u64 b = (u64)blockIdx.x * 256u + threadIdx.x; b <= batch; b += (u64)gridDim.x * 256u)
        mine += offsets_mix(b, offL[b], offR[b], offOut[b]);
    __shared__ u64 wsum[4];
    for (int off = 32; off > 0; off >>= 1)
        mine += (u64)__shfl_down(mine, off, 64);
    if ((threadIdx.x & (kWave - 1)) == 0)
        wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(reinterpret_cast<unsigned long long *>(sum + (blockIdx.x % kSumSlots)),
                  (unsigned long long)(wsum[0] + wsum[1] + wsum[2] + wsum[3]));
}

u32 offsets_checksum_words() { return kSumSlots; }

hipError_t offsets_checksum(u64 batch, const u64 *offL, const u64 *offR, const u64 *offOut, u64 *d_sum, hipStream_t s)
{
    hipError_t e = zero_words(d_sum, kSumSlots, s);
    if (e != hipSuccess)
        return e;
    const u32 blocks = (u32)std::min<u64>((batch + 256) / 256, 2048);
    k_offsets_checksum<<<blocks, 256, 0, s>>>(batch, offL, offR, offOut, d_sum);
    return hipGetLastError();
}

hipError_t mul_ragged(u64 n_bits, u64 batch, const u64 *L, const u64 *offL, const u64 *R,
                      const u64 *offR, u64 *out, const u64 *offOut, u64 max_t1, u64 max_t2,
                      u64 total_out_terms, hipStream_t s, const MulPlanNotes *notes, u64 operand_terms,
                      const u64 *d_gate, const u64 *d_huge)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0 || max_t1 == 0 || max_t2 == 0 || total_out_terms == 0)
        return hipSuccess;
    if (batch >= (1ull << 32))
        return hipErrorInvalidValue;
    // Every pair has the largest shape (t1_b <= max_t1, t2_b <= max_t2 and the products sum to
    // batch * max_t1 * max_t2): the CSR arrays describe a UNIFORM batch, so no lane has to look
    // anything up -- a million fresh 1x1 pairs given as a ragged batch run k_and_stream (6.2 TB/s
    // instead of 3.4 through the CSR kernel).
    if (!d_gate && total_out_terms == batch * max_t1 * max_t2 && csgn::tune(TUNE_RAGGED_FLAT) == 0)
        return mul_uniform(n_bits, batch, max_t1, max_t2, L, R, out, 0, s);
    const bool wide = (dL % 2 == 0) && aligned16(L) && aligned16(R) && aligned16(out);
    const u32 U = (u32)(wide ? dL / 2 : dL);
    // Nearly uniform batches of large products keep the LDS-tiled kernel (one grid sized for the
    // largest shape); anything skewed or small goes through the flat ragged kernel, whose grid
    // is the real output.
    const bool tiled = !d_gate && csgn::tune(TUNE_RAGGED_FLAT) == 0 && max_t1 * max_t2 * U > 8192 &&
                       batch * max_t1 * max_t2 <= 2 * total_out_terms;
    // Batches of small pairs with small MAXIMA (rows of at most 256 units, at most 16 rows, mean product at least a
    // quarter of the largest): the tiled kernel with ONE narrow workgroup per pair and 8-row tile -- the pair is the
    // block index, so there is no lookup of any kind, where the CSR kernel below pays the offset window in every turn.
    // A million pairs of 0-5 x 0-5 terms 4.50 -> 5.10 TB/s, 2^18 pairs of 4-12 x 4-12 3.46 -> 4.26 (end of round 3).
    const bool small_tiled = !d_gate && csgn::tune(TUNE_RAGGED_FLAT) == 0 && wide && max_t2 * U <= 256 && max_t1 <= 16 &&
                             batch * max_t1 * max_t2 <= 4 * total_out_terms;
    if (small_tiled) {
        MulArgs a = {};
        a.L = L;
        a.R = R;
        a.out = out;
        a.offL = offL;
        a.offR = offR;
        a.offOut = offOut;
        a.t1 = (u32)max_t1;
        a.t2 = (u32)max_t2;
        return launch_tiled<unit16, true>(a, batch, U, s, max_t2 * U <= 64 ? 64u : max_t2 * U <= 128 ? 128u : 256u, 8u);
    }
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
    // Size classes (round 4; an experiment kept behind a knob): in a long-tailed batch of small pairs
    // nearly every 16 KiB turn of the CSR kernel pays the offset window and a 9-step search (3.7 TB/s at a log-normal
    // mean of 8 x 8 terms).  With the plan's lists at hand, every
    // non-empty class gets ONE launch of the LDS-tiled kernel shaped for the class (pair = list[block / tiles]; a
    // pair smaller than its class leaves early from the tiles it does not fill: at most 2x per side), and the CSR
    // kernel below skips what they wrote.
    u32 skip_t1 = 0, skip_t2 = 0;
    bool only_classes = false;
    {
        static const MulPlanNotes kNone = MulPlanNotes();
        const MulPlanNotes &cn = notes ? *notes : kNone;
        const bool match = notes && cn.lists && cn.offL == offL && cn.offR == offR && cn.offOut == offOut &&
                           cn.batch == batch && cn.total == total_out_terms;
        u64 class_terms = 0;
        for (u32 c = 0; c < kNumClasses; ++c)
            class_terms += cn.cls_terms[c];
        // MEASURED SLOWER (profiles/r04/ragged_size_classes.log: log-normal mean 8 x 8 2.6 TB/s against the CSR kernel's
        // 3.6, mean 16 x 16 2.8 against 4.4, mean 32 x 32 3.6 against 5.4): every class launch writes a sparse subset
        // of the output and a third to a half of its workgroups find nothing to do.  Off unless knob ragged_classes = 1
        // (the plan only builds the lists then).
        const bool want = csgn::tune(TUNE_RAGGED_CLASSES) == 1;
        if (match && wide && !d_gate && want && class_terms && csgn::tune(TUNE_RAGGED_FLAT) == 0) {
            for (u32 c = 0; c < kNumClasses; ++c) {
                if (!cn.cls_pairs[c])
                    continue;
                MulArgs a = {};
                a.L = L;
                a.R = R;
                a.out = out;
                a.offL = offL;
                a.offR = offR;
                a.offOut = offOut;
                a.t1 = class_max_t1((int)c);
                a.t2 = class_max_t2((int)c);
                a.pair_list = cn.lists + cn.cls_base[c];
                const u32 cu = a.t2 * U;
                const u32 bs = cu <= 64u ? 64u : cu <= 128u ? 128u : 256u;
                const hipError_t e = launch_tiled<unit16, true>(a, cn.cls_pairs[c], U, s, bs, 8u);
                if (e != hipSuccess)
                    return e;
            }
            skip_t1 = kClsMaxT1;
            skip_t2 = kClsMaxT2;
            only_classes = class_terms == total_out_terms;
        }
    }
    if (only_classes)
        return hipSuccess;
    // 4 KiB chunks per workgroup: as many as 8 while the grid keeps >= 8192 workgroups (ragged_chunks).  Round 3,
    // cold, profiles/r03/bench_ragged.log: on the 2.8 GB log-normal batch C=4 and C=8 are within run-to-run noise
    // (5.1-5.5 TB/s each over three runs), C=16 4.85, C=2 5.2, C=1 3.9; the 178 MB skewed batch C=1 4.74, C=2 4.66,
    // C=4 4.47, C=8 4.28 -- small outputs want many short workgroups, which the rule gives them.  A per-workgroup
    // start table from a pre-kernel (instead of each workgroup's 64-ary search) was tried and measured 0-8 % SLOWER
    // on all four batches, so it is not here.
    // Batches of small pairs (mean product under 512 terms) take 16: nearly every turn of theirs pays the offset
    // window, the start-up search is the one phase a longer workgroup saves, and +3-5 % on four such batches says so
    // (pairs of 0-5, 4-12, mean-8 and mean-16 log-normal terms: 3.66 / 3.03 / 2.94 / 4.73 -> 3.87 / 3.19 / 3.05 / 4.92);
    // at mean 32x32 it costs 5-10 %.
    int chunks = ragged_chunks(total_units);
    if (chunks == 8 && csgn::tune(TUNE_RAGGED_C) == 0 && total_out_terms / batch < 512u &&
        total_units / (256u * 16u) >= 8192u)
        chunks = 16;
    const u32 pf_pairs = (u32)std::max(0, csgn::tune(TUNE_RAGGED_PF));    // operand prefetch distance in pairs, 0 = off
    const u32 xcd_group = (u32)std::max(0, csgn::tune(TUNE_RAGGED_XCD_GROUP));   // logical blocks per XCD turn, 0 = contiguous eighths
    const int turn = csgn::tune(TUNE_RAGGED_M);                           // 4 KiB chunks that share one pair bet: 1, 2, 4
    // The plan this thread made for exactly these offset arrays, if any (the documented sequence
    // csgn_mul_ragged_plan -> csgn_mul_ragged): it knows the batch's operand size and its huge pairs.
    static const MulPlanNotes kNoNotes = MulPlanNotes();
    const MulPlanNotes &rp = notes ? *notes : kNoNotes;
    const bool remembered = notes && rp.offL == offL && rp.offR == offR && rp.offOut == offOut &&
                            rp.batch == batch && rp.total == total_out_terms && rp.max_t1 == max_t1 &&
                            rp.max_t2 == max_t2;
    // Large outputs go in slices, each preceded by a touch of the operands its pairs need
    // (k_touch_ragged): the flat kernel's first touch of a left term is then a cache hit instead of
    // an HBM miss under full write load, as in the uniform path.  Knob ragged_touch = 0 turns it off.
    // A slice is 1 GiB of output unless the plan says that would bring more than ~80 MB of operands with
    // it -- about what the memory-side cache keeps under the write stream (the uniform path cuts at 64 MB)
    // and short of the 96 MB above which k_touch_ragged gives up: then 512 MiB.  A log-normal batch of
    // mean 16x16 (344 MB of operands for 2.7 GB of products) ran untouched at 3.6 TB/s in 1 GiB slices
    // and runs at 4.7 in 512 MiB ones; mean 32x32 and 64x64 keep 1 GiB (5.6-5.8; 5.3 and 4.7 in smaller
    // slices) -- profiles/r03/ab_ragged_slice_experiment.log.
    // Where even a 512 MiB slice would come with more than that -- batches of small pairs, whose operands are a
    // quarter of their products and more -- nothing is sliced or touched: a dozen 256 MiB slices, each a touch and
    // a launch of its own, ran 7-16 % SLOWER than one untouched launch (pairs of 0-5, 4-12 and mean-8 log-normal
    // terms: 3.7 / 3.0 / 2.9 TB/s sliced, 4.3 / 3.2 / 3.4 whole).
    u64 slice_units = 1ull << 26;
    bool slice_touch = true;
    if (remembered && rp.operand_terms != 0)
        operand_terms = rp.operand_terms;
    if (operand_terms != 0) {
        const u64 operand_units = operand_terms * U;
        auto too_much = [&](u64 su) {
            return (unsigned __int128)su * operand_units > (unsigned __int128)total_units * ((80ull << 20) / sizeof(unit16));
        };
        if (too_much(slice_units))
            slice_units >>= 1;
        if (too_much(slice_units))
            slice_touch = false;
    } else if (total_out_terms / batch < 512u) {
        // operand size unknown (no plan at hand: csgn_mul_ragged, csgn_mul_ragged_async, whose total is the caller's bound):
        // a batch that averages under 512 product terms a pair is taken for one of small pairs with a large operand share
        slice_units >>= 1;
    }
    if (csgn::tune(TUNE_RAGGED_SLICE_MB) > 0) {                  // (experiments: slices of this many MiB, touched)
        slice_units = ((u64)csgn::tune(TUNE_RAGGED_SLICE_MB) << 20) / sizeof(unit16);
        slice_touch = true;
    }
    // The wave-cooperative kernel (k_mul_ragged_coop): a pair's stretch is addressed by 32-bit unit offsets from the
    // pair's bases, so the largest product of the batch must stay under 2^28 units (4 GiB).
    // (csgn_mul_ragged_async does not know the shapes: the caller's bound on the whole output stands in.)
    const u64 pair_bound = d_gate ? total_out_terms : max_t1 * max_t2;
    // It is for SMALL pairs, not tiny ones: a single-term pair costs a wave a block of its own, and a stretch of
    // them leaves the wave with a sixth of its lanes at work (65 535 singles: 3.1 TB/s against the CSR kernel's 3.5, a
    // million: 1.7 against 4.0).  Auto (-1): where the range averages 16 product terms a pair and more (below).
    const int coop_mode = csgn::tune(TUNE_RAGGED_COOP);
    const bool coop_ok = coop_mode != 0 && pair_bound * U < (1ull << 28) && skip_t1 == 0;
    // csgn_mul_ragged_async with the plan's records at hand: the huge pairs by the launch's first workgroups (mul_huge_tiles), the CSR kernel around them
    // -- where the CSR kernel is what the (one) range gets; the wave-cooperative kernel has no skip and takes everything
    const u64 *huge_csr = nullptr;
    if (d_gate && d_huge && csgn::tune(TUNE_RAGGED_FLAT) == 0) {
        const u64 all_terms = total_units / U;
        const bool coop_all = coop_ok && (coop_mode == 1 || all_terms >= 32u * batch);
        // (and the 1024 workgroups in front must fit the launch: always, short of a 60 GB output)
        if (!coop_all && ceil_div_u64(total_units, 256u * (u32)chunks) + 1024u <= kMaxBlocks256)
            huge_csr = d_huge;
    }
    auto flat_range = [&](u64 range_begin, u64 range_end, u64 range_pairs) -> hipError_t {
        const u64 range_units = range_end - range_begin;
        // (the average is taken WITHOUT the largest product the shapes allow: one 1024 x 1024 pair among 65 535 singles
        // averages 17 terms a pair and is no batch of small pairs; where the shapes are unknown -- the async call -- the
        // bar is 32: that batch runs at 2.0 TB/s here and 2.8-3.5 in the CSR kernel)
        const u64 range_terms = range_units / U;
        const bool small_pairs = d_gate ? range_terms >= 32u * range_pairs
                                        : range_terms - std::min(range_terms, max_t1 * max_t2) >= 16u * range_pairs;
        const bool coop = coop_ok && (coop_mode == 1 || small_pairs);
        // Slices behind k_touch_ragged where 1 GiB slices do (log-normal mean 32 x 32, cold: 5.7 TB/s sliced, 5.2-5.3 in one
        // launch).  Where the operand share asks for 512 MiB slices, a range the wave-cooperative kernel takes goes in ONE
        // launch whose waves touch their own operands (mean 16 x 16: 5.2 against 4.9-5.0 sliced), as do the batches of
        // smaller pairs that are never sliced (mean 8 x 8: 4.8-5.3 against 4.0-4.5 without any touch).
        const bool own_touch = coop && slice_units < (1ull << 26) && csgn::tune(TUNE_RAGGED_COOP_TOUCH) > 0 && csgn::tune(TUNE_RAGGED_SLICE_MB) <= 0;
        const bool touch = wide && slice_touch && !own_touch && range_units > slice_units && csgn::tune(TUNE_RAGGED_TOUCH) != 0;
        const u64 per_launch = touch ? slice_units : kMaxBlocks256 * 256u;   // units
        hipError_t result = hipSuccess;
        for (u64 u0 = range_begin; u0 < range_end && result == hipSuccess; u0 += per_launch) {
            const u64 nu = (range_end - u0 < per_launch) ? range_end - u0 : per_launch;
            const u32 blocks = ceil_div_u64(nu, 256u * (u32)chunks);
            // (the huge pairs' workgroups ride in front of the range's FIRST launch: 1024 of them, four to a CU)
            const u32 hb = (huge_csr && u0 == range_begin && range_begin == 0) ? 1024u : 0u;
            if (touch)
                k_touch_ragged<<<2048, 256, 0, s>>>(reinterpret_cast<const u32 *>(L), offL,
                                                   reinterpret_cast<const u32 *>(R), offR, offOut, (u32)batch,
                                                   u0 / U, (u0 + nu + U - 1) / U, (u64)dL * 8u, d_gate);
            if (coop) {
                // the whole output in one launch: the virtual axis with 64 units of padding per pair, stretches sized for
                // >= 16 K waves but 8 to 64 blocks each; a slice of a large output: the unit axis, 16 blocks a wave
                const bool whole_output = u0 == 0 && nu == total_units;
                const u32 vw = whole_output ? kWave : 0u;
                const u64 v_begin = u0, v_end = u0 + nu + (u64)vw * batch;
                // (at most 64 blocks a wave; 32 where the range averages under 128 product terms a pair -- log-normal mean 8 x 8:
                // +4 % on two boxes, mean 16 x 16 +-0, mean 32 x 32 -4 % with 32)
                const u64 span_cap = range_terms < 128u * range_pairs ? 2048u : 4096u;
                u32 span = whole_output ? (u32)std::min<u64>(span_cap, std::max<u64>(512, ((v_end - v_begin) / 16384u) & ~63ull)) : 1024u;
                if (csgn::tune(TUNE_RAGGED_COOP_SPAN) > 0)
                    span = kWave * (u32)csgn::tune(TUNE_RAGGED_COOP_SPAN);
                const u64 wgs = (v_end - v_begin + 4ull * span - 1) / (4ull * span);
                if (wgs > kMaxBlocks256)
                    return hipErrorInvalidValue;
                const bool k2 = csgn::tune(TUNE_RAGGED_COOP_K) == 2, pipe = csgn::tune(TUNE_RAGGED_COOP_PIPE) != 0;
                const u32 coop_group = csgn::tune(TUNE_RAGGED_COOP_XCD_GROUP) > 0 ? (u32)csgn::tune(TUNE_RAGGED_COOP_XCD_GROUP) : 0u;
                // the in-kernel operand touch (up to this many terms a side and window; knob in KiB): not behind a slice's own touch
                const u32 touch_terms = touch ? 0u : (u32)(((u64)std::max(0, csgn::tune(TUNE_RAGGED_COOP_TOUCH)) << 10) / ((u64)dL * 8u));
#define CSGN_COOP(UNIT, KK)                                                                         \
    if (pipe)                                                                                       \
        CSGN_COOP_(UNIT, KK, true);                                                                 \
    else                                                                                            \
        CSGN_COOP_(UNIT, KK, false)
#define CSGN_COOP_(UNIT, KK, PP)                                                                    \
    k_mul_ragged_coop<UNIT, KK, PP><<<(u32)wgs, 256, 0, s>>>(                                           \
        reinterpret_cast<const UNIT *>(L), offL, reinterpret_cast<const UNIT *>(R), offR,           \
        reinterpret_cast<UNIT *>(out), offOut, (u32)batch, v_begin, v_end, U, dU, span, vw, d_gate, touch_terms, coop_group)
                if (wide) {
                    if (k2) {
                        CSGN_COOP(unit16, 2);
                    } else {
                        CSGN_COOP(unit16, 4);
                    }
                } else {
                    if (k2) {
                        CSGN_COOP(unit8, 2);
                    } else {
                        CSGN_COOP(unit8, 4);
                    }
                }
#undef CSGN_COOP
#undef CSGN_COOP_
                result = hipGetLastError();
                continue;
            }
#define CSGN_RAGGED_FLAT(CH, MM)                                                                    \
    do {                                                                                            \
        if (wide)                                                                                   \
            k_mul_ragged_flat<unit16, CH, MM><<<blocks + hb, 256, 0, s>>>(                          \
                reinterpret_cast<const unit16 *>(L), offL, reinterpret_cast<const unit16 *>(R), offR, \
                reinterpret_cast<unit16 *>(out), offOut, (u32)batch, u0, u0 + nu, U, dU, pf_pairs,  \
                d_gate, skip_t1, skip_t2, xcd_group, huge_csr, hb);                         \
        else                                                                                        \
            k_mul_ragged_flat<unit8, CH, MM><<<blocks + hb, 256, 0, s>>>(L, offL, R, offR, out, offOut,  \
                                                                    (u32)batch, u0, u0 + nu, U, dU, \
                                                                    pf_pairs, d_gate, skip_t1, skip_t2, xcd_group, huge_csr, hb); \
    } while (0)
#define CSGN_RAGGED_LAUNCH(CH)                                                                      \
    do {                                                                                            \
        if (turn >= 4 && (CH) % 4 == 0)                                                             \
            CSGN_RAGGED_FLAT(CH, ((CH) % 4 == 0 ? 4 : 1));                                          \
        else if (turn >= 2 && (CH) % 2 == 0)                                                        \
            CSGN_RAGGED_FLAT(CH, ((CH) % 2 == 0 ? 2 : 1));                                          \
        else                                                                                        \
            CSGN_RAGGED_FLAT(CH, 1);                                                                \
    } while (0)
            switch (chunks) {
            case 1: CSGN_RAGGED_LAUNCH(1); break;
            case 2: CSGN_RAGGED_LAUNCH(2); break;
            case 4: CSGN_RAGGED_LAUNCH(4); break;
            case 16: CSGN_RAGGED_LAUNCH(16); break;
            default: CSGN_RAGGED_LAUNCH(8); break;
            }
#undef CSGN_RAGGED_FLAT
#undef CSGN_RAGGED_LAUNCH
            result = hipGetLastError();
        }
        return result;
    };
    // Huge pairs the plan wrote down (same offset arrays, same batch, same plan numbers as the last
    // csgn_mul_ragged_plan of this thread): each gets the UNIFORM kernels on its own sub-buffers -- no
    // lookup of any kind inside what is usually nearly all of a skewed batch's output -- provided it
    // is worth a launch of its own (24 MB of output: ~3.5 us of HBM time against ~3 launches); the CSR
    // kernel runs on the stretches between them.  Knob ragged_flat = 1 keeps everything in the CSR kernel.
    const bool planned = remembered && csgn::tune(TUNE_RAGGED_FLAT) == 0 && rp.n != 0;
    u64 cursor = 0, cursor_pair = 0;
    // ... and only when those pairs are most of the batch: every split costs launches (a uniform one and the
    // CSR kernel's on either side, each with its own ramp and tail), which a few 30 MB pairs inside a 2.8 GB
    // log-normal batch do not repay (measured: 5.03 TB/s split, 5.35 unsplit)
    bool split = false;
    if (planned) {
        u64 huge_terms = 0;
        for (u32 i = 0; i < rp.n; ++i)
            if (rp.rec[i][3] * rp.rec[i][4] * dL * 8u >= (24ull << 20))
                huge_terms += rp.rec[i][3] * rp.rec[i][4];
        split = huge_terms * 2 >= total_out_terms;
    }
    if (planned) {
        for (u32 i = 0; i < rp.n; ++i) {
            const u64 pb = rp.rec[i][0], l0 = rp.rec[i][1], r0 = rp.rec[i][2], t1 = rp.rec[i][3], t2 = rp.rec[i][4],
                      o0 = rp.rec[i][5];
            if (!split || pb >= batch || t1 * t2 * dL * 8u < (24ull << 20) || o0 + t1 * t2 > total_out_terms || o0 * U < cursor)
                continue;
            if (o0 * U > cursor) {
                const hipError_t e = flat_range(cursor, o0 * U, pb > cursor_pair ? pb - cursor_pair : 1);
                if (e != hipSuccess)
                    return e;
            }
            const hipError_t e = mul_uniform(n_bits, 1, t1, t2, L + l0 * dL, R + r0 * dL, out + o0 * dL, 0, s);
            if (e != hipSuccess)
                return e;
            cursor = (o0 + t1 * t2) * U;
            cursor_pair = pb + 1;
        }
    }
    hipError_t result = hipSuccess;
    if (cursor < total_units)
        result = flat_range(cursor, total_units, batch > cursor_pair ? batch - cursor_pair : 1);
    return result;
}


// ------------------------------------------------------------------ csgn_mul_ragged_async
// d_plan = [gate: real output terms for the CSR kernel | does not fit | all pairs 1x1][mul_ragged_plan's block]
constexpr u64 kGateWords = 8;
u64 mul_ragged_async_plan_words(u64 batch) { return kGateWords + mul_ragged_plan_scratch_words(batch); }

// the 1x1 stream of k_and_stream, its length read from the gate (0: some other kernel has the batch)
template <typename Unit>
__global__ void __launch_bounds__(256) k_and_stream_gated(const Unit *__restrict__ a, const Unit *__restrict__ b,
                                                          Unit *__restrict__ o, u32 U, const u64 *__restrict__ gate,
                                                          const u64 *__restrict__ offL, const u64 *__restrict__ offR)
{
    const u64 n_units = gate[2] * U;
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n_units) {
        // the operands start where the offset arrays say (a sub-range of a larger CSR: ADVICE r4)
        a += offL[0] * U;
        b += offR[0] * U;
        unit_store<Unit, true>(o + i, a[i] & b[i]);
    }
}

hipError_t mul_ragged_async(u64 n_bits, u64 batch, const u64 *L, const u64 *offL, const u64 *R, const u64 *offR,
                            u64 *out, u64 *offOut, u64 capacity_terms, u64 *d_plan, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0)
        return hipSuccess;
    if (batch >= (1ull << 32))
        return hipErrorInvalidValue;
    u64 *gate = d_plan, *work = d_plan + kGateWords;
    // a batch of 1x1 pairs (fresh ciphertexts handed over as CSR): the stream kernel, if the gate says so
    const bool wide = (dL % 2 == 0) && aligned16(L) && aligned16(R) && aligned16(out);
    const u32 U = (u32)(wide ? dL / 2 : dL);
    const u64 stream_blocks = (batch * U + 255) / 256;
    const bool can_stream = capacity_terms >= batch && stream_blocks <= kMaxBlocks256;
    // the plan kernels; the last of them also writes the gate (real output terms / does not fit / all pairs 1x1)
    hipError_t e = mul_ragged_plan(batch, offL, offR, offOut, work, s, gate, capacity_terms, can_stream);
    if (e != hipSuccess)
        return e;
    if (capacity_terms == 0)
        return hipGetLastError();
    // Round 5, tried for the skewed batch (one 1024 x 1024 pair + 65 535 singles, 70 us end to end) and NOT kept: the stream
    // folded into the CSR kernel (a launch saved: 116.9 against 111.3 us on a million 1 x 1 pairs -- k_and_stream's one
    // unit per lane and one short-lived workgroup per 4 KiB is what that stream wants); an operand touch of the huge
    // pairs from inside the plan kernel, and CSR workgroups that start from the plan's huge-pair records instead of the
    // 64-ary search (no change, 71.7 / 71.9 us: the batch's 21 MB of operands stay in the memory-side cache anyway and the
    // search was not what the time went to).  What it went to was the TAIL: the 640 latency-bound workgroups of the
    // singles, 6 % of the output, all on the one XCD that owned the last contiguous eighth of the launch --
    // xcd_grouped_block (csgn_device.h), 70 -> 54 us.
    if (can_stream) {
        const u64 blocks = stream_blocks;
        {
            if (wide)
                k_and_stream_gated<unit16><<<(u32)blocks, 256, 0, s>>>(reinterpret_cast<const unit16 *>(L),
                                                                        reinterpret_cast<const unit16 *>(R),
                                                                        reinterpret_cast<unit16 *>(out), U, gate, offL, offR);
            else
                k_and_stream_gated<unit8><<<(u32)blocks, 256, 0, s>>>(L, R, out, U, gate, offL, offR);
        }
    }
    // everything else: the CSR kernel over the caller's bound, stopping at the real end
    return mul_ragged(n_bits, batch, L, offL, R, offR, out, offOut, 1, 1, capacity_terms, s, nullptr, 0, gate, work + 4);
}

} // namespace csgn
