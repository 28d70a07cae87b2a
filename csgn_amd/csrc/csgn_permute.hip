// csgn_permute.hip -- Ciphertext::applyPermutation: bit-plane kernel (in-wave 64x64 bit transposes) and ballot bit-gather.
// Hand-written CDNA4 (gfx950) HIP; shared helpers in csgn_device.h, design notes in DESIGN.md.
#include "csgn_device.h"

namespace csgn {

namespace {

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

} // namespace

// ------------------------------------------------------------------------------ public

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
        if (terms_in != 0 && out_terms >= 16 && dL <= 64 && lds <= 160 * 1024 && !tune(TUNE_PERM_BALLOT)) {
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
                              !tune(TUNE_PERM_NARROW);
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
        static thread_local size_t asked_lds = 0;     /* the query is remembered per kernel and LDS size */ \
        static thread_local int asked_per_cu = 0;                                                   \
        if (asked_lds != lds) {                                                                     \
            int q = 0;                                                                              \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, k_permute_planes<LQ, PIPE, UW>, 64, \
                                                             lds) != hipSuccess || q < 1)           \
                q = 1;                                                                              \
            asked_per_cu = q;                                                                       \
            asked_lds = lds;                                                                        \
        }                                                                                           \
        int per_cu = asked_per_cu;                                                                  \
        per_cu = std::max(1, std::min(per_cu, (int)(160 * 1024 / ((lds + 1023) / 1024 * 1024))));   \
        if (const int cap = tune(TUNE_PERM_WAVES))                                          \
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

} // namespace csgn
