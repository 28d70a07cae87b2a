// csgn_permute.hip -- Ciphertext::applyPermutation: bit-plane kernel (in-wave 64x64 bit transposes) and ballot bit-gather.
// Hand-written CDNA4 (gfx950) HIP; shared helpers in csgn_device.h, design notes in DESIGN.md.
#include "csgn_device.h"

#include <algorithm>
#include <type_traits>

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
// (6 exchange stages: v_permlane32_swap; ds_swizzle + v_perm for the half-word and DPP + v_perm for the
// byte stage; lane exchange + v_alignbit + v_bfi for the three inside a byte) leaves "bit j of 64 terms"
// in one 64-bit word.  Moving bit j to bit j' is then a plain 8-byte LDS move, and a second
// transpose turns the planes back into terms.
// ~1.2 lane-operations per bit instead of ~6 for the ballot form above.
// ---------------------------------------------------------------------------------------
template <int K>
__device__ inline u32 lane_xor(u32 v)
{
    // on the VALU (DPP) where one DPP move does it
    if (K == 8)
        return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x128 /* row_ror:8 */, 0xF, 0xF, true);
    if (K == 4)   // no DPP pattern gives lane^4 in one go (two masked row shifts + a zeroed destination = 3 VALU
                  // slots): the LDS crossbar does it in one ds_swizzle, and with 5 waves per SIMD its round trip is
                  // covered (round 3, profiles/r03/ab_permute_swizzle.log: +5-7 % at N=1247; for the stages that
                  // cost one DPP the swizzle is no gain, for all four a loss)
        return (u32)__builtin_amdgcn_ds_swizzle((int)v, (4 << 10) | 0x1F);
    if (K == 2)
        return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, true);
    return (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true);
}

// per-lane constants of the four in-word stages (K = 8, 4, 2, 1)
struct TrLane {
    u32 rot[4];     // rotate-right amount that lines the partner's half up with mine
    u32 keep[4];    // bits of my own word that stay
    u32 sel8;       // v_perm_b32 selector of the byte stage
    u32 sel16;      // ... of the half-word stage when it goes through the LDS crossbar
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
    c.sel16 = (lane & 16u) ? 0x03020706u : 0x05040100u;
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
//   16: between 16-bit halves and 16-lane rows: the partner's register comes over the LDS crossbar
//       (ds_swizzle xor 16) and one v_perm_b32 keeps my half and takes the partner's -- 1 VALU + 1 DS
//       per register where gather / v_permlane16_swap / scatter took 2.5 VALU (round 3: +2-4 % at
//       N=1247 and 4096, +8 % at N=1300, profiles/r03/ab_permute_swizzle16.log).
template <int Q>
__device__ inline void wave_transpose64_n(u32 (&h)[2 * Q], const TrLane &c)
{
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        auto s32 = __builtin_amdgcn_permlane32_swap(h[2 * q], h[2 * q + 1], false, false);
        h[2 * q] = s32[0];
        h[2 * q + 1] = s32[1];
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        u32 y[2 * Q];
#pragma unroll
        for (int r = 0; r < 2 * Q; ++r)
            y[r] = (u32)__builtin_amdgcn_ds_swizzle((int)h[r], (16 << 10) | 0x1F);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 2 * Q; ++r)
            h[r] = __builtin_amdgcn_perm(y[r], h[r], c.sel16);
        __builtin_amdgcn_sched_barrier(0);
    }
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

// ---------------------------------------------------------------------------------------
// Bit-plane permutation kernel.  ONE LDS array per 64-term group, W waves per group, the source table
// in registers, coalesced HBM access.  (Two earlier forms -- rows, planes and table all in LDS with one
// wave per group; and no row staging at all -- measured slower, 3.6 / 3.1-3.9 TB/s against 4.9-5.0, and
// were removed in round 3; DESIGN 4.6 keeps the numbers, the history keeps the code.)  The one LDS array
// is used in turn as term rows (coalesced 16-byte units in, one word per lane out), as bit planes, and
// as term rows again for the way back; what has to survive a change of role waits in registers.
//   A  prefetched units -> rows          B  rows -> my words (lane = term)
//   C  words -> planes (transposes)          [next group's loads are issued here]
//   D  planes -> my new words (gather through the source table + transposes), kept in registers
//   E  new words -> rows                 F  rows -> HBM, coalesced
// Six workgroup barriers per turn (wave-level when W = 1); LDS per group = max(planes, rows) =
// 10.5 KB at N=1247, 33 KB at N=4096.
// ---------------------------------------------------------------------------------------
template <int MAXI, int CI, int UW, int MAXT>
__global__ void __launch_bounds__(MAXT, 1024 / MAXT) k_permute_planes3(u32 n_bits, u32 dL, u64 out_terms,
                                                                     u64 in_stride_words,
                                                                     const u64 *__restrict__ terms,
                                                                     const u32 *__restrict__ perm,
                                                                     u64 *__restrict__ out)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    u64 *buf = reinterpret_cast<u64 *>(smem_raw);            // planes: buf[j], j <= none;  rows: buf[t*SA + k]
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6, T = blockDim.x, W = T >> 6;
    const u32 NI = dL / UW;                                  // items (UW words) per term
    const u32 nu = 64u * NI;                                 // staging units of a full group
    const u32 SA = dL | 1u;                                  // odd row stride: conflict-free column access
    const u32 none = dL * 64u;
    const TrLane trc = tr_lane(lane);
    typedef u64 StageUnit __attribute__((ext_vector_type(UW)));
    const u64 groups = (out_terms + 63) / 64;

    // source planes of the output bits this lane assembles (item i = wave + k*W), two per register
    constexpr int NS = (MAXI * UW + 1) / 2;
    u32 srcpk[NS];
#pragma unroll
    for (int e = 0; e < NS; ++e)
        srcpk[e] = 0;
#pragma unroll
    for (int k = 0; k < MAXI; ++k)
#pragma unroll
        for (int u = 0; u < UW; ++u) {
            const u32 i = wave + (u32)k * W;
            const u32 j = (i * UW + (u32)u) * 64u + 63u - lane;
            // an unconditional load of a clamped entry: `if (in range) p = perm[j]` compiles to a branch, a load and a full
            // wait per entry -- MAXI * UW dependent round trips (10 us and more) in front of every workgroup's first group
            const u32 pv = perm[min(j, n_bits - 1u)];
            const u32 p = (i < NI && j < n_bits && pv < n_bits) ? pv : none;
            const int e = k * UW + u;
            srcpk[e / 2] |= p << (16 * (e & 1));
        }
    // staging slots of this thread: unit q*T + tid of the group -> (term, word offset)
    u32 rowoff[MAXI], tq[MAXI], inoff[MAXI];
#pragma unroll
    for (int q = 0; q < MAXI; ++q) {
        const u32 u = min((u32)q * T + tid, nu - 1u);
        const u32 t = u / NI;
        tq[q] = t;
        rowoff[q] = t * SA + (u - t * NI) * UW;
        inoff[q] = (u - t * NI) * UW;
    }
    StageUnit uu[MAXI];
    auto fetch = [&](u64 g) {
        const u64 t0 = g * 64;
        const u32 nt = (u32)min((u64)64, out_terms - t0);
        const u64 *base = terms + t0 * in_stride_words;
#pragma unroll
        for (int q = 0; q < MAXI; ++q)                       // terms past the end re-read the last one
            uu[q] = *reinterpret_cast<const StageUnit *>(base + (u64)min(tq[q], nt - 1u) * in_stride_words + inoff[q]);
    };

    u64 g = blockIdx.x;
    if (g < groups)
        fetch(g);
    for (; g < groups; g += gridDim.x) {
        const u64 t0 = g * 64;
        const u32 nt = (u32)min((u64)64, out_terms - t0);
        // A. units -> rows
#pragma unroll
        for (int q = 0; q < MAXI; ++q)
            if ((u32)q * T + tid < nu) {
#pragma unroll
                for (int u = 0; u < UW; ++u)
                    buf[rowoff[q] + u] = uu[q][u];
            }
        __syncthreads();
        // B. my words of term `lane`
        StageUnit w[MAXI];
#pragma unroll
        for (int k = 0; k < MAXI; ++k) {
            const u32 i = min(wave + (u32)k * W, NI - 1u);
#pragma unroll
            for (int u = 0; u < UW; ++u)
                w[k][u] = buf[lane * SA + i * UW + u];
        }
        __syncthreads();
        if (tid == 0)
            buf[none] = 0;                                   // the all-zero plane ("no source")
        // C. words -> planes
        auto to_planes = [&](auto cn_tag, int c0) {
            constexpr int CN = decltype(cn_tag)::value;
            u32 h[2 * CN * UW];
#pragma unroll
            for (int c = 0; c < CN; ++c)
#pragma unroll
                for (int u = 0; u < UW; ++u) {
                    const u64 x = w[c0 + c][u];
                    h[2 * (c * UW + u)] = (u32)x;
                    h[2 * (c * UW + u) + 1] = (u32)(x >> 32);
                }
            wave_transpose64_n<CN * UW>(h, trc);
#pragma unroll
            for (int c = 0; c < CN; ++c) {
                const u32 i = wave + (u32)(c0 + c) * W;
                if (i < NI) {                                // wave-uniform
#pragma unroll
                    for (int u = 0; u < UW; ++u)
                        buf[(i * UW + (u32)u) * 64u + 63u - lane] =
                            ((u64)h[2 * (c * UW + u) + 1] << 32) | h[2 * (c * UW + u)];
                }
            }
        };
        // every wave runs every batch, also one whose items are all past the end for it (the group waits at
        // the barrier for the wave that does have them; unconditional code lets the registers of w die here)
#pragma unroll
        for (int c0 = 0; c0 + CI <= MAXI; c0 += CI)
            to_planes(std::integral_constant<int, CI>(), c0);
        if constexpr (MAXI % CI != 0)
            to_planes(std::integral_constant<int, MAXI % CI>(), MAXI / CI * CI);
        __syncthreads();
        if (g + gridDim.x < groups)
            fetch(g + gridDim.x);                            // travels during D..F
        // D. planes -> my new words (registers)
        auto from_planes = [&](auto cn_tag, int c0) {
            constexpr int CN = decltype(cn_tag)::value;
            u32 h[2 * CN * UW];
#pragma unroll
            for (int c = 0; c < CN; ++c)
#pragma unroll
                for (int u = 0; u < UW; ++u) {
                    const int e = (c0 + c) * UW + u;
                    const u32 pl = (e & 1) ? srcpk[e / 2] >> 16 : srcpk[e / 2] & 0xFFFFu;
                    const u64 y = buf[pl];
                    h[2 * (c * UW + u)] = (u32)y;
                    h[2 * (c * UW + u) + 1] = (u32)(y >> 32);
                }
            wave_transpose64_n<CN * UW>(h, trc);
#pragma unroll
            for (int c = 0; c < CN; ++c)
#pragma unroll
                for (int u = 0; u < UW; ++u)
                    w[c0 + c][u] = ((u64)h[2 * (c * UW + u) + 1] << 32) | h[2 * (c * UW + u)];
        };
#pragma unroll
        for (int c0 = 0; c0 + CI <= MAXI; c0 += CI)
            from_planes(std::integral_constant<int, CI>(), c0);
        if constexpr (MAXI % CI != 0)
            from_planes(std::integral_constant<int, MAXI % CI>(), MAXI / CI * CI);
        __syncthreads();
        // E. new words -> rows
#pragma unroll
        for (int k = 0; k < MAXI; ++k) {
            const u32 i = wave + (u32)k * W;
            if (i < NI) {
#pragma unroll
                for (int u = 0; u < UW; ++u)
                    buf[lane * SA + i * UW + u] = w[k][u];
            }
        }
        __syncthreads();
        // F. rows -> out, coalesced (the group's output is one contiguous run of nt*dL words)
        StageUnit *obase = reinterpret_cast<StageUnit *>(out + t0 * dL);
#pragma unroll
        for (int q = 0; q < MAXI; ++q) {
            const u32 u = (u32)q * T + tid;
            if (u < nt * NI) {
                StageUnit o;
#pragma unroll
                for (int x = 0; x < UW; ++x)
                    o[x] = buf[rowoff[q] + x];
                obase[u] = o;
            }
        }
        __syncthreads();
    }
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
    // bit-plane kernel: one LDS array per 64-term group, W waves per group (knob perm_ballot forces the
    // ballot form below, which also takes batches under 16 terms and dL > 256)
    if (terms_in != 0 && out_terms >= 16 && dL <= 256 && !tune(TUNE_PERM_BALLOT)) {
        const size_t lds = std::max<size_t>(((size_t)dL * 64 + 1) * 8, (size_t)(64 * (dL | 1) * 8));
        int cus = 256;
        {
            int dev = 0;
            if (hipGetDevice(&dev) == hipSuccess)
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        }
        const bool wide = dL % 2 == 0 && stride % 2 == 0 && (((uintptr_t)terms | (uintptr_t)out) & 15) == 0 &&
                          !tune(TUNE_PERM_NARROW);
        const u32 UW = wide ? 2u : 1u;
        const u32 NI = (u32)dL / UW;
        // LDS comes in 1 KiB granules; aim at ~16 waves per CU
        const u32 groups_per_cu = std::max<u32>(1u, (u32)((160u * 1024u) / ((lds + 1023) / 1024 * 1024)));
        // Waves per group.  At least enough to put ~16 waves on the CU and to keep a wave's items
        // within its register budget (4 items = 112 VGPRs); among the
        // admissible counts the one that wastes the fewest transposes -- every wave runs whole
        // batches of CI items and the group waits for its slowest wave, so the cost of a choice is
        // W * roundup(ceil(NI/W), CI) transposed items for NI useful ones.  N=1247 (10 items): 5 waves
        // x 2 items, measured 5.0 TB/s against 4.5 for 3 waves x 4/3/3 (profiles/r02/ab_permute.log).
        auto bucket = [&](u32 per_wave, u32 *ci) -> u32 {
            const u32 mx = per_wave <= 2 ? 2u : per_wave <= 4 ? 4u : per_wave <= 5 ? 5u : per_wave <= 8 ? 8u : 10u;
            *ci = wide ? 2u : (mx == 2 ? 2u : (mx == 5 || mx == 10) ? 5u : 4u);
            return mx;
        };
        u32 w_min = std::max<u32>(1u, (16u + groups_per_cu - 1) / groups_per_cu);
        w_min = std::max<u32>(w_min, (NI + 3u) / 4u);
        w_min = std::min<u32>(w_min, std::min<u32>(16u, NI));
        u32 W = w_min;
        {
            u64 best = ~0ull;
            for (u32 cand = w_min; cand <= std::min<u32>(16u, NI); ++cand) {
                const u32 pw = (NI + cand - 1) / cand;
                if (pw > 10)
                    continue;
                u32 ci = 2;
                (void)bucket(pw, &ci);
                const u64 cost = (u64)cand * ((pw + ci - 1) / ci * ci);
                if (cost < best) {
                    best = cost;
                    W = cand;
                }
            }
        }
        if (const int forced = tune(TUNE_PERM_WAVES))
            W = std::min<u32>(std::max<u32>((u32)std::max(1, forced), (NI + 9u) / 10u), std::min<u32>(16u, NI));
        const u32 per_wave = (NI + W - 1) / W;
        const u64 groups = (out_terms + 63) / 64;
        const int persist = tune(TUNE_PERM_PERSIST);
        if (groups * 64u * W >= (1ull << 32) && persist == 0)
            return hipErrorInvalidValue;
        // Persistent workgroups: as many groups per CU as are really resident (registers and LDS: the
        // runtime's answer, remembered per kernel and LDS size), but not more than ~20 waves -- an
        // oversubscribed CU runs the surplus workgroups after the others (measured at N=1247, 5 waves
        // per group: 4 groups per CU 4.9-5.0 TB/s, 6 groups 4.3).  Equal trip counts per workgroup.
        auto grid_for = [&](int occ) -> u32 {
            u64 per_cu = (u64)std::max(1, occ);
            per_cu = std::min<u64>(per_cu, std::max<u32>(1u, (20u + W - 1) / W));
            if (persist > 1)
                per_cu = (u64)persist;
            const u64 resident = persist == 0 ? groups : (u64)cus * per_cu;
            const u64 rounds = (groups + resident - 1) / resident;
            return (u32)((groups + rounds - 1) / rounds);
        };
#define CSGN_PLANES_K(KERNEL, MAXI, CI, UWV, MAXT)                                                        \
    do {                                                                                                  \
        if (lds > 65536) {                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL<MAXI, CI, UWV, MAXT>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
            if (e != hipSuccess)                                                                          \
                return e;                                                                                 \
        }                                                                                                 \
        static thread_local size_t asked_lds = 0;                                                         \
        static thread_local u32 asked_w = 0;                                                              \
        static thread_local int asked_occ = 0;                                                            \
        if (asked_lds != lds || asked_w != W) {                                                           \
            int q = 0;                                                                                    \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, KERNEL<MAXI, CI, UWV, MAXT>, (int)(64u * W), \
                                                             lds) != hipSuccess || q < 1)                 \
                q = 1;                                                                                    \
            asked_occ = q;                                                                                \
            asked_lds = lds;                                                                              \
            asked_w = W;                                                                                  \
        }                                                                                                 \
        KERNEL<MAXI, CI, UWV, MAXT><<<grid_for(asked_occ), 64u * W, lds, s>>>((u32)n_bits, (u32)dL, out_terms, \
                                                                              stride, terms, perm, out);  \
    } while (0)
#define CSGN_PLANES2_T(MAXI, CI, UWV, MAXT) CSGN_PLANES_K(k_permute_planes3, MAXI, CI, UWV, MAXT)
#define CSGN_PLANES2(MAXI, CI, UWV)                 \
    do {                                            \
        if (W == 1)                                 \
            CSGN_PLANES2_T(MAXI, CI, UWV, 64);      \
        else if (W <= 4)                            \
            CSGN_PLANES2_T(MAXI, CI, UWV, 256);     \
        else                                        \
            CSGN_PLANES2_T(MAXI, CI, UWV, 1024);    \
    } while (0)
        if (per_wave > 10) {
            // more than 10 items per wave even with 16 waves (odd dL > 160): the forms below take it
        } else if (wide) {
            if (per_wave <= 2)
                CSGN_PLANES2(2, 2, 2);
            else if (per_wave <= 4)
                CSGN_PLANES2(4, 2, 2);
            else if (per_wave <= 5)
                CSGN_PLANES2(5, 2, 2);
            else if (per_wave <= 8)
                CSGN_PLANES2(8, 2, 2);
            else
                CSGN_PLANES2(10, 2, 2);
        } else {
            if (per_wave <= 2)
                CSGN_PLANES2(2, 2, 1);
            else if (per_wave <= 4)
                CSGN_PLANES2(4, 4, 1);
            else if (per_wave <= 5)
                CSGN_PLANES2(5, 5, 1);
            else if (per_wave <= 8)
                CSGN_PLANES2(8, 4, 1);
            else
                CSGN_PLANES2(10, 5, 1);
        }
#undef CSGN_PLANES2
#undef CSGN_PLANES2_T
#undef CSGN_PLANES_K
        if (per_wave <= 10)
            return hipGetLastError();
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
    else if (dL <= 32 || (dL % 32 == 0 && dL % 64 != 0))
        CSGN_PERMUTE_LAUNCH(32);
    else
        CSGN_PERMUTE_LAUNCH(64);
#undef CSGN_PERMUTE_LAUNCH
    return hipGetLastError();
}

} // namespace csgn
