// csgn_add.hip -- add = term-list concatenation, uniform and ragged.
// Hand-written CDNA4 (gfx950) HIP; shared helpers in csgn_device.h, design notes in DESIGN.md.
#include "csgn_device.h"

namespace csgn {

namespace {

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
            // bet on the pair the wave was in (as k_mul_ragged_flat): its offsets are wave-uniform scalar
            // loads that arrive in one round trip with the end-of-pair test; lanes beyond it walk on
            const u64 s_l0 = offL[pw], s_l1 = offL[pw + 1], s_r0 = offR[pw], s_r1 = offR[pw + 1];
            u64 l0 = s_l0, rr0 = s_r0, t1 = s_l1 - s_l0;
            if (term >= s_l1 + s_r1) {                          // offOut[pw + 1] = offL[pw + 1] + offR[pw + 1]
                p = csr_gallop(offOut, pw, batch, term);
                l0 = offL[p];
                rr0 = offR[p];
                t1 = offL[p + 1] - l0;
            }
            const u64 q = term - (l0 + rr0);                    // offOut[p] = l0 + rr0
            const Unit v = (q < t1) ? L[(l0 + q) * U + k] : R[(rr0 + (q - t1)) * U + k];
            unit_store<Unit, true>(out + g, v);
        }
        pw = (u32)__builtin_amdgcn_readfirstlane((int)p);
    }
}

// ---------------------------------------------------------------------------------------
// add = concatenation (src/Ciphertext.cpp:107-122).  Flat map over output units.
// ---------------------------------------------------------------------------------------
template <typename Unit, bool NT>
__global__ void __launch_bounds__(256) k_add_flat(const Unit *__restrict__ L,
                                                  const Unit *__restrict__ R,
                                                  Unit *__restrict__ out, u32 total_units, u32 LU,
                                                  u32 RU, FastDiv dOU, u32 xcd)
{
    // one unit per lane, < 2^32 units per launch (see k_and_stream for why)
    const u32 OU = LU + RU;
    const u32 bid = xcd ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const u32 g = bid * 256u + threadIdx.x;
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

} // namespace

// ------------------------------------------------------------------------------ public

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
    const u32 sxcd = stream_xcd(batch * OU);
    for (u64 p0 = 0; p0 < batch; p0 += pairs_per) {
        const u64 np = (batch - p0 < pairs_per) ? batch - p0 : pairs_per;
        const u32 tot = (u32)(np * OU);
        const u32 blocks = ceil_div_u64(tot, 256u);
        if (wide)
            k_add_flat<unit16, true><<<blocks, 256, 0, s>>>(
                reinterpret_cast<const unit16 *>(L) + p0 * t1 * U,
                reinterpret_cast<const unit16 *>(R) + p0 * t2 * U,
                reinterpret_cast<unit16 *>(out) + p0 * OU, tot, (u32)(t1 * U), (u32)(t2 * U), d, sxcd);
        else
            k_add_flat<unit8, true><<<blocks, 256, 0, s>>>(L + p0 * t1 * U, R + p0 * t2 * U,
                                                           out + p0 * OU, tot, (u32)(t1 * U),
                                                           (u32)(t2 * U), d, sxcd);
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

} // namespace csgn
