// csgn_add.hip -- add = term-list concatenation, uniform and ragged.
// Hand-written CDNA4 (gfx950) HIP; shared helpers in csgn_device.h, design notes in DESIGN.md.
#include "csgn_device.h"

#include <algorithm>

namespace csgn {

namespace {

// Ragged add (CSR offsets): out_b = L_b || R_b (src/Ciphertext.cpp:107-122, no XOR, no de-duplication).
// Same skeleton as the ragged multiply (csgn_mul.hip, k_mul_ragged_flat): the grid covers the flattened
// output, a workgroup owns C consecutive 4 KiB chunks and takes them M at a time; its first pair comes
// from a 64-ary wave search; per turn it bets on the pair it was in (scalar offset loads) and, when that
// pair does not own the whole turn, stages the offsets of the next 256 pairs in LDS with one coalesced
// load per thread and array, where every lane finds its pair by an LDS binary search.  offOut = offL +
// offR, so two windows suffice.
constexpr u32 kAddWin = 256;

template <typename Unit, int C, int M>
__global__ void __launch_bounds__(256) k_add_ragged_flat(const Unit *__restrict__ L,
                                                         const u64 *__restrict__ offL,
                                                         const Unit *__restrict__ R,
                                                         const u64 *__restrict__ offR,
                                                         Unit *__restrict__ out,
                                                         const u64 *__restrict__ offOut, u32 batch,
                                                         u64 unit_base, u64 total_units, u32 U, FastDiv dU,
                                                         u32 device_end, u32 xcd_group)
{
    static_assert(C % M == 0, "chunks per workgroup must be a multiple of the chunks per turn");
    // operands whose sizes only the device knows (a circuit value behind a compaction): the launch was
    // sized for their static bound, the real end of the output is in the offsets
    if (device_end)
        total_units = min(total_units, (offL[batch] + offR[batch]) * U);
    __shared__ u64 w_l[kAddWin + 2], w_r[kAddWin + 2];
    __shared__ u32 s_next;
    const u32 bid = xcd_grouped_block(blockIdx.x, gridDim.x, xcd_group);      // (groups: see csgn_device.h)
    const u64 g_begin = unit_base + (u64)bid * (256u * C);
    if (g_begin >= total_units)
        return;
    const u64 term0 = g_begin / U;
    const u32 r0blk = (u32)(g_begin - term0 * U);
    u32 pw = wave_find(offOut, 0u, batch, term0);               // the same answer in every wave
#pragma unroll 1
    for (int c0 = 0; c0 < C; c0 += M) {
        if (g_begin + (u32)c0 * 256u >= total_units)
            break;
        const u32 pw2 = min(pw + 2u, batch);
        const u64 s_l0 = offL[pw], s_l1 = offL[pw + 1], s_l2 = offL[pw2];
        const u64 s_r0 = offR[pw], s_r1 = offR[pw + 1], s_r2 = offR[pw2];
        const u64 s_o1 = s_l1 + s_r1, s_o2 = s_l2 + s_r2;       // offOut[pw + 1], offOut[pw + 2]
        const u64 turn_end = min(g_begin + (u64)(c0 + M) * 256u, total_units);
        const u64 last_term = term0 + csgn_fastdiv(r0blk + (u32)(turn_end - g_begin) - 1u, dU);
        const bool whole = last_term < s_o1;                    // workgroup-uniform
        const bool two = !whole && last_term < s_o2;            // the turn crosses ONE pair boundary: no window
        if (!whole && !two) {
            const u32 i = threadIdx.x;
            const u32 pi = min(pw + i, batch);
            w_l[i] = offL[pi];
            w_r[i] = offR[pi];
            if (i < 2u) {
                const u32 pe = min(pw + kAddWin + i, batch);
                w_l[kAddWin + i] = offL[pe];
                w_r[kAddWin + i] = offR[pe];
            }
            __syncthreads();
        }
        // A window whose 256 pairs all have the shape of its first one (the common ragged batch: equal ciphertexts handed
        // over as CSR -- a million 1 + 1 sums): a lane's pair is a DIVISION away, not the nine dependent LDS reads of the
        // binary search.  (Past the end of the batch the offsets repeat: the last window searches.)
        u32 uni_t = 0;                                          // terms of a pair's sum if the window is uniform, else 0
        if (!whole && !two) {
            const u32 i = threadIdx.x;
            const u64 dl0 = w_l[1] - w_l[0], dr0 = w_r[1] - w_r[0];
            const bool same = w_l[i + 1] - w_l[i] == dl0 && w_r[i + 1] - w_r[i] == dr0;
            if (__syncthreads_and(same ? 1 : 0) && dl0 + dr0 < (1ull << 20))
                uni_t = (u32)(dl0 + dr0);                       // (256 of them stay under 2^32; 0 = runs of empty pairs: search)
        }
        u32 p[M];
        u64 src[M];                                             // unit index into L (from_l) or R
        bool from_l[M], live[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int c = c0 + m;
            const u64 g = g_begin + (u32)c * 256u + threadIdx.x;
            live[m] = g < total_units;
            // lanes past the end (last workgroup only) redo the LAST unit: their loads stay unconditional
            // and in range whichever of L and R is empty; they store nothing
            const u32 back = live[m] ? 0u : (u32)(g - (total_units - 1u));
            const u32 r = r0blk + (u32)c * 256u + threadIdx.x - back;
            const u32 dt = csgn_fastdiv(r, dU);
            const u64 term = term0 + dt;
            const u32 k = r - dt * U;
            p[m] = pw;
            src[m] = 0;
            from_l[m] = true;
            {
                u64 l0 = s_l0, rr0 = s_r0, t1 = s_l1 - s_l0;
                if (two && term >= s_o1) {                      // the second pair of the bet
                    p[m] = pw + 1u;
                    l0 = s_l1;
                    rr0 = s_r1;
                    t1 = s_l2 - s_l1;
                } else if (!whole && term >= s_o1) {
                    // largest j in [0, kAddWin] with w_l[j] + w_r[j] <= term
                    u32 lo = 0, hi = kAddWin + 1u;
                    if (uni_t) {                                // (workgroup-uniform)
                        const u64 ahead = term - (w_l[0] + w_r[0]);
                        lo = ahead >= (u64)uni_t * kAddWin ? kAddWin : (u32)ahead / uni_t;
                    } else {
#pragma unroll
                        for (int step = 0; step < 9; ++step) {
                            const u32 mid = (lo + hi) >> 1;
                            const bool le = w_l[mid] + w_r[mid] <= term;
                            lo = le ? mid : lo;
                            hi = le ? hi : mid;
                        }
                    }
                    if (lo == kAddWin && pw + kAddWin < batch) { // beyond the window (long runs of empty pairs)
                        p[m] = csr_gallop(offOut, pw + kAddWin, batch, term);
                        l0 = offL[p[m]];
                        rr0 = offR[p[m]];
                        t1 = offL[p[m] + 1] - l0;
                    } else {
                        p[m] = pw + lo;
                        l0 = w_l[lo];
                        rr0 = w_r[lo];
                        t1 = w_l[lo + 1] - l0;
                    }
                }
                const u64 q = term - (l0 + rr0);                // offOut[p] = l0 + rr0
                from_l[m] = q < t1;
                src[m] = from_l[m] ? (l0 + q) * U + k : (rr0 + (q - t1)) * U + k;
            }
        }
        // ONE unconditional load per unit from a selected base: `from_l ? L[i] : R[i]` compiles to a branch around each
        // of two loads with a full vmcnt(0) wait per element -- the M loads of a lane one after the other
        Unit v[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const Unit *base = from_l[m] ? L : R;
            v[m] = base[src[m]];
        }
        // (all M loads are issued HERE: left to itself the compiler sinks a load into the `if (live)` of its store,
        // behind the other stores, with a vmcnt(0) of its own)
#pragma unroll
        for (int m = 0; m < M; ++m)
            asm volatile("" : "+v"(v[m]));
#pragma unroll
        for (int m = 0; m < M; ++m)
            if (live[m])
                unit_store<Unit, true>(out + g_begin + (u32)(c0 + m) * 256u + threadIdx.x, v[m]);
        if (two) {
            pw += 1u;                                           // the turn ended in the second pair
        } else if (!whole) {
            if (threadIdx.x == 255u)
                s_next = p[M - 1];
            __syncthreads();
            pw = s_next;
        }
    }
}

// ---------------------------------------------------------------------------------------
// add = concatenation (src/Ciphertext.cpp:107-122).  Flat map over output units.
// ---------------------------------------------------------------------------------------
// PITCH (circuit placement, csgn_circuit.hip): element p's LU + RU units go to out + p * opitch -- a slice of a larger
// sum; with LU or RU zero this is the strided copy of ONE operand into its slice (the other was written there by its
// producer).
template <typename Unit, bool NT, bool PITCH = false>
__global__ void __launch_bounds__(256) k_add_flat(const Unit *__restrict__ L,
                                                  const Unit *__restrict__ R,
                                                  Unit *__restrict__ out, u32 total_units, u32 LU,
                                                  u32 RU, FastDiv dOU, u32 xcd, u32 opitch = 0)
{
    // one unit per lane, < 2^32 units per launch (see k_and_stream for why)
    const u32 OU = LU + RU;
    const u32 bid = xcd ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x;
    const u32 g = bid * 256u + threadIdx.x;
    if (g < total_units) {
        const u32 pair = csgn_fastdiv(g, dOU);
        const u32 r = g - pair * OU;
        const Unit v = (r < LU) ? L[(u64)pair * LU + r] : R[(u64)pair * RU + (r - LU)];
        unit_store<Unit, NT>(out + (PITCH ? (u64)pair * opitch + r : (u64)g), v);
    }
}

// A LIST of strided copies in one launch (circuit prologue, csgn_circuit.hip): entry e moves `batch` elements of
// elem_words words from src + i * src_pitch to dst + i * dst_pitch.  A workgroup takes 256 consecutive units (16 bytes when
// everything of the entry is 16-byte aligned, else 8) of one entry; first[e] = the entry's first workgroup.
__global__ void __launch_bounds__(256) k_copy_list(const CopyEntry *__restrict__ entries, const u32 *__restrict__ first, u32 n_entries)
{
    u32 e = 0;
    while (e + 1u < n_entries && first[e + 1u] <= blockIdx.x)      // workgroup-uniform: scalar loads, a few dozen entries at most
        ++e;
    const CopyEntry en = entries[e];
    const u32 local = (blockIdx.x - first[e]) * 256u + threadIdx.x;
    const bool wide = ((reinterpret_cast<uintptr_t>(en.src) | reinterpret_cast<uintptr_t>(en.dst)) & 15u) == 0u &&
                      ((en.elem_words | en.src_pitch | en.dst_pitch) & 1u) == 0u;
    if (wide) {
        const u32 eu = en.elem_words / 2u;
        if (local < en.batch * eu) {
            const u32 i = local / eu, k = local - i * eu;
            const unit16 v = reinterpret_cast<const unit16 *>(en.src)[(u64)i * (en.src_pitch / 2u) + k];
            unit_store<unit16, true>(reinterpret_cast<unit16 *>(en.dst) + (u64)i * (en.dst_pitch / 2u) + k, v);
        }
    } else if (local < en.batch * en.elem_words) {
        const u32 i = local / en.elem_words, k = local - i * en.elem_words;
        en.dst[(u64)i * en.dst_pitch + k] = en.src[(u64)i * en.src_pitch + k];
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
                       hipStream_t s, u64 out_pitch_words)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0 || t1 + t2 == 0)
        return hipSuccess;
    if (out_pitch_words && (out_pitch_words < (t1 + t2) * dL || out_pitch_words >= (1ull << 32)))
        return hipErrorInvalidValue;
    const bool wide = (dL % 2 == 0) && aligned16(L) && aligned16(R) && aligned16(out) && out_pitch_words % 2 == 0;
    const u32 U = (u32)(wide ? dL / 2 : dL);
    const u32 opitch = (u32)(wide ? out_pitch_words / 2 : out_pitch_words);
    const u64 OU = (t1 + t2) * U;
    const u64 pairs_per = (0xFFFFFF00ull / OU) ? (0xFFFFFF00ull / OU) : 1;       // units (= threads) per launch < 2^32
    const FastDiv d = csgn_fastdiv_make((u32)OU);
    const u32 sxcd = stream_xcd(batch * OU);
    for (u64 p0 = 0; p0 < batch; p0 += pairs_per) {
        const u64 np = (batch - p0 < pairs_per) ? batch - p0 : pairs_per;
        const u32 tot = (u32)(np * OU);
        const u32 blocks = ceil_div_u64(tot, 256u);
        if (opitch && wide)
            k_add_flat<unit16, true, true><<<blocks, 256, 0, s>>>(
                reinterpret_cast<const unit16 *>(L) + p0 * t1 * U,
                reinterpret_cast<const unit16 *>(R) + p0 * t2 * U,
                reinterpret_cast<unit16 *>(out) + p0 * opitch, tot, (u32)(t1 * U), (u32)(t2 * U), d, sxcd, opitch);
        else if (opitch)
            k_add_flat<unit8, true, true><<<blocks, 256, 0, s>>>(L + p0 * t1 * U, R + p0 * t2 * U,
                                                                 out + p0 * opitch, tot, (u32)(t1 * U),
                                                                 (u32)(t2 * U), d, sxcd, opitch);
        else if (wide)
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

u32 copy_list_blocks(const CopyEntry &e)
{
    // (a launcher-side bound that holds for both unit widths: 8-byte units need twice the workgroups of 16-byte ones)
    return ceil_div_u64((u64)e.batch * e.elem_words, 256u);
}

hipError_t copy_list(const CopyEntry *d_entries, const u32 *d_first, u32 n_entries, u32 total_blocks, hipStream_t s)
{
    if (n_entries == 0 || total_blocks == 0)
        return hipSuccess;
    k_copy_list<<<total_blocks, 256, 0, s>>>(d_entries, d_first, n_entries);
    return hipGetLastError();
}

hipError_t add_ragged(u64 n_bits, u64 batch, const u64 *L, const u64 *offL, const u64 *R,
                      const u64 *offR, u64 *out, u64 *offOut, u64 total_terms_out, hipStream_t s, bool device_end,
                      u64 max_t1, u64 max_t2)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch >= (1ull << 32))
        return hipErrorInvalidValue;
    k_off_sum<<<ceil_div_u64(batch + 1, 256), 256, 0, s>>>(batch + 1, offL, offR, offOut);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || batch == 0 || total_terms_out == 0)
        return e;
    // the caller's bounds met with equality: every pair is max_t1 + max_t2 terms and the CSR arrays describe a uniform
    // batch -- no lane has to find its pair (a million 1+1 sums: 4.5 TB/s through the CSR kernel, 6 through this one)
    if (!device_end && (max_t1 | max_t2) != 0 && batch * (max_t1 + max_t2) == total_terms_out)
        return add_uniform(n_bits, batch, max_t1, max_t2, L, R, out, s);
    const bool wide = (dL % 2 == 0) && aligned16(L) && aligned16(R) && aligned16(out);
    const u32 U = (u32)(wide ? dL / 2 : dL);
    const u64 total_units = total_terms_out * U;
    const FastDiv dU = csgn_fastdiv_make(U);
    // chunks per workgroup as the ragged multiply (ragged_chunks: up to 8 while the grid keeps >= 8192
    // workgroups); M = min(4, C) chunks per turn
    const int chunks = ragged_chunks(total_units);
    const int turn = csgn::tune(TUNE_RAGGED_M);
    const u32 xcd_group = (u32)std::max(0, csgn::tune(TUNE_RAGGED_XCD_GROUP));
    const u64 per_launch = kMaxBlocks256 * 256u;         // units: a multiple of every 256*C
    for (u64 u0 = 0; u0 < total_units; u0 += per_launch) {
        const u64 nu = (total_units - u0 < per_launch) ? total_units - u0 : per_launch;
        const u32 blocks = ceil_div_u64(nu, 256u * (u32)chunks);
#define CSGN_RAGGED_ADD(CH, MM)                                                                     \
    do {                                                                                            \
        if (wide)                                                                                   \
            k_add_ragged_flat<unit16, CH, MM><<<blocks, 256, 0, s>>>(                              \
                reinterpret_cast<const unit16 *>(L), offL, reinterpret_cast<const unit16 *>(R), offR, \
                reinterpret_cast<unit16 *>(out), offOut, (u32)batch, u0, u0 + nu, U, dU,            \
                device_end ? 1u : 0u, xcd_group);                                                   \
        else                                                                                        \
            k_add_ragged_flat<unit8, CH, MM><<<blocks, 256, 0, s>>>(L, offL, R, offR, out, offOut, \
                                                                    (u32)batch, u0, u0 + nu, U, dU, \
                                                                    device_end ? 1u : 0u, xcd_group); \
    } while (0)
#define CSGN_RAGGED_LAUNCH(CH)                                   \
    do {                                                         \
        if (turn >= 4 && (CH) % 4 == 0)                          \
            CSGN_RAGGED_ADD(CH, ((CH) % 4 == 0 ? 4 : 1));        \
        else if (turn >= 2 && (CH) % 2 == 0)                     \
            CSGN_RAGGED_ADD(CH, ((CH) % 2 == 0 ? 2 : 1));        \
        else                                                     \
            CSGN_RAGGED_ADD(CH, 1);                              \
    } while (0)
        switch (chunks) {
        case 1: CSGN_RAGGED_LAUNCH(1); break;
        case 2: CSGN_RAGGED_LAUNCH(2); break;
        case 4: CSGN_RAGGED_LAUNCH(4); break;
        case 16: CSGN_RAGGED_LAUNCH(16); break;
        default: CSGN_RAGGED_LAUNCH(8); break;
        }
#undef CSGN_RAGGED_LAUNCH
#undef CSGN_RAGGED_ADD
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess)
            return le;
    }
    return hipSuccess;
}

} // namespace csgn
