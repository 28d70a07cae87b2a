// csgn_encrypt.hip -- encrypt: explicit-randomness (parity) and device-RNG (throughput) forms.
// Hand-written CDNA4 (gfx950) HIP; shared helpers in csgn_device.h, design notes in DESIGN.md.
#include "csgn_device.h"

namespace csgn {

namespace {

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

} // namespace

// ------------------------------------------------------------------------------ public

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
        if (k_seg && tune(TUNE_ENC_LDS) == 0) {
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

} // namespace csgn
