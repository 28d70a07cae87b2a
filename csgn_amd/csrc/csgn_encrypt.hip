// csgn_encrypt.hip -- encrypt: explicit-randomness (parity) form and keyed device-generator (ChaCha, throughput) form.
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
__global__ void __launch_bounds__(256) k_encrypt(u64 n_bits, u32 dL, u64 D, u64 batch, u32 CB,
                                                 FastDiv ddL, const uint8_t *__restrict__ plain,
                                                 const u64 *__restrict__ rnd,
                                                 const u32 *__restrict__ chosen,
                                                 const uint8_t *__restrict__ last,
                                                 const u64 *__restrict__ mask,
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
        u64 w = rnd[gw];
        if (k == dL - 1)
            w &= tail;
        tile[u] = w;
    }
    for (u32 k = tid; k < dL; k += 256u)
        lmask[k] = mask[k];
    __syncthreads();

    if (tid < nc && !(plain[c0 + tid] & 1u)) {
        const u64 c = c0 + tid;
        const u64 pos = chosen[c];
        const u32 spare = last[c] & 1u;
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

template <typename Unit, int K>
__global__ void __launch_bounds__(256) k_encrypt_seg(u64 n_bits, u32 dL, u32 U, FastDiv dU, u64 D, u64 batch,
                                                     u32 TB, const uint8_t *__restrict__ plain,
                                                     const Unit *__restrict__ rnd,
                                                     const u32 *__restrict__ chosen,
                                                     const uint8_t *__restrict__ last,
                                                     const Unit *__restrict__ mask,
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
                pos = chosen[c];
                sp = last[c] & 1u;
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
        w[j] = unit_to_words(rnd[g]);
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


// ---------------------------------------------------------------------------------------
// Keyed device generator: ChaCha (D. J. Bernstein's block function, 64-bit counter + 64-bit nonce
// layout, ROUNDS = 8, 12 or 20) in counter mode under a 256-bit SECRET key.  The output cannot be
// inverted to the key or to other outputs (the round-1 generator, splitmix64 of seed + index, was a
// bijection of its state: one mask-free ciphertext word gave away the seed and with it the secret
// positions -- ADVICE r1).  Pure 32-bit add / xor / rotate.  Measured on gfx950 (tools/valu_bench.hip,
// profiles/r03/valu_issue.txt): v_add_u32 and v_xor_b32 issue at 2.25 cycles per wave64 instruction and
// SIMD with two or more waves resident, v_alignbit_b32 -- like every shift, permute and 3-operand
// integer op -- at 4.25, and the ChaCha mix of the three at 3.7-3.9: that, not 2, is the issue ceiling
// this kernel is priced against.
//
// Keystream layout (restated independently in oracle/csgn_oracle.c): a ciphertext is U = ceil(dL/2)
// 16-byte units; P = U / gcd(U, 256) and Gc = 256*P/U, so that a GROUP of Gc ciphertexts is exactly
// P*256 units.  Unit j of ciphertext c (a GLOBAL index: shards of a batch see the same stream) is
//     g = c / Gc,  r = (c % Gc)*U + j,  p = r / 256,  q = (r % 256) / 64,  L = r % 64
//     32-bit words 4q..4q+3 of ChaCha(key, nonce, counter = (g*P + p)*64 + L)
// i.e. the keystream is dealt out in 4 KiB tiles, lane L of a wave computes block L of the tile and
// its four quarters are four consecutive coalesced 1 KiB stores.
// ---------------------------------------------------------------------------------------
struct EncKeyed {
    u32 key[8];
    u32 nonce_lo, nonce_hi;
};

__device__ inline u32 rotl32(u32 x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }

#define CSGN_QR(a, b, c, d)     \
    a += b; d ^= a; d = rotl32(d, 16); \
    c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  \
    c += d; b ^= c; b = rotl32(b, 7)

// DRAW selects the constant words of the state: the keystream proper uses ChaCha's own "expand 32-byte k";
// the stream the plaintext-0 rule draws its position from uses "csgn draw pos v1".  Different
// constants make the two streams of one (key, nonce) unrelated functions of the counter, so no value
// of the nonce turns one into the other (round 2 separated them by flipping the nonce's top bit: the
// draw stream of nonce N was the DATA stream of N ^ 2^63 -- ADVICE r2).
template <int ROUNDS, bool DRAW = false>
__device__ inline void chacha_block(const EncKeyed &k, u32 nonce_lo, u32 nonce_hi, u32 ctr_lo, u32 ctr_hi,
                                    u32 (&x)[16])
{
    const u32 c0 = DRAW ? 0x6e677363u : 0x61707865u, c1 = DRAW ? 0x61726420u : 0x3320646eu,
              c2 = DRAW ? 0x6f702077u : 0x79622d32u, c3 = DRAW ? 0x31762073u : 0x6b206574u;
    x[0] = c0; x[1] = c1; x[2] = c2; x[3] = c3;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        x[4 + i] = k.key[i];
    x[12] = ctr_lo; x[13] = ctr_hi; x[14] = nonce_lo; x[15] = nonce_hi;
#pragma unroll
    for (int r = 0; r < ROUNDS; r += 2) {
        CSGN_QR(x[0], x[4], x[8], x[12]);
        CSGN_QR(x[1], x[5], x[9], x[13]);
        CSGN_QR(x[2], x[6], x[10], x[14]);
        CSGN_QR(x[3], x[7], x[11], x[15]);
        CSGN_QR(x[0], x[5], x[10], x[15]);
        CSGN_QR(x[1], x[6], x[11], x[12]);
        CSGN_QR(x[2], x[7], x[8], x[13]);
        CSGN_QR(x[3], x[4], x[9], x[14]);
    }
    x[0] += c0; x[1] += c1; x[2] += c2; x[3] += c3;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        x[4 + i] += k.key[i];
    x[12] += ctr_lo; x[13] += ctr_hi; x[14] += nonce_lo; x[15] += nonce_hi;
}

// TWO blocks at once with the instruction order forced: each of a quarter round's twelve steps is done
// for all eight quarter rounds in flight (four per block) before the next step starts.  Measured
// (tools/valu_bench.hip, profiles/r03/valu_issue.txt): the VALU issues a stream that alternates between
// its full-rate class (add, xor: 2.25 cycles per wave64 instruction) and its half-rate class (rotate:
// 4.25) at 3.7-3.9 cycles per instruction when the classes change every 1-4 instructions -- which is
// what hipcc's own schedule of one block does -- but at 3.2 in runs of eight: the price of mixing is
// paid per switch.  __builtin_amdgcn_sched_barrier(0) keeps hipcc from interleaving the steps again.
// The two blocks may differ in key, nonce and counter (encrypt: two counters of one stream; fused
// chain: one counter of two streams).  Words identical to chacha_block (same tests).
#define CSGN_ST1(a, b, c, d) a += b;
#define CSGN_ST2(a, b, c, d) d ^= a;
#define CSGN_ST3(a, b, c, d) d = rotl32(d, 16);
#define CSGN_ST4(a, b, c, d) c += d;
#define CSGN_ST5(a, b, c, d) b ^= c;
#define CSGN_ST6(a, b, c, d) b = rotl32(b, 12);
#define CSGN_ST7(a, b, c, d) a += b;
#define CSGN_ST8(a, b, c, d) d ^= a;
#define CSGN_ST9(a, b, c, d) d = rotl32(d, 8);
#define CSGN_ST10(a, b, c, d) c += d;
#define CSGN_ST11(a, b, c, d) b ^= c;
#define CSGN_ST12(a, b, c, d) b = rotl32(b, 7);
#define CSGN_COLQ(ST, X) ST(X[0], X[4], X[8], X[12]) ST(X[1], X[5], X[9], X[13]) ST(X[2], X[6], X[10], X[14]) ST(X[3], X[7], X[11], X[15])
#define CSGN_DIAQ(ST, X) ST(X[0], X[5], X[10], X[15]) ST(X[1], X[6], X[11], X[12]) ST(X[2], X[7], X[8], X[13]) ST(X[3], X[4], X[9], X[14])
#define CSGN_BOTH(Q, ST) Q(ST, xa) Q(ST, xb) __builtin_amdgcn_sched_barrier(0);
#define CSGN_HALF(Q)                                                                                      \
    CSGN_BOTH(Q, CSGN_ST1) CSGN_BOTH(Q, CSGN_ST2) CSGN_BOTH(Q, CSGN_ST3) CSGN_BOTH(Q, CSGN_ST4)            \
    CSGN_BOTH(Q, CSGN_ST5) CSGN_BOTH(Q, CSGN_ST6) CSGN_BOTH(Q, CSGN_ST7) CSGN_BOTH(Q, CSGN_ST8)            \
    CSGN_BOTH(Q, CSGN_ST9) CSGN_BOTH(Q, CSGN_ST10) CSGN_BOTH(Q, CSGN_ST11) CSGN_BOTH(Q, CSGN_ST12)

template <int ROUNDS>
__device__ inline void chacha_block2(const EncKeyed &ka, u32 na_lo, u32 na_hi, u32 ca_lo, u32 ca_hi, const EncKeyed &kb,
                                     u32 nb_lo, u32 nb_hi, u32 cb_lo, u32 cb_hi, u32 (&xa)[16], u32 (&xb)[16])
{
    const u32 c0 = 0x61707865u, c1 = 0x3320646eu, c2 = 0x79622d32u, c3 = 0x6b206574u;   // "expand 32-byte k"
    xa[0] = c0; xa[1] = c1; xa[2] = c2; xa[3] = c3;
    xb[0] = c0; xb[1] = c1; xb[2] = c2; xb[3] = c3;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xa[4 + i] = ka.key[i];
        xb[4 + i] = kb.key[i];
    }
    xa[12] = ca_lo; xa[13] = ca_hi; xa[14] = na_lo; xa[15] = na_hi;
    xb[12] = cb_lo; xb[13] = cb_hi; xb[14] = nb_lo; xb[15] = nb_hi;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < ROUNDS; r += 2) {
        CSGN_HALF(CSGN_COLQ)
        CSGN_HALF(CSGN_DIAQ)
    }
    xa[0] += c0; xa[1] += c1; xa[2] += c2; xa[3] += c3;
    xb[0] += c0; xb[1] += c1; xb[2] += c2; xb[3] += c3;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xa[4 + i] += ka.key[i];
        xb[4 + i] += kb.key[i];
    }
    xa[12] += ca_lo; xa[13] += ca_hi; xa[14] += na_lo; xa[15] += na_hi;
    xb[12] += cb_lo; xb[13] += cb_hi; xb[14] += nb_lo; xb[15] += nb_hi;
}
#undef CSGN_HALF
#undef CSGN_BOTH
#undef CSGN_DIAQ
#undef CSGN_COLQ
#undef CSGN_QR

// The draw of src/SecretKey.cpp:51 for ciphertext c in keyed mode: a separate stream of the same key and
// nonce (constants "csgn draw pos v1"), block counter = c, first output word range-reduced to [0, D).

template <int ROUNDS>
__device__ inline u32 keyed_draw_pos(const EncKeyed &k, u32 nonce_lo, u32 nonce_hi, u64 c,
                                     const u64 *__restrict__ key_idx, u32 D)
{
    u32 x[16];
    chacha_block<ROUNDS, true>(k, nonce_lo, nonce_hi, (u32)c, (u32)(c >> 32), x);
    return (u32)key_idx[__umulhi(x[0], D)];
}

struct EncWaveArgs {
    EncKeyed rng;
    const u64 *epoch;             // optional device word added to the nonce (circuits: bumped per replay)
    const uint8_t *plain;         // batch bytes
    const u64 *key_idx;           // D secret indices
    const unit16 *mask;           // U units
    unit16 *out;                  // batch*U units
    u64 first_ct, batch;          // global index of plain[0] / out[0], ciphertext count
    u64 group0, ngroups;          // first group and number of groups the launch covers
    u32 U, Gc, D, iters;          // iters: groups per wave (every wave of the grid runs the same count)
    u32 n_bits;
    u32 tail_lo, tail_hi;         // the last word's valid-bit mask
    FastDiv dU;
};

// all mask bits set in v?  (~v & m) OR-ed over the four dwords: one v_bitop3_b32 each
__device__ inline bool unit_covers_chain(unit16 v, unit16 m)
{
    u32 t = ~v.x & m.x;
    t |= ~v.y & m.y;
    t |= ~v.z & m.z;
    t |= ~v.w & m.w;
    return t == 0u;
}

// One wave = one group of Gc ciphertexts = P passes of 256 units.  Everything that depends only on
// a unit's place r in the group -- its key-mask unit, the valid-bit mask of its second word, the
// ciphertext it belongs to -- was tabulated in LDS by the workgroup (mtab / ttab / ctab), so the
// per-unit work beside the generator is: two LDS reads, one plaintext byte, the cover test, the OR
// of the key mask, one store.  FULL: every ciphertext of the group is inside [first_ct, first_ct +
// batch) (all but the first and last group of a launch).
// COMPACT: instead of the three per-unit tables (26 bytes per unit: 33 KB for the 1 280 units of an
// N=1247 group, four workgroups per CU) ONE 4-byte entry per unit -- local ciphertext number, byte
// offset of its key-mask unit, "last unit of a ciphertext" flag -- and the U mask units themselves
// (5.3 KB): five more VALU instructions per unit, four times the resident waves.
// plw: the plaintext bits of the group's Gc ciphertexts as bytes 0x00 / 0xFF, put into LDS by the wave before
// the call (0 for ciphertexts outside the launch's range).  NOTHING in the unit loop loads from global memory:
// a load's s_waitcnt vmcnt() also waits for the non-temporal stores issued before it, i.e. for an HBM
// write round trip per unit (round 2's form did exactly that and sat at 67 % of the issue rate).
template <int ROUNDS, int P, bool FULL, bool COMPACT>
__device__ inline void encrypt_group(const EncWaveArgs &a, const unit16 *mtab, const uint2 *ttab,
                                     const unsigned short *ctab, u64 *cover, const signed char *plw, u32 lane,
                                     u64 group, u32 nonce_lo, u32 nonce_hi)
{
    const u32 U = a.U;
    const u64 cbase = group * a.Gc;                        // first ciphertext of the group (global index)
    u32 cl_lo = 0, cl_hi = a.Gc;                           // valid ciphertexts of the group, local indices
    if (!FULL) {
        cl_lo = cbase < a.first_ct ? (u32)(a.first_ct - cbase) : 0u;
        const u64 end = a.first_ct + a.batch;
        cl_hi = cbase + a.Gc > end ? (u32)(end - cbase) : a.Gc;
    }
    // out / plain addressed from the group's own origin (may lie before the buffers for a partial
    // first group; only in-range elements are touched)
    const long long origin = (long long)cbase - (long long)a.first_ct;
    unit16 *outg = a.out + origin * (long long)U;
    const u64 blk0 = group * (u64)P * 64u + lane;
    u32 acc_lo = 0, acc_hi = 0;                            // lane s keeps the cover ballot of slot s = p*4+q
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const u64 ctr = blk0 + (u64)p * 64u;
        u32 x[16];
        chacha_block<ROUNDS>(a.rng, nonce_lo, nonce_hi, (u32)ctr, (u32)(ctr >> 32), x);
        unit16 *const outp = outg + ((u32)p * 256u + lane);   // one address per pass; the four stores differ by an immediate
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 r = (u32)p * 256u + (u32)q * 64u + lane;
            unit16 m;
            uint2 tl;
            u32 cl;
            if (COMPACT) {
                // mtab: the U mask units; ttab (as u32[]): one packed entry per unit of the group
                const u32 e = reinterpret_cast<const u32 *>(ttab)[r];
                cl = e & 0xFFFFu;
                m = *reinterpret_cast<const unit16 *>(reinterpret_cast<const unsigned char *>(mtab) + ((e >> 16) & 0x7FFFu));
                const u32 inner = (u32)((int)e >> 31);                    // ~0 unless this is a ciphertext's last unit
                tl.x = a.tail_lo | inner;
                tl.y = a.tail_hi | inner;
            } else {
                m = mtab[r];
                tl = ttab[r];
                cl = ctab[r];
            }
            const bool inr = FULL || (cl - cl_lo < cl_hi - cl_lo);
            // plaintext 1 -> ~0 (the byte is 0x00 or 0xFF, read sign-extended): OR the key mask in (src/SecretKey.cpp:44-45)
            const u32 pm = (u32)(int)plw[cl];
            unit16 v;
            v.x = x[4 * q];
            v.y = x[4 * q + 1];
            v.z = x[4 * q + 2];
            v.w = x[4 * q + 3];
            // all secret positions of this unit came out 1?  Tested on the words BEFORE the tail mask: the key mask
            // has no bit in the padding, so the padding bits cannot change the verdict
            const u64 b = __ballot(unit_covers_chain(v, m));
            write_lane64(b, p * 4 + q, acc_lo, acc_hi);
            v.x |= m.x & pm;
            v.y |= m.y & pm;
            v.z = (v.z & tl.x) | (m.z & pm);                // padding bits of a ciphertext's last word stay 0
            v.w = (v.w & tl.y) | (m.w & pm);
            if (inr)
                unit_store<unit16, true>(outp + q * 64, v);
        }
    }
    if (lane < (u32)P * 4u)
        cover[lane] = ((u64)acc_hi << 32) | acc_lo;
}

template <int ROUNDS, int P, bool COMPACT>
__global__ void __launch_bounds__(256) k_encrypt_wave(EncWaveArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr u32 R = 256u * P;                                                  // units per group
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const u32 U = a.U;
    // full tables:    [mtab: R mask units][ttab: R tail masks][cover: 4 x P*4 words][ctab: R u16]
    // compact tables: [mtab: U mask units, 16-byte aligned][etab: R u32][cover]
    unit16 *mtab = reinterpret_cast<unit16 *>(smem_raw);
    const size_t m_bytes = COMPACT ? (size_t)U * 16u : (size_t)R * 16u;
    uint2 *ttab = reinterpret_cast<uint2 *>(smem_raw + m_bytes);
    const size_t t_bytes = COMPACT ? (size_t)R * 4u : (size_t)R * 8u;
    u64 *cover_all = reinterpret_cast<u64 *>(smem_raw + ((m_bytes + t_bytes + 7u) & ~(size_t)7u));
    signed char *plw = reinterpret_cast<signed char *>(cover_all + 4u * P * 4u) + wave * 256u;   // Gc <= 256 bytes per wave: 0x00 / 0xFF
    unsigned short *ctab = reinterpret_cast<unsigned short *>(cover_all + 4u * P * 4u + 4u * 32u);
    if (COMPACT) {
        for (u32 k = tid; k < U; k += 256u)
            mtab[k] = a.mask[k];
        u32 *etab = reinterpret_cast<u32 *>(ttab);
        for (u32 r = tid; r < R; r += 256u) {
            const u32 cl = csgn_fastdiv(r, a.dU), j = r - cl * U;
            etab[r] = cl | ((j * 16u) << 16) | (j == U - 1u ? 0u : 0x80000000u);   // bit 31: NOT the last unit
        }
    } else {
        for (u32 r = tid; r < R; r += 256u) {
            const u32 cl = csgn_fastdiv(r, a.dU), j = r - cl * U;
            mtab[r] = a.mask[j];
            ttab[r] = (j == U - 1u) ? make_uint2(a.tail_lo, a.tail_hi) : make_uint2(~0u, ~0u);
            ctab[r] = (unsigned short)cl;
        }
    }
    u64 *cover = cover_all + wave * (P * 4);
    __syncthreads();

    u32 nonce_lo = a.rng.nonce_lo, nonce_hi = a.rng.nonce_hi;
    if (a.epoch) {
        const u64 nn = (((u64)nonce_hi << 32) | nonce_lo) + *a.epoch;
        nonce_lo = (u32)nn;
        nonce_hi = (u32)(nn >> 32);
    }
    // every wave walks `iters` groups, a whole grid apart each time (wave-local state only: no
    // workgroup barrier inside the loop)
    for (u32 it = 0; it < a.iters; ++it) {
        const u64 gi = ((u64)it * gridDim.x + blockIdx.x) * 4u + wave;           // relative to group0
        if (gi >= a.ngroups)
            break;
        const u64 group = a.group0 + gi;
        const u64 cbase = group * a.Gc;
        const bool full = cbase >= a.first_ct && cbase + a.Gc <= a.first_ct + a.batch;
        // the group's plaintext bytes -> LDS (the one global read of the group, before any of its stores)
        for (u32 cl = lane; cl < a.Gc; cl += kWave) {
            const u64 c = cbase + cl;
            const bool in = c >= a.first_ct && c < a.first_ct + a.batch;
            plw[cl] = in ? (signed char)(0 - (int)(a.plain[c - a.first_ct] & 1u)) : (signed char)0;
        }
        __builtin_amdgcn_wave_barrier();
        if (full)
            encrypt_group<ROUNDS, P, true, COMPACT>(a, mtab, ttab, ctab, cover, plw, lane, group, nonce_lo, nonce_hi);
        else
            encrypt_group<ROUNDS, P, false, COMPACT>(a, mtab, ttab, ctab, cover, plw, lane, group, nonce_lo, nonce_hi);
        __builtin_amdgcn_wave_barrier();        // cover[] is private to this wave; its LDS operations run in order
        // src/SecretKey.cpp:51-76 for plaintext 0: if ALL D secret positions came out 1 the chosen one is
        // cleared (the reference draws the chosen position first and forces it to 0 when the others are
        // all 1; drawing every position and clearing the chosen one afterwards is the same distribution).
        // Probability 2^-D per ciphertext, so the branch below is almost never taken.
        for (u32 cl = lane; cl < a.Gc; cl += kWave) {
            const u64 c = cbase + cl;
            if (c < a.first_ct || c >= a.first_ct + a.batch)
                continue;
            bool all = true;
            for (u32 bit = cl * U, left = U; left && all;) {
                const u32 w = bit >> 6, sh = bit & 63u, take = min(left, 64u - sh);
                const u64 need = take >= 64u ? ~0ull : ((1ull << take) - 1ull);
                all = ((cover[w] >> sh) & need) == need;
                bit += take;
                left -= take;
            }
            if (!all || plw[cl])
                continue;
            // with a single distinct secret position there are no "other" positions and the reference
            // never clears (its v stays 0, src/SecretKey.cpp:55-76)
            u32 secret_bits = 0;
            for (u32 k = 0; k < 2u * U; ++k)
                secret_bits += (u32)__popcll(reinterpret_cast<const u64 *>(a.mask)[k]);
            if (secret_bits < 2u)
                continue;
            const u32 pos = keyed_draw_pos<ROUNDS>(a.rng, nonce_lo, nonce_hi, c, a.key_idx, a.D);
            if (pos >= a.n_bits)
                continue;                                       // a corrupt d_key must not write outside the ciphertext
            __threadfence();                                    // this wave's stores of the word have landed
            u64 *word = reinterpret_cast<u64 *>(a.out) + (c - a.first_ct) * (u64)(2u * U) + (pos >> 6);
            atomicAnd(reinterpret_cast<unsigned long long *>(word), ~(1ull << (63u - (pos & 63u))));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// General form of the keyed encrypt (any dL, any alignment): one lane per ciphertext walks its units
// and evaluates the SAME keystream definition unit by unit (a whole ChaCha block per 16-byte unit,
// so 4x the generator work of the wave kernel).  Used for term sizes whose group would need more
// than 5 passes, for odd dL, and as the A/B partner of the wave kernel (knob enc_wave = 0).
template <int ROUNDS>
__global__ void __launch_bounds__(256) k_encrypt_keyed_ct(EncKeyed rng, const u64 *__restrict__ epoch,
                                                          const uint8_t *__restrict__ plain,
                                                          const u64 *__restrict__ key_idx,
                                                          const u64 *__restrict__ mask, u64 *__restrict__ out,
                                                          u64 first_ct, u64 batch, u32 dL, u32 U, u32 P, u32 Gc,
                                                          u32 D, u64 tail)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i >= batch)
        return;
    u32 nonce_lo = rng.nonce_lo, nonce_hi = rng.nonce_hi;
    if (epoch) {
        const u64 nn = (((u64)nonce_hi << 32) | nonce_lo) + *epoch;
        nonce_lo = (u32)nn;
        nonce_hi = (u32)(nn >> 32);
    }
    const u64 c = first_ct + i, g = c / Gc;
    const u32 r0 = (u32)(c - g * Gc) * U;
    const u32 pl = plain[i] & 1u;
    const u64 pm = pl ? ~0ull : 0ull;
    u64 *o = out + i * dL;
    bool all = true;
    u32 secret_bits = 0;
    for (u32 j = 0; j < U; ++j) {
        const u32 r = r0 + j, p = r >> 8, q = (r >> 6) & 3u, L = r & 63u;
        const u64 ctr = (g * P + p) * 64u + L;
        u32 x[16];
        chacha_block<ROUNDS>(rng, nonce_lo, nonce_hi, (u32)ctr, (u32)(ctr >> 32), x);
        u64 w[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32 lo = 0, hi = 0;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)                   // static register indices only
                if ((u32)qq == q) {
                    lo = x[4 * qq + 2 * h];
                    hi = x[4 * qq + 2 * h + 1];
                }
            w[h] = ((u64)hi << 32) | lo;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32 k = 2u * j + (u32)h;
            if (k >= dL)
                continue;
            u64 v = w[h];
            if (k == dL - 1u)
                v &= tail;
            const u64 m = mask[k];
            secret_bits += (u32)__popcll(m);
            all = all && ((v & m) == m);
            o[k] = v | (m & pm);
        }
    }
    if (!pl && all && secret_bits >= 2u) {     // one distinct position: the reference never clears
        const u32 pos = keyed_draw_pos<ROUNDS>(rng, nonce_lo, nonce_hi, c, key_idx, D);
        if ((pos >> 6) < dL)                               // a corrupt d_key must not write outside the ciphertext
            o[pos >> 6] &= ~(1ull << (63u - (pos & 63u)));     // same lane wrote the word: program order holds
    }
}

// ---------------------------------------------------------------------------------------
// Fused fresh chain (SURVEY 8f-2 "fused chains"; the reference's canonical flow
// tests/basic_operations.cpp:26-40: encrypt, encrypt, operator*, decrypt): pair c's product
//     out_c = Enc_A(plain_a[c]) & Enc_B(plain_b[c])
// with BOTH operands generated in registers -- operand A from (rng_a, position c), operand B from
// (rng_b, position c), each exactly the ciphertext csgn_encrypt_keyed would have written -- ANDed
// (Ciphertext::defaultN_multiply, src/Ciphertext.cpp:124-131) and stored once: 8*dL bytes of HBM
// traffic per pair where the unfused chain makes five passes (two ciphertexts written, both read
// back, the product written).  Optionally the product is decrypted on the spot: a 1x1 product
// decrypts to 1 iff every unit of BOTH operands covers the key mask (src/SecretKey.cpp:82-102 on a
// word-wise AND), which the kernel knows from the cover ballots it needs for the plaintext-0 rule
// anyway.  That rule (clear the drawn position when all D came out 1) is applied to the PRODUCT:
// clearing a bit of one factor clears the same bit of the AND.
// ---------------------------------------------------------------------------------------
struct EncMulArgs {
    EncKeyed rng_a, rng_b;
    const u64 *epoch;
    const uint8_t *plain_a, *plain_b;
    const u64 *key_idx;
    const unit16 *mask;
    unit16 *out;
    uint8_t *bits;                // optional: Dec(out_c), one byte per pair
    u64 first_ct, batch, group0, ngroups;
    u32 U, Gc, D, iters, n_bits;
    u32 tail_lo, tail_hi;
    FastDiv dU;
};

struct EncMulPassCtx {
    const EncMulArgs &a;
    const unit16 *mtab;
    const u32 *etab;
    const signed char *plw_a, *plw_b;                              // plaintext bits as bytes 0x00 / 0xFF
    unit16 *outg;
    u64 blk0;
    u32 lane, cl_lo, cl_hi, na_lo, na_hi, nb_lo, nb_hi;
};

template <int Q, int PI>
__device__ __forceinline__ void encmul_unit(const EncMulPassCtx &c, const u32 (&xa)[16], const u32 (&xb)[16], u32 &aa_lo,
                                            u32 &aa_hi, u32 &ab_lo, u32 &ab_hi)
{
    const u32 r = (u32)PI * 256u + (u32)Q * 64u + c.lane;
    const u32 e = c.etab[r];
    const u32 cl = e & 0xFFFFu;
    const unit16 m = *reinterpret_cast<const unit16 *>(reinterpret_cast<const unsigned char *>(c.mtab) + ((e >> 16) & 0x7FFFu));
    const u32 inner = (u32)((int)e >> 31);                         // ~0 unless this is a ciphertext's last unit
    const u32 tlx = c.a.tail_lo | inner, tly = c.a.tail_hi | inner;
    unit16 va, vb;
    va.x = xa[4 * Q]; va.y = xa[4 * Q + 1]; va.z = xa[4 * Q + 2]; va.w = xa[4 * Q + 3];
    vb.x = xb[4 * Q]; vb.y = xb[4 * Q + 1]; vb.z = xb[4 * Q + 2]; vb.w = xb[4 * Q + 3];
    // cover verdicts on the words before the tail mask: the key mask has no bit in the padding
    write_lane64(__ballot(unit_covers_chain(va, m)), PI * 4 + Q, aa_lo, aa_hi);
    write_lane64(__ballot(unit_covers_chain(vb, m)), PI * 4 + Q, ab_lo, ab_hi);
    // plaintext 1 (byte 0xFF, read sign-extended -> ~0): OR the key mask in (src/SecretKey.cpp:44-45)
    const u32 pma = (u32)(int)c.plw_a[cl], pmb = (u32)(int)c.plw_b[cl];
    unit16 v;                                                      // Ciphertext::defaultN_multiply, src/Ciphertext.cpp:124-131
    v.x = (va.x | (m.x & pma)) & (vb.x | (m.x & pmb));
    v.y = (va.y | (m.y & pma)) & (vb.y | (m.y & pmb));
    // the padding bits of a last word: masked once, on the product (the key mask has none, so masking each
    // operand first gives the same word)
    v.z = (va.z | (m.z & pma)) & (vb.z | (m.z & pmb)) & tlx;
    v.w = (va.w | (m.w & pma)) & (vb.w | (m.w & pmb)) & tly;
    if (cl - c.cl_lo < c.cl_hi - c.cl_lo)
        unit_store<unit16, true>(c.outg + ((u32)PI * 256u + c.lane) + Q * 64, v);
}

template <int ROUNDS, int PI>
__device__ __forceinline__ void encmul_pass(const EncMulPassCtx &c, u32 &aa_lo, u32 &aa_hi, u32 &ab_lo, u32 &ab_hi)
{
    const u64 ctr = c.blk0 + (u64)PI * 64u;
    u32 xa[16], xb[16];
    chacha_block2<ROUNDS>(c.a.rng_a, c.na_lo, c.na_hi, (u32)ctr, (u32)(ctr >> 32), c.a.rng_b, c.nb_lo, c.nb_hi, (u32)ctr,
                          (u32)(ctr >> 32), xa, xb);
    encmul_unit<0, PI>(c, xa, xb, aa_lo, aa_hi, ab_lo, ab_hi);
    encmul_unit<1, PI>(c, xa, xb, aa_lo, aa_hi, ab_lo, ab_hi);
    encmul_unit<2, PI>(c, xa, xb, aa_lo, aa_hi, ab_lo, ab_hi);
    encmul_unit<3, PI>(c, xa, xb, aa_lo, aa_hi, ab_lo, ab_hi);
}

template <int ROUNDS, int P>
__global__ void __launch_bounds__(256) k_encrypt_mul_wave(EncMulArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr u32 R = 256u * P;
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const u32 U = a.U;
    // [mtab: U mask units][etab: R u32][cover_a, cover_b: 4 waves x P*4 words each][plw_a, plw_b: 4 x 256 bytes][secret bit count]
    unit16 *mtab = reinterpret_cast<unit16 *>(smem_raw);
    u32 *etab = reinterpret_cast<u32 *>(smem_raw + (size_t)U * 16u);
    u64 *cover_all = reinterpret_cast<u64 *>(smem_raw + (((size_t)U * 16u + (size_t)R * 4u + 7u) & ~(size_t)7u));
    u64 *cover_a = cover_all + wave * (P * 4), *cover_b = cover_all + (4u + wave) * (P * 4);
    signed char *plw_a = reinterpret_cast<signed char *>(cover_all + 8u * P * 4u) + wave * 256u;
    signed char *plw_b = plw_a + 4u * 256u;
    u32 *secret_bits = reinterpret_cast<u32 *>(reinterpret_cast<unsigned char *>(cover_all + 8u * P * 4u) + 8u * 256u);
    if (tid == 0)
        *secret_bits = 0;
    __syncthreads();
    u32 pop = 0;
    for (u32 k = tid; k < U; k += 256u) {
        const unit16 m = a.mask[k];
        mtab[k] = m;
        pop += __popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w);
    }
    if (pop)
        atomicAdd(secret_bits, pop);
    for (u32 r = tid; r < R; r += 256u) {
        const u32 cl = csgn_fastdiv(r, a.dU), j = r - cl * U;
        etab[r] = cl | ((j * 16u) << 16) | (j == U - 1u ? 0u : 0x80000000u);   // bit 31: NOT the last unit
    }
    __syncthreads();
    // with a single distinct secret position the reference never clears (src/SecretKey.cpp:55-76)
    const bool fixable = *secret_bits >= 2u;

    u32 na_lo = a.rng_a.nonce_lo, na_hi = a.rng_a.nonce_hi, nb_lo = a.rng_b.nonce_lo, nb_hi = a.rng_b.nonce_hi;
    if (a.epoch) {
        const u64 e = *a.epoch;
        const u64 na = (((u64)na_hi << 32) | na_lo) + e, nb = (((u64)nb_hi << 32) | nb_lo) + e;
        na_lo = (u32)na; na_hi = (u32)(na >> 32);
        nb_lo = (u32)nb; nb_hi = (u32)(nb >> 32);
    }
    for (u32 it = 0; it < a.iters; ++it) {
        const u64 gi = ((u64)it * gridDim.x + blockIdx.x) * 4u + wave;
        if (gi >= a.ngroups)
            break;
        const u64 group = a.group0 + gi;
        const u64 cbase = group * a.Gc;
        const u64 end = a.first_ct + a.batch;
        const u32 cl_lo = cbase < a.first_ct ? (u32)(a.first_ct - cbase) : 0u;
        const u32 cl_hi = cbase + a.Gc > end ? (u32)(end - cbase) : a.Gc;
        for (u32 cl = lane; cl < a.Gc; cl += kWave) {
            const bool in = cl - cl_lo < cl_hi - cl_lo;
            const u64 i = cbase + cl - a.first_ct;
            plw_a[cl] = in ? (signed char)(0 - (int)(a.plain_a[i] & 1u)) : (signed char)0;
            plw_b[cl] = in ? (signed char)(0 - (int)(a.plain_b[i] & 1u)) : (signed char)0;
        }
        __builtin_amdgcn_wave_barrier();
        const long long origin = (long long)cbase - (long long)a.first_ct;
        unit16 *outg = a.out + origin * (long long)U;
        const u64 blk0 = group * (u64)P * 64u + lane;
        u32 aa_lo = 0, aa_hi = 0, ab_lo = 0, ab_hi = 0;      // lane s keeps the cover ballots of slot s = p*4+q
        EncMulPassCtx cx{a, mtab, etab, plw_a, plw_b, outg, blk0, lane, cl_lo, cl_hi, na_lo, na_hi, nb_lo, nb_hi};
        // passes expanded by template index, not by a loop: the v_writelane lane numbers must be literal
        // constants, and hipcc does not promise to unroll a loop whose body holds two ChaCha blocks
        encmul_pass<ROUNDS, 0>(cx, aa_lo, aa_hi, ab_lo, ab_hi);
        if (P > 1) encmul_pass<ROUNDS, 1>(cx, aa_lo, aa_hi, ab_lo, ab_hi);
        if (P > 2) encmul_pass<ROUNDS, 2>(cx, aa_lo, aa_hi, ab_lo, ab_hi);
        if (P > 3) encmul_pass<ROUNDS, 3>(cx, aa_lo, aa_hi, ab_lo, ab_hi);
        if (P > 4) encmul_pass<ROUNDS, 4>(cx, aa_lo, aa_hi, ab_lo, ab_hi);
        if (lane < (u32)P * 4u) {
            cover_a[lane] = ((u64)aa_hi << 32) | aa_lo;
            cover_b[lane] = ((u64)ab_hi << 32) | ab_lo;
        }
        __builtin_amdgcn_wave_barrier();
        for (u32 cl = lane; cl < a.Gc; cl += kWave) {
            if (!(cl - cl_lo < cl_hi - cl_lo))
                continue;
            const u64 c = cbase + cl;
            bool all_a = true, all_b = true;
            for (u32 bit = cl * U, left = U; left;) {
                const u32 w = bit >> 6, sh = bit & 63u, take = min(left, 64u - sh);
                const u64 need = take >= 64u ? ~0ull : ((1ull << take) - 1ull);
                all_a = all_a && ((cover_a[w] >> sh) & need) == need;
                all_b = all_b && ((cover_b[w] >> sh) & need) == need;
                bit += take;
                left -= take;
            }
            const bool pl_a = plw_a[cl] != 0, pl_b = plw_b[cl] != 0;
            const bool fix_a = all_a && !pl_a && fixable, fix_b = all_b && !pl_b && fixable;
            if (fix_a || fix_b) {                             // probability 2^-D per operand
                __threadfence();                              // this wave's stores of the product have landed
                u64 *ct = reinterpret_cast<u64 *>(a.out) + (c - a.first_ct) * (u64)(2u * U);
                if (fix_a) {
                    const u32 pos = keyed_draw_pos<ROUNDS>(a.rng_a, na_lo, na_hi, c, a.key_idx, a.D);
                    if (pos < a.n_bits)
                        atomicAnd(reinterpret_cast<unsigned long long *>(ct + (pos >> 6)), ~(1ull << (63u - (pos & 63u))));
                }
                if (fix_b) {
                    const u32 pos = keyed_draw_pos<ROUNDS>(a.rng_b, nb_lo, nb_hi, c, a.key_idx, a.D);
                    if (pos < a.n_bits)
                        atomicAnd(reinterpret_cast<unsigned long long *>(ct + (pos >> 6)), ~(1ull << (63u - (pos & 63u))));
                }
            }
            if (a.bits)                                       // Dec of the product = both factors cover the key mask
                a.bits[c - a.first_ct] = ((pl_a || (all_a && !fix_a)) && (pl_b || (all_b && !fix_b))) ? 1 : 0;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// General form of the fused chain (odd dL, P > 5, unaligned buffers): one lane per pair.
template <int ROUNDS>
__global__ void __launch_bounds__(256) k_encrypt_mul_keyed_ct(EncKeyed rng_a, EncKeyed rng_b, const u64 *__restrict__ epoch,
                                                              const uint8_t *__restrict__ plain_a,
                                                              const uint8_t *__restrict__ plain_b,
                                                              const u64 *__restrict__ key_idx,
                                                              const u64 *__restrict__ mask, u64 *__restrict__ out,
                                                              uint8_t *__restrict__ bits, u64 first_ct, u64 batch,
                                                              u32 dL, u32 U, u32 P, u32 Gc, u32 D, u64 tail)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i >= batch)
        return;
    u32 na_lo = rng_a.nonce_lo, na_hi = rng_a.nonce_hi, nb_lo = rng_b.nonce_lo, nb_hi = rng_b.nonce_hi;
    if (epoch) {
        const u64 e = *epoch;
        const u64 na = (((u64)na_hi << 32) | na_lo) + e, nb = (((u64)nb_hi << 32) | nb_lo) + e;
        na_lo = (u32)na; na_hi = (u32)(na >> 32);
        nb_lo = (u32)nb; nb_hi = (u32)(nb >> 32);
    }
    const u64 c = first_ct + i, g = c / Gc;
    const u32 r0 = (u32)(c - g * Gc) * U;
    const bool pl_a = plain_a[i] & 1u, pl_b = plain_b[i] & 1u;
    const u64 pma = pl_a ? ~0ull : 0ull, pmb = pl_b ? ~0ull : 0ull;
    u64 *o = out + i * dL;
    bool all_a = true, all_b = true;
    u32 secret_bits = 0;
    for (u32 j = 0; j < U; ++j) {
        const u32 r = r0 + j, p = r >> 8, q = (r >> 6) & 3u, L = r & 63u;
        const u64 ctr = (g * P + p) * 64u + L;
        u32 xa[16], xb[16];
        chacha_block<ROUNDS>(rng_a, na_lo, na_hi, (u32)ctr, (u32)(ctr >> 32), xa);
        chacha_block<ROUNDS>(rng_b, nb_lo, nb_hi, (u32)ctr, (u32)(ctr >> 32), xb);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32 alo = 0, ahi = 0, blo = 0, bhi = 0;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)                   // static register indices only
                if ((u32)qq == q) {
                    alo = xa[4 * qq + 2 * h]; ahi = xa[4 * qq + 2 * h + 1];
                    blo = xb[4 * qq + 2 * h]; bhi = xb[4 * qq + 2 * h + 1];
                }
            const u32 k = 2u * j + (u32)h;
            if (k >= dL)
                continue;
            u64 va = ((u64)ahi << 32) | alo, vb = ((u64)bhi << 32) | blo;
            if (k == dL - 1u) {
                va &= tail;
                vb &= tail;
            }
            const u64 m = mask[k];
            secret_bits += (u32)__popcll(m);
            all_a = all_a && ((va & m) == m);
            all_b = all_b && ((vb & m) == m);
            o[k] = (va | (m & pma)) & (vb | (m & pmb));
        }
    }
    const bool fixable = secret_bits >= 2u;
    const bool fix_a = all_a && !pl_a && fixable, fix_b = all_b && !pl_b && fixable;
    if (fix_a) {
        const u32 pos = keyed_draw_pos<ROUNDS>(rng_a, na_lo, na_hi, c, key_idx, D);
        if ((pos >> 6) < dL)
            o[pos >> 6] &= ~(1ull << (63u - (pos & 63u)));
    }
    if (fix_b) {
        const u32 pos = keyed_draw_pos<ROUNDS>(rng_b, nb_lo, nb_hi, c, key_idx, D);
        if ((pos >> 6) < dL)
            o[pos >> 6] &= ~(1ull << (63u - (pos & 63u)));
    }
    if (bits)
        bits[i] = ((pl_a || (all_a && !fix_a)) && (pl_b || (all_b && !fix_b))) ? 1 : 0;
}

} // namespace

// ------------------------------------------------------------------------------ public

hipError_t encrypt(u64 n_bits, u64 d, u64 batch, const uint8_t *plain, const u64 *rnd,
                   const u32 *chosen, const uint8_t *last, const u64 *mask, u64 *out, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0)
        return hipSuccess;
    // fast form: K aligned 4 KiB segments = TB whole ciphertexts per workgroup, one lane per unit
    {
        const bool wide = (dL % 2 == 0) && aligned16(out) && aligned16(mask) && aligned16(rnd);
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
#define CSGN_ENC_SEG(UNIT, K)                                                                       \
    k_encrypt_seg<UNIT, K><<<(u32)nblk, 256, 0, s>>>(n_bits, (u32)dL, U, dU, d, batch, tb, plain,   \
                                                     reinterpret_cast<const UNIT *>(rnd), chosen, last, \
                                                     reinterpret_cast<const UNIT *>(mask),              \
                                                     reinterpret_cast<UNIT *>(out))
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
    k_encrypt<<<(u32)blocks64, 256, lds, s>>>(n_bits, (u32)dL, d, batch, cb, ddL, plain, rnd, chosen, last, mask, out);
    return hipGetLastError();
}

} // namespace csgn

namespace csgn {

namespace {
u32 gcd_u32(u32 a, u32 b)
{
    while (b) {
        const u32 t = a % b;
        a = b;
        b = t;
    }
    return a;
}
} // namespace

namespace {
__global__ void k_bump_epoch(u64 *epoch) { *epoch += 1; }
} // namespace

hipError_t bump_epoch(u64 *d_epoch, hipStream_t s)
{
    k_bump_epoch<<<1, 1, 0, s>>>(d_epoch);
    return hipGetLastError();
}

void encrypt_keyed_layout(u64 n_bits, u32 *U, u32 *P, u32 *Gc)
{
    const u64 dL = (n_bits + 63) / 64;
    *U = (u32)((dL + 1) / 2);
    *P = *U / gcd_u32(*U, 256u);
    *Gc = 256u * *P / *U;
}

hipError_t encrypt_keyed(u64 n_bits, u64 d, u64 batch, u64 first_ct, const uint8_t *plain, const u64 *key_idx,
                         const u64 *mask, const u32 rng_key[8], u64 nonce, u32 rounds,
                         const u64 *d_epoch, u64 *out, hipStream_t s)
{
    if (batch == 0)
        return hipSuccess;
    if (rounds != 8 && rounds != 12 && rounds != 20)
        return hipErrorInvalidValue;
    const u64 dL = (n_bits + 63) / 64;
    u32 U, P, Gc;
    encrypt_keyed_layout(n_bits, &U, &P, &Gc);
    const u32 rem = (u32)(n_bits & 63);
    const u64 tail = rem ? ~0ull << (64 - rem) : ~0ull;
    EncKeyed rk;
    for (int i = 0; i < 8; ++i)
        rk.key[i] = rng_key[i];
    rk.nonce_lo = (u32)nonce;
    rk.nonce_hi = (u32)(nonce >> 32);
    const bool wave_ok = dL % 2 == 0 && P <= 5 && aligned16(out) && aligned16(mask) && tune(TUNE_ENC_WAVE) != 0;
    if (wave_ok) {
        EncWaveArgs a;
        a.rng = rk;
        a.epoch = d_epoch;
        a.plain = plain;
        a.key_idx = key_idx;
        a.mask = reinterpret_cast<const unit16 *>(mask);
        a.out = reinterpret_cast<unit16 *>(out);
        a.first_ct = first_ct;
        a.batch = batch;
        a.iters = 1;
        a.group0 = first_ct / Gc;
        a.ngroups = (first_ct + batch - 1) / Gc - a.group0 + 1;
        a.U = U;
        a.Gc = Gc;
        a.D = (u32)d;
        a.n_bits = (u32)n_bits;
        a.tail_lo = (u32)tail;
        a.tail_hi = (u32)(tail >> 32);
        a.dU = csgn_fastdiv_make(U);
        // persistent workgroups: the LDS tables are built once per workgroup, so each wave should walk
        // several groups; 256 CUs x 4 workgroups fill the chip (33 KB of LDS per workgroup at P = 5)
        int cus = 256;
        {
            int dev = 0;
            if (hipGetDevice(&dev) == hipSuccess)
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        }
        const u64 wg_needed = (a.ngroups + 3) / 4;
        // Compact LDS tables where the full ones would limit residency (U*16 <= 32 KB for the offset
        // field; knob enc_compact: -1 = auto, 0 / 1 forced)
        const int ck = tune(TUNE_ENC_COMPACT);
        const bool compact = U * 16u <= 32768u && (ck < 0 ? P >= 3 : ck != 0);
        // Workgroups per CU (knob enc_wave = k > 1 forces k; >= 64 = one group per wave): 16 with the
        // small tables, 4 where the full tables take 33 KB of LDS
        const int per_cu = tune(TUNE_ENC_WAVE) > 1 ? tune(TUNE_ENC_WAVE) : ((P >= 3 && !compact) ? 4 : 16);
        const u64 resident = per_cu >= 64 ? wg_needed : (u64)cus * (u64)per_cu;
        u64 blocks = wg_needed <= resident ? wg_needed : resident;
        a.iters = (u32)((wg_needed + blocks - 1) / blocks);
        blocks = (wg_needed + a.iters - 1) / a.iters;         // equal trip counts
        const size_t lds = (compact ? (((size_t)U * 16u + 256u * P * 4u + 7u) & ~(size_t)7u)
                                    : (size_t)256u * P * 24u) + 4u * P * 4u * 8u + 4u * 256u /* plw */ +
                           (compact ? 0u : (size_t)256u * P * 2u);
#define CSGN_ENC_WAVE(R, PP)                                                \
    do {                                                                    \
        if (compact)                                                        \
            k_encrypt_wave<R, PP, true><<<(u32)blocks, 256, lds, s>>>(a);   \
        else                                                                \
            k_encrypt_wave<R, PP, false><<<(u32)blocks, 256, lds, s>>>(a);  \
    } while (0)
#define CSGN_ENC_WAVE_P(R)                  \
    switch (P) {                            \
    case 1: CSGN_ENC_WAVE(R, 1); break;     \
    case 2: CSGN_ENC_WAVE(R, 2); break;     \
    case 3: CSGN_ENC_WAVE(R, 3); break;     \
    case 4: CSGN_ENC_WAVE(R, 4); break;     \
    default: CSGN_ENC_WAVE(R, 5); break;    \
    }
        if (rounds == 8) {
            CSGN_ENC_WAVE_P(8)
        } else if (rounds == 12) {
            CSGN_ENC_WAVE_P(12)
        } else {
            CSGN_ENC_WAVE_P(20)
        }
#undef CSGN_ENC_WAVE_P
#undef CSGN_ENC_WAVE
        return hipGetLastError();
    }
    const u64 blocks = (batch + 255) / 256;
    if (blocks > kMaxBlocks256)
        return hipErrorInvalidValue;
#define CSGN_ENC_CT(R)                                                                                       \
    k_encrypt_keyed_ct<R><<<(u32)blocks, 256, 0, s>>>(rk, d_epoch, plain, key_idx, mask, out, first_ct, batch, \
                                                      (u32)dL, U, P, Gc, (u32)d, tail)
    if (rounds == 8)
        CSGN_ENC_CT(8);
    else if (rounds == 12)
        CSGN_ENC_CT(12);
    else
        CSGN_ENC_CT(20);
#undef CSGN_ENC_CT
    return hipGetLastError();
}

hipError_t encrypt_mul_keyed(u64 n_bits, u64 d, u64 batch, u64 first_ct, const uint8_t *plain_a,
                             const uint8_t *plain_b, const u64 *key_idx, const u64 *mask, const u32 key_a[8],
                             u64 nonce_a, const u32 key_b[8], u64 nonce_b, u32 rounds, const u64 *d_epoch, u64 *out,
                             uint8_t *bits, hipStream_t s)
{
    if (batch == 0)
        return hipSuccess;
    if (rounds != 8 && rounds != 12 && rounds != 20)
        return hipErrorInvalidValue;
    const u64 dL = (n_bits + 63) / 64;
    u32 U, P, Gc;
    encrypt_keyed_layout(n_bits, &U, &P, &Gc);
    const u32 rem = (u32)(n_bits & 63);
    const u64 tail = rem ? ~0ull << (64 - rem) : ~0ull;
    EncKeyed ra, rb;
    for (int i = 0; i < 8; ++i) {
        ra.key[i] = key_a[i];
        rb.key[i] = key_b[i];
    }
    ra.nonce_lo = (u32)nonce_a;
    ra.nonce_hi = (u32)(nonce_a >> 32);
    rb.nonce_lo = (u32)nonce_b;
    rb.nonce_hi = (u32)(nonce_b >> 32);
    const bool wave_ok = dL % 2 == 0 && P <= 5 && U * 16u <= 32768u && aligned16(out) && aligned16(mask) &&
                         tune(TUNE_ENC_WAVE) != 0;
    if (wave_ok) {
        EncMulArgs a;
        a.rng_a = ra;
        a.rng_b = rb;
        a.epoch = d_epoch;
        a.plain_a = plain_a;
        a.plain_b = plain_b;
        a.key_idx = key_idx;
        a.mask = reinterpret_cast<const unit16 *>(mask);
        a.out = reinterpret_cast<unit16 *>(out);
        a.bits = bits;
        a.first_ct = first_ct;
        a.batch = batch;
        a.group0 = first_ct / Gc;
        a.ngroups = (first_ct + batch - 1) / Gc - a.group0 + 1;
        a.U = U;
        a.Gc = Gc;
        a.D = (u32)d;
        a.n_bits = (u32)n_bits;
        a.tail_lo = (u32)tail;
        a.tail_hi = (u32)(tail >> 32);
        a.dU = csgn_fastdiv_make(U);
        int cus = 256;
        {
            int dev = 0;
            if (hipGetDevice(&dev) == hipSuccess)
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        }
        const u64 wg_needed = (a.ngroups + 3) / 4;
        const u64 resident = (u64)cus * 4u;      // 98-122 VGPRs: four workgroups of four waves per CU
        u64 blocks = wg_needed <= resident ? wg_needed : resident;
        a.iters = (u32)((wg_needed + blocks - 1) / blocks);
        blocks = (wg_needed + a.iters - 1) / a.iters;
        const size_t lds = (((size_t)U * 16u + 256u * P * 4u + 7u) & ~(size_t)7u) + 8u * P * 4u * 8u + 8u * 256u + 8u;
#define CSGN_ENCMUL_P(R)                                                                 \
    switch (P) {                                                                         \
    case 1: k_encrypt_mul_wave<R, 1><<<(u32)blocks, 256, lds, s>>>(a); break;            \
    case 2: k_encrypt_mul_wave<R, 2><<<(u32)blocks, 256, lds, s>>>(a); break;            \
    case 3: k_encrypt_mul_wave<R, 3><<<(u32)blocks, 256, lds, s>>>(a); break;            \
    case 4: k_encrypt_mul_wave<R, 4><<<(u32)blocks, 256, lds, s>>>(a); break;            \
    default: k_encrypt_mul_wave<R, 5><<<(u32)blocks, 256, lds, s>>>(a); break;           \
    }
        if (rounds == 8) {
            CSGN_ENCMUL_P(8)
        } else if (rounds == 12) {
            CSGN_ENCMUL_P(12)
        } else {
            CSGN_ENCMUL_P(20)
        }
#undef CSGN_ENCMUL_P
        return hipGetLastError();
    }
    const u64 blocks = (batch + 255) / 256;
    if (blocks > kMaxBlocks256)
        return hipErrorInvalidValue;
#define CSGN_ENCMUL_CT(R)                                                                                      \
    k_encrypt_mul_keyed_ct<R><<<(u32)blocks, 256, 0, s>>>(ra, rb, d_epoch, plain_a, plain_b, key_idx, mask, out, \
                                                          bits, first_ct, batch, (u32)dL, U, P, Gc, (u32)d, tail)
    if (rounds == 8)
        CSGN_ENCMUL_CT(8);
    else if (rounds == 12)
        CSGN_ENCMUL_CT(12);
    else
        CSGN_ENCMUL_CT(20);
#undef CSGN_ENCMUL_CT
    return hipGetLastError();
}

} // namespace csgn
