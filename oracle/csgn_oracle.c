/*
 * csgn_oracle.c -- CPU restatement of the certFHE/CSGN ciphertext-arithmetic hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see csgn_oracle.h).  Plain C99, single-threaded, no
 * dependencies.  Parity: PINNED against oracle/_ref (the real reference, compiled by
 * oracle/Makefile) and against tests/golden/.
 *
 * Citations are /root/reference/-relative.  The reference's 32-bit `int` loop counters
 * and its out-of-bounds writes are NOT reproduced: this file restates the
 * results the reference produces where it works, with 64-bit indexing throughout.
 */
#include "csgn_oracle.h"

#include <stdlib.h>
#include <string.h>

#define WORD_BITS 64u
#define GOLDEN 0x9E3779B97F4A7C15ull

/* ------------------------------------------------------------------ helpers ---- */

static uint64_t splitmix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* bit `pos` (0 = most significant) of word w -- the reference's MSB-first convention,
 * src/SecretKey.cpp:181 / src/SecretKey.cpp:92-93. */
static unsigned msb_bit(uint64_t w, unsigned pos)
{
    return (unsigned)((w >> (WORD_BITS - 1u - pos)) & 1u);
}

static int key_contains(const uint64_t *key, uint64_t d, uint64_t value)
{
    /* src/Helpers.cpp:18-26 */
    for (uint64_t i = 0; i < d; ++i)
        if (key[i] == value)
            return 1;
    return 0;
}

/* Unpack a (v, bitlen) stream into one byte per bit (src/SecretKey.cpp:110-124,
 * src/Ciphertext.cpp:16-31).  Returns the number of bits; *bits_out is malloc'ed. */
static uint64_t unpack_stream(uint64_t n_bits, const uint64_t *v, uint64_t len,
                              const uint64_t *bitlen, uint8_t **bits_out)
{
    uint64_t dl = csgn_oracle_default_len(n_bits);
    uint64_t rem = n_bits % WORD_BITS;
    uint64_t total = 0;
    for (uint64_t i = 0; i < len; ++i) {
        uint64_t b = bitlen ? bitlen[i]
                            : ((rem != 0 && dl != 0 && (i % dl) == dl - 1) ? rem : WORD_BITS);
        total += b;
    }
    uint8_t *bits = (uint8_t *)malloc(total ? total : 1);
    uint64_t q = 0;
    for (uint64_t i = 0; i < len; ++i) {
        uint64_t b = bitlen ? bitlen[i]
                            : ((rem != 0 && dl != 0 && (i % dl) == dl - 1) ? rem : WORD_BITS);
        for (uint64_t k = 0; k < b; ++k)
            bits[q++] = (uint8_t)msb_bit(v[i], (unsigned)k);
    }
    *bits_out = bits;
    return total;
}

/* ------------------------------------------------------------------ Context ---- */

uint64_t csgn_oracle_default_len(uint64_t n_bits)
{
    /* src/Context.cpp:24-28 */
    return n_bits / WORD_BITS + ((n_bits % WORD_BITS) ? 1u : 0u);
}

uint64_t csgn_oracle_context_s(uint64_t n_bits, uint64_t d)
{
    /* src/Context.cpp:22 */
    return n_bits / (2 * d);
}

void csgn_oracle_bitlen(uint64_t n_bits, uint64_t terms, uint64_t *bitlen)
{
    /* src/SecretKey.cpp:171-173 (pattern per term; no overflow when N%64==0) */
    uint64_t dl = csgn_oracle_default_len(n_bits);
    uint64_t rem = n_bits % WORD_BITS;
    for (uint64_t t = 0; t < terms; ++t)
        for (uint64_t k = 0; k < dl; ++k)
            bitlen[t * dl + k] = (rem != 0 && k == dl - 1) ? rem : WORD_BITS;
}

/* ----------------------------------------------------------------- multiply ---- */

uint64_t csgn_oracle_mul_len(uint64_t dl, uint64_t len1, uint64_t len2)
{
    if (dl == 0)
        return 0;
    if (len1 == dl && len1 == len2)          /* src/Ciphertext.cpp:137-138 */
        return len1;
    return ((len1 / dl) * len2) / dl * dl;   /* src/Ciphertext.cpp:146, as parsed by C */
}

uint64_t csgn_oracle_mul(uint64_t dl,
                         const uint64_t *c1, uint64_t len1, const uint64_t *bitlen_in1,
                         const uint64_t *c2, uint64_t len2,
                         uint64_t *out, uint64_t *bitlen_out)
{
    uint64_t newlen = csgn_oracle_mul_len(dl, len1, len2);
    if (dl == 0)
        return 0;

    if (len1 == dl && len1 == len2) {
        /* fast path: src/Ciphertext.cpp:124-131, 137-144 */
        for (uint64_t k = 0; k < dl; ++k)
            out[k] = c1[k] & c2[k];
        if (bitlen_in1 && bitlen_out)
            memcpy(bitlen_out, bitlen_in1, dl * sizeof(uint64_t));
        return newlen;
    }

    /* general path: src/Ciphertext.cpp:150-163.  Output term (i*T2 + j) is the AND of
     * left term i with right term j; the left operand is the slow index. */
    uint64_t t1 = len1 / dl, t2 = len2 / dl;
    uint64_t *dst = out;
    for (uint64_t i = 0; i < t1; ++i) {
        const uint64_t *lhs = c1 + i * dl;
        const uint64_t *rhs = c2;
        for (uint64_t j = 0; j < t2; ++j, rhs += dl, dst += dl)
            for (uint64_t k = 0; k < dl; ++k)
                dst[k] = lhs[k] & rhs[k];
    }
    /* words the reference leaves unwritten when len2 is not a multiple of dl */
    for (uint64_t w = t1 * t2 * dl; w < newlen; ++w)
        out[w] = 0;

    if (bitlen_in1 && bitlen_out) {
        /* second pass: src/Ciphertext.cpp:165-176 -- bitlen comes from the LEFT term */
        uint64_t *bd = bitlen_out;
        for (uint64_t i = 0; i < t1; ++i)
            for (uint64_t j = 0; j < t2; ++j, bd += dl)
                memcpy(bd, bitlen_in1 + i * dl, dl * sizeof(uint64_t));
        for (uint64_t w = t1 * t2 * dl; w < newlen; ++w)
            bitlen_out[w] = 0;
    }
    return newlen;
}

uint64_t csgn_oracle_mul_reference_cost(uint64_t dl,
                                        const uint64_t *c1, uint64_t len1,
                                        const uint64_t *c2, uint64_t len2)
{
    /* Cost structure of Ciphertext::operator* (src/Ciphertext.cpp:231-247): the L1
     * kernel allocates res + bitlenout and makes two passes (:148-176); the result
     * object then deep-copies both arrays (:344-358) and the temporaries are freed. */
    uint64_t newlen = csgn_oracle_mul_len(dl, len1, len2);
    uint64_t *bitlen1 = (uint64_t *)malloc((len1 ? len1 : 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < len1; ++i)
        bitlen1[i] = WORD_BITS;
    uint64_t *res = (uint64_t *)malloc((newlen ? newlen : 1) * sizeof(uint64_t));
    uint64_t *blo = (uint64_t *)malloc((newlen ? newlen : 1) * sizeof(uint64_t));
    csgn_oracle_mul(dl, c1, len1, bitlen1, c2, len2, res, blo);
    uint64_t *v_copy = (uint64_t *)malloc((newlen ? newlen : 1) * sizeof(uint64_t));
    uint64_t *b_copy = (uint64_t *)malloc((newlen ? newlen : 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < newlen; ++i) {
        v_copy[i] = res[i];
        b_copy[i] = blo[i];
    }
    free(res);
    free(blo);
    uint64_t dig = csgn_oracle_digest(v_copy, newlen, 0) + b_copy[newlen ? newlen - 1 : 0];
    free(v_copy);
    free(b_copy);
    free(bitlen1);
    return dig;
}

/* ---------------------------------------------------------------------- add ---- */

uint64_t csgn_oracle_add(const uint64_t *c1, uint64_t len1, const uint64_t *bitlen1,
                         const uint64_t *c2, uint64_t len2, const uint64_t *bitlen2,
                         uint64_t *out, uint64_t *bitlen_out)
{
    /* src/Ciphertext.cpp:107-122: plain concatenation, no XOR / no de-duplication. */
    if (len1)
        memcpy(out, c1, len1 * sizeof(uint64_t));
    if (len2)
        memcpy(out + len1, c2, len2 * sizeof(uint64_t));
    if (bitlen_out) {
        /* src/Ciphertext.cpp:215-223 */
        if (len1 && bitlen1)
            memcpy(bitlen_out, bitlen1, len1 * sizeof(uint64_t));
        if (len2 && bitlen2)
            memcpy(bitlen_out + len1, bitlen2, len2 * sizeof(uint64_t));
    }
    return len1 + len2;
}

/* ------------------------------------------------------------------- keygen ---- */

int64_t csgn_oracle_keygen(uint64_t n_bits, uint64_t d,
                           const int32_t *draws, uint64_t n_draws, uint64_t *key)
{
    /* src/SecretKey.cpp:322-335: rejection-sample d distinct indices in [0,N).
     * The reference tests membership against the whole (partly uninitialised) array;
     * the restatement tests against the indices accepted so far. */
    uint64_t used = 0, count = 0;
    while (count < d) {
        if (used >= n_draws)
            return -1;
        uint64_t cand = (uint64_t)draws[used++] % n_bits;
        if (key_contains(key, count, cand))
            continue;
        key[count++] = cand;
    }
    return (int64_t)used;
}

void csgn_oracle_key_mask(uint64_t n_bits, const uint64_t *key, uint64_t d, uint64_t *mask)
{
    uint64_t dl = csgn_oracle_default_len(n_bits);
    memset(mask, 0, dl * sizeof(uint64_t));
    for (uint64_t i = 0; i < d; ++i)
        mask[key[i] / WORD_BITS] |= 1ull << (WORD_BITS - 1u - (unsigned)(key[i] % WORD_BITS));
}

/* ------------------------------------------------------------------ encrypt ---- */

int64_t csgn_oracle_encrypt(uint64_t n_bits, uint64_t d, const uint64_t *key,
                            unsigned bit, const int32_t *draws, uint64_t n_draws,
                            uint64_t *out)
{
    uint64_t dl = csgn_oracle_default_len(n_bits);
    uint8_t *vec = (uint8_t *)calloc(n_bits ? n_bits : 1, 1);
    uint64_t used = 0;
    int ok = 1;

#define NEXT_DRAW(dst)                         \
    do {                                       \
        if (used >= n_draws) { ok = 0; }       \
        else { (dst) = draws[used++]; }        \
    } while (0)

    if (bit & 1u) {
        /* src/SecretKey.cpp:41-48: secret positions forced to 1, the rest rand()%2 */
        for (uint64_t i = 0; i < n_bits && ok; ++i) {
            if (key_contains(key, d, i)) {
                vec[i] = 1;
            } else {
                int32_t r = 0;
                NEXT_DRAW(r);
                vec[i] = (uint8_t)(r % 2);
            }
        }
    } else {
        /* src/SecretKey.cpp:51-76: pick one secret slot, randomise everything else,
         * then force the chosen slot to 0 iff every OTHER secret slot came out 1. */
        int32_t r = 0;
        NEXT_DRAW(r);
        uint64_t chosen = key[(uint64_t)r % d];
        unsigned and_of_others = 0;
        int first = 1;
        for (uint64_t i = 0; i < n_bits && ok; ++i) {
            if (i == chosen)
                continue;
            NEXT_DRAW(r);
            vec[i] = (uint8_t)(r % 2);
            if (key_contains(key, d, i)) {
                if (first) {
                    and_of_others = vec[i];
                    first = 0;
                }
                and_of_others &= vec[i];
            }
        }
        if (ok) {
            if (and_of_others == 1u) {
                vec[chosen] = 0;
            } else {
                NEXT_DRAW(r);
                vec[chosen] = (uint8_t)(r % 2);
            }
        }
    }
#undef NEXT_DRAW

    if (ok) {
        /* src/SecretKey.cpp:175-197: pack MSB-first, word j/64, bit 63 - j%64 */
        memset(out, 0, dl * sizeof(uint64_t));
        for (uint64_t j = 0; j < n_bits; ++j)
            out[j / WORD_BITS] |= (uint64_t)(vec[j] & 1u)
                                  << (WORD_BITS - 1u - (unsigned)(j % WORD_BITS));
    }
    free(vec);
    return ok ? (int64_t)used : -1;
}

/* ---- keyed device generator (definition shared with the HIP side, NOT reference code) ----
 * csgn_encrypt_keyed (include/csgn_hip.h) draws its randomness from ChaCha in counter mode.  This
 * is an independent restatement of that definition: the ChaCha block function as published by
 * D. J. Bernstein ("ChaCha, a variant of Salsa20", 2008; 64-bit counter in state words 12-13,
 * 64-bit nonce in 14-15), checked in tests/test_oracle_golden.py against the RFC 8439 section
 * 2.3.2 known-answer block; the keystream layout and the plaintext rule follow the comment in
 * include/csgn_hip.h.  The plaintext-0 rule is proven distribution-identical to
 * src/SecretKey.cpp:51-76 by exhaustive enumeration in tests/test_oracle_golden.py. */
#define ROTL32(x, n) (((x) << (n)) | ((x) >> (32 - (n))))
#define QR(a, b, c, d)                    \
    a += b; d ^= a; d = ROTL32(d, 16);    \
    c += d; b ^= c; b = ROTL32(b, 12);    \
    a += b; d ^= a; d = ROTL32(d, 8);     \
    c += d; b ^= c; b = ROTL32(b, 7)

/* ChaCha block with the 16 constant bytes given as text: "expand 32-byte k" is Bernstein's own (the
 * keystream); "csgn draw pos v1" gives the stream the plaintext-0 rule draws its position from,
 * "csgn node key v1" (20 rounds) the key of a circuit's encrypt node (include/csgn_hip.h). */
static void chacha_block_sigma(const char sigma[16], const uint32_t key[8], uint64_t nonce, uint64_t counter,
                               unsigned rounds, uint32_t out[16])
{
    uint32_t in[16], x[16];
    for (int i = 0; i < 4; ++i)
        in[i] = (uint32_t)(unsigned char)sigma[4 * i] | (uint32_t)(unsigned char)sigma[4 * i + 1] << 8 |
                (uint32_t)(unsigned char)sigma[4 * i + 2] << 16 | (uint32_t)(unsigned char)sigma[4 * i + 3] << 24;
    for (int i = 0; i < 8; ++i)
        in[4 + i] = key[i];
    in[12] = (uint32_t)counter;
    in[13] = (uint32_t)(counter >> 32);
    in[14] = (uint32_t)nonce;
    in[15] = (uint32_t)(nonce >> 32);
    memcpy(x, in, sizeof(x));
    for (unsigned r = 0; r < rounds; r += 2) {
        QR(x[0], x[4], x[8], x[12]);
        QR(x[1], x[5], x[9], x[13]);
        QR(x[2], x[6], x[10], x[14]);
        QR(x[3], x[7], x[11], x[15]);
        QR(x[0], x[5], x[10], x[15]);
        QR(x[1], x[6], x[11], x[12]);
        QR(x[2], x[7], x[8], x[13]);
        QR(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; ++i)
        out[i] = x[i] + in[i];
}
#undef QR
#undef ROTL32

void csgn_oracle_chacha_block(const uint32_t key[8], uint64_t nonce, uint64_t counter,
                              unsigned rounds, uint32_t out[16])
{
    chacha_block_sigma("expand 32-byte k", key, nonce, counter, rounds, out);
}

/* key of a circuit encrypt node built from (key, nonce): csgn_circuit_node_key */
void csgn_oracle_node_key(const uint32_t key[8], uint64_t nonce, uint32_t node_key[8])
{
    uint32_t block[16];
    chacha_block_sigma("csgn node key v1", key, nonce, 0, 20, block);
    memcpy(node_key, block, 8 * sizeof(uint32_t));
}

void csgn_oracle_rng_from_seed(uint64_t seed, uint32_t key[8], uint64_t *nonce)
{
    for (int i = 0; i < 4; ++i) {
        uint64_t w = splitmix64(seed + GOLDEN * (uint64_t)(i + 1));
        key[2 * i] = (uint32_t)w;
        key[2 * i + 1] = (uint32_t)(w >> 32);
    }
    *nonce = splitmix64(seed ^ 0xD1B54A32D192ED03ull);
}

static uint64_t gcd_u64(uint64_t a, uint64_t b)
{
    while (b) {
        uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

void csgn_oracle_keyed_layout(uint64_t n_bits, uint64_t *units, uint64_t *passes, uint64_t *group)
{
    uint64_t dl = csgn_oracle_default_len(n_bits);
    *units = (dl + 1) / 2;
    *passes = *units / gcd_u64(*units, 256);
    *group = 256 * *passes / *units;
}

/* random word k of ciphertext c (global index) */
static uint64_t keyed_word(const uint32_t key[8], uint64_t nonce, unsigned rounds, uint64_t U,
                           uint64_t P, uint64_t Gc, uint64_t c, uint64_t k)
{
    uint64_t j = k / 2, g = c / Gc, r = (c % Gc) * U + j;
    uint64_t p = r / 256, q = (r % 256) / 64, L = r % 64;
    uint32_t x[16];
    csgn_oracle_chacha_block(key, nonce, (g * P + p) * 64 + L, rounds, x);
    uint64_t base = 4 * q + 2 * (k % 2);
    return ((uint64_t)x[base + 1] << 32) | x[base];
}

void csgn_oracle_encrypt_keyed(uint64_t n_bits, uint64_t d, const uint64_t *key_idx, uint64_t batch,
                               uint64_t first_ciphertext, const uint8_t *plain,
                               const uint32_t rng_key[8], uint64_t nonce, unsigned rounds,
                               uint64_t *out)
{
    uint64_t dl = csgn_oracle_default_len(n_bits), U, P, Gc;
    csgn_oracle_keyed_layout(n_bits, &U, &P, &Gc);
    uint64_t *mask = (uint64_t *)calloc(dl ? dl : 1, sizeof(uint64_t));
    csgn_oracle_key_mask(n_bits, key_idx, d, mask);
    unsigned rem = (unsigned)(n_bits % WORD_BITS);
    uint64_t tail = rem ? ~0ull << (WORD_BITS - rem) : ~0ull;
    uint64_t secret_bits = 0;
    for (uint64_t k = 0; k < dl; ++k)
        secret_bits += (uint64_t)__builtin_popcountll(mask[k]);
    for (uint64_t i = 0; i < batch; ++i) {
        uint64_t c = first_ciphertext + i, *o = out + i * dl;
        int all = 1;
        for (uint64_t k = 0; k < dl; ++k) {
            uint64_t v = keyed_word(rng_key, nonce, rounds, U, P, Gc, c, k);
            if (k == dl - 1)
                v &= tail;
            if ((v & mask[k]) != mask[k])
                all = 0;
            o[k] = (plain[i] & 1u) ? (v | mask[k]) : v;
        }
        if (!(plain[i] & 1u) && all && secret_bits >= 2) {
            /* all D secret positions came out 1: clear s[draw % D], draw = first word of block c of
             * the draw stream (same key and nonce, constants "csgn draw pos v1") */
            uint32_t x[16];
            chacha_block_sigma("csgn draw pos v1", rng_key, nonce, c, rounds, x);
            uint64_t pos = key_idx[((uint64_t)x[0] * d) >> 32];
            o[pos / WORD_BITS] &= ~(1ull << (WORD_BITS - 1u - (unsigned)(pos % WORD_BITS)));
        }
    }
    free(mask);
}

/* ------------------------------------------------------------------ decrypt ---- */

unsigned csgn_oracle_decrypt(uint64_t n_bits, uint64_t d, const uint64_t *key,
                             const uint64_t *v, uint64_t len, const uint64_t *bitlen)
{
    uint64_t dl = csgn_oracle_default_len(n_bits);
    uint8_t *bits = NULL;
    uint64_t total = unpack_stream(n_bits, v, len, bitlen, &bits);
    unsigned result = 0;

    if (len == dl) {
        /* src/SecretKey.cpp:97-99: single term, logical AND over the key */
        unsigned dec = 1;
        for (uint64_t i = 0; i < d; ++i)
            dec = dec && (key[i] < total ? bits[key[i]] : 0);
        result = dec;
    } else {
        /* src/SecretKey.cpp:126-140: XOR over terms of (AND over the key) */
        uint64_t terms = dl ? len / dl : 0;
        for (uint64_t k = 0; k < terms; ++k) {
            unsigned dec = 1;
            for (uint64_t i = 0; i < d; ++i) {
                uint64_t q = n_bits * k + key[i];
                dec &= (q < total ? bits[q] : 0);
            }
            result = (dec + result) % 2;
        }
    }
    free(bits);
    return result;
}

unsigned csgn_oracle_decrypt_canonical(uint64_t n_bits, uint64_t d, const uint64_t *key,
                                       const uint64_t *v, uint64_t len)
{
    uint64_t dl = csgn_oracle_default_len(n_bits);
    uint64_t terms = dl ? len / dl : 0;
    unsigned result = 0;
    for (uint64_t k = 0; k < terms; ++k) {
        const uint64_t *term = v + k * dl;
        unsigned dec = 1;
        for (uint64_t i = 0; i < d && dec; ++i)
            dec &= msb_bit(term[key[i] / WORD_BITS], (unsigned)(key[i] % WORD_BITS));
        result ^= dec;
    }
    return result;
}

/* ------------------------------------------------------------- permutations ---- */

int64_t csgn_oracle_perm_random(uint64_t size, const int32_t *draws, uint64_t n_draws,
                                uint64_t *perm)
{
    /* src/Permutation.cpp:145-156: slot i takes the first draw (mod size) not yet used.
     * Slots start at (uint64_t)-1, which never collides with a candidate. */
    uint64_t used = 0;
    for (uint64_t i = 0; i < size; ++i)
        perm[i] = UINT64_MAX;
    for (uint64_t i = 0; i < size; ++i) {
        for (;;) {
            if (used >= n_draws)
                return -1;
            uint64_t cand = (uint64_t)draws[used++] % size;
            if (!key_contains(perm, size, cand)) {
                perm[i] = cand;
                break;
            }
        }
    }
    return (int64_t)used;
}

void csgn_oracle_perm_inverse(const uint64_t *perm, uint64_t size, uint64_t *inv)
{
    /* src/Permutation.cpp:12-22: inv[i] = smallest j with perm[j] == i */
    for (uint64_t i = 0; i < size; ++i)
        inv[i] = 0;
    for (uint64_t j = size; j-- > 0;)
        if (perm[j] < size)
            inv[perm[j]] = j;
}

int csgn_oracle_perm_compose(const uint64_t *a, uint64_t len_a,
                             const uint64_t *b, uint64_t len_b, uint64_t *out)
{
    /* src/Permutation.cpp:65-73: length mismatch yields the empty permutation */
    if (len_a != len_b)
        return -1;
    for (uint64_t i = 0; i < len_a; ++i)
        out[i] = a[b[i]];
    return 0;
}

uint64_t csgn_oracle_permute_ciphertext(uint64_t n_bits, const uint64_t *perm,
                                        const uint64_t *v, uint64_t len,
                                        const uint64_t *bitlen, uint64_t *out)
{
    /* src/Ciphertext.cpp:16-69.  temp2[i] = temp[perm[i % N]] never adds the term
     * offset, and only the first N permuted bits are re-packed (:36-69), so the
     * result is the permuted FIRST term whatever the input length. */
    uint64_t dl = csgn_oracle_default_len(n_bits);
    uint8_t *bits = NULL;
    uint64_t total = unpack_stream(n_bits, v, len, bitlen, &bits);
    memset(out, 0, dl * sizeof(uint64_t));
    for (uint64_t j = 0; j < n_bits && j < total; ++j) {
        uint64_t src = perm[j];
        unsigned b = (src < total) ? bits[src] : 0u;
        out[j / WORD_BITS] |= (uint64_t)b << (WORD_BITS - 1u - (unsigned)(j % WORD_BITS));
    }
    free(bits);
    return dl;
}

uint64_t csgn_oracle_permute_key(uint64_t n_bits, const uint64_t *perm,
                                 const uint64_t *key, uint64_t d, uint64_t *new_key)
{
    /* src/SecretKey.cpp:231-250: index i belongs to the new key iff perm[i] belonged
     * to the old one; indices come out in ascending order. */
    uint8_t *member = (uint8_t *)calloc(n_bits ? n_bits : 1, 1);
    for (uint64_t i = 0; i < d; ++i)
        if (key[i] < n_bits)
            member[key[i]] = 1;
    uint64_t count = 0;
    for (uint64_t i = 0; i < n_bits; ++i)
        if (perm[i] < n_bits && member[perm[i]] && count < d)
            new_key[count++] = i;
    free(member);
    return count;
}

/* ------------------------------------------------ extension checker: compact ---- */

static uint64_t g_cmp_dl;
static const uint64_t *g_cmp_base;

static int cmp_term_index(const void *pa, const void *pb)
{
    const uint64_t a = *(const uint64_t *)pa, b = *(const uint64_t *)pb;
    int c = memcmp(g_cmp_base + a * g_cmp_dl, g_cmp_base + b * g_cmp_dl, g_cmp_dl * sizeof(uint64_t));
    if (c)
        return c;
    return (a > b) - (a < b);            /* equal terms: ascending index */
}

uint64_t csgn_oracle_compact(uint64_t dl, const uint64_t *v, uint64_t terms, uint64_t *out)
{
    if (terms == 0 || dl == 0)
        return 0;
    uint64_t *idx = (uint64_t *)malloc(terms * sizeof(uint64_t));
    uint8_t *keep = (uint8_t *)calloc(terms, 1);
    for (uint64_t i = 0; i < terms; ++i)
        idx[i] = i;
    g_cmp_dl = dl;
    g_cmp_base = v;
    qsort(idx, terms, sizeof(uint64_t), cmp_term_index);
    for (uint64_t s = 0; s < terms;) {
        uint64_t e = s + 1;
        while (e < terms && memcmp(v + idx[s] * dl, v + idx[e] * dl, dl * sizeof(uint64_t)) == 0)
            ++e;
        if ((e - s) & 1u)
            keep[idx[s]] = 1;            /* first occurrence of an odd-multiplicity term */
        s = e;
    }
    uint64_t kept = 0;
    for (uint64_t i = 0; i < terms; ++i)
        if (keep[i])
            memcpy(out + (kept++) * dl, v + i * dl, dl * sizeof(uint64_t));
    free(idx);
    free(keep);
    return kept;
}

/* ---------------------------------------------------------- harness helpers ---- */

uint64_t csgn_oracle_synth_word(uint64_t seed, uint64_t idx)
{
    return splitmix64(seed + GOLDEN * (idx + 1));
}

void csgn_oracle_synth_fill(uint64_t seed, uint64_t n_bits, uint64_t first_word,
                            uint64_t n_words, uint64_t *out)
{
    uint64_t dl = csgn_oracle_default_len(n_bits);
    unsigned rem = (unsigned)(n_bits % WORD_BITS);
    uint64_t tail_mask = rem ? ~0ull << (WORD_BITS - rem) : ~0ull;
    for (uint64_t i = 0; i < n_words; ++i) {
        uint64_t idx = first_word + i;
        uint64_t w = csgn_oracle_synth_word(seed, idx);
        if (dl && (idx % dl) == dl - 1)
            w &= tail_mask;
        out[i] = w;
    }
}

uint64_t csgn_oracle_digest(const uint64_t *w, uint64_t n_words, uint64_t first_index)
{
    uint64_t acc = 0;
    for (uint64_t i = 0; i < n_words; ++i)
        acc += splitmix64(w[i] + GOLDEN * (first_index + i + 1));
    return acc;
}
