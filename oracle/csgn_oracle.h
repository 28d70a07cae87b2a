/*
 * csgn_oracle.h -- CPU restatement of the certFHE/CSGN ciphertext-arithmetic hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (csgn_amd/, include/)
 * never links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against the
 * real reference compiled from /root/reference/src (oracle/_ref/libcsgn_ref.so, built
 * by oracle/Makefile) in tests/test_oracle_vs_ref.py, and against the committed golden
 * vectors in tests/golden/ (generated from that same reference build by
 * tests/golden/gen_golden.py) in tests/test_oracle_golden.py.
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef CSGN_ORACLE_H
#define CSGN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Context (src/Context.cpp:20-29) ------------------------------------------- */
uint64_t csgn_oracle_default_len(uint64_t n_bits);          /* ceil(N/64)            */
uint64_t csgn_oracle_context_s(uint64_t n_bits, uint64_t d); /* S = N/(2D)            */

/* Canonical bitlen side-array of a T-term ciphertext: 64,...,64,(N%64) per term
 * (src/SecretKey.cpp:171-173).  When N%64==0 the reference writes one word past the
 * end; the restatement simply emits 64 for every word. */
void csgn_oracle_bitlen(uint64_t n_bits, uint64_t terms, uint64_t *bitlen);

/* ---- multiply (src/Ciphertext.cpp:124-179) ---------------------------------------
 * len1/len2 are in 64-bit words.  out/bitlen_out must hold csgn_oracle_mul_len() words.
 * bitlen_in1 / bitlen_out may be NULL (then the bitlen pass is skipped).
 * Returns newlen. */
uint64_t csgn_oracle_mul_len(uint64_t dl, uint64_t len1, uint64_t len2);
uint64_t csgn_oracle_mul(uint64_t dl,
                         const uint64_t *c1, uint64_t len1, const uint64_t *bitlen_in1,
                         const uint64_t *c2, uint64_t len2,
                         uint64_t *out, uint64_t *bitlen_out);

/* Same product, but with the cost structure of Ciphertext::operator*
 * (src/Ciphertext.cpp:231-247): fresh result + bitlen buffers, values pass, bitlen pass,
 * then the deep copy done by the 4-arg constructor (src/Ciphertext.cpp:344-358), then
 * frees.  Used as the single-core "port" CPU baseline.  Returns a 64-bit digest of the
 * product so the work cannot be optimised away. */
uint64_t csgn_oracle_mul_reference_cost(uint64_t dl,
                                        const uint64_t *c1, uint64_t len1,
                                        const uint64_t *c2, uint64_t len2);

/* ---- add = concatenation (src/Ciphertext.cpp:107-122, 204-229) ------------------- */
uint64_t csgn_oracle_add(const uint64_t *c1, uint64_t len1, const uint64_t *bitlen1,
                         const uint64_t *c2, uint64_t len2, const uint64_t *bitlen2,
                         uint64_t *out, uint64_t *bitlen_out);

/* ---- key generation (src/SecretKey.cpp:308-337) ----------------------------------
 * draws[] are successive rand() results.  Returns the number consumed, or -1 if the
 * draws ran out. */
int64_t csgn_oracle_keygen(uint64_t n_bits, uint64_t d,
                           const int32_t *draws, uint64_t n_draws, uint64_t *key);

/* dL-word MSB-first bitmask with bit s[i] set for every key index. */
void csgn_oracle_key_mask(uint64_t n_bits, const uint64_t *key, uint64_t d, uint64_t *mask);

/* ---- encrypt (src/SecretKey.cpp:35-80 bit vector, 153-206 packing) ---------------
 * draws[] are the successive rand() results the reference would obtain.  out receives
 * dL packed words.  Returns the number of draws consumed (N-D.. for bit 1, N or N+1 for
 * bit 0), or -1 if the draws ran out. */
int64_t csgn_oracle_encrypt(uint64_t n_bits, uint64_t d, const uint64_t *key,
                            unsigned bit, const int32_t *draws, uint64_t n_draws,
                            uint64_t *out);

/* ---- keyed device generator: definitions shared with the HIP side (NOT reference code) ----
 * ChaCha block function (Bernstein 2008; 64-bit counter, 64-bit nonce), the seed expansion of
 * csgn_rng_from_seed, the keystream layout and the whole of csgn_encrypt_keyed
 * (include/csgn_hip.h), restated independently of the kernels. */
void csgn_oracle_chacha_block(const uint32_t key[8], uint64_t nonce, uint64_t counter,
                              unsigned rounds, uint32_t out[16]);
/* key of a circuit encrypt node (csgn_circuit_node_key of include/csgn_hip.h): words 0..7 of the
 * ChaCha20 block with constants "csgn node key v1", counter 0 */
void csgn_oracle_node_key(const uint32_t key[8], uint64_t nonce, uint32_t node_key[8]);
void csgn_oracle_rng_from_seed(uint64_t seed, uint32_t key[8], uint64_t *nonce);
void csgn_oracle_keyed_layout(uint64_t n_bits, uint64_t *units, uint64_t *passes, uint64_t *group);
void csgn_oracle_encrypt_keyed(uint64_t n_bits, uint64_t d, const uint64_t *key_idx, uint64_t batch,
                               uint64_t first_ciphertext, const uint8_t *plain,
                               const uint32_t rng_key[8], uint64_t nonce, unsigned rounds,
                               uint64_t *out);

/* ---- decrypt (src/SecretKey.cpp:82-147) -------------------------------------------
 * Faithful form: unpacks the stream bit by bit according to bitlen[], then applies the
 * AND-over-key / XOR-over-terms reduction.  bitlen may be NULL => canonical. */
unsigned csgn_oracle_decrypt(uint64_t n_bits, uint64_t d, const uint64_t *key,
                             const uint64_t *v, uint64_t len, const uint64_t *bitlen);
/* Same answer for canonical bitlen without the unpacked scratch copy. */
unsigned csgn_oracle_decrypt_canonical(uint64_t n_bits, uint64_t d, const uint64_t *key,
                                       const uint64_t *v, uint64_t len);

/* ---- permutations (src/Permutation.cpp, src/Ciphertext.cpp:7-82,
 *      src/SecretKey.cpp:226-259) --------------------------------------------------- */
int64_t csgn_oracle_perm_random(uint64_t size, const int32_t *draws, uint64_t n_draws,
                                uint64_t *perm);                 /* Permutation.cpp:139-157 */
void csgn_oracle_perm_inverse(const uint64_t *perm, uint64_t size, uint64_t *inv); /* :8-27  */
int  csgn_oracle_perm_compose(const uint64_t *a, uint64_t len_a,
                              const uint64_t *b, uint64_t len_b, uint64_t *out);  /* :63-78 */
/* Ciphertext permutation with the reference's multi-term truncation: the
 * result is always ONE term, the permuted first term.  Returns the output length (dL). */
uint64_t csgn_oracle_permute_ciphertext(uint64_t n_bits, const uint64_t *perm,
                                        const uint64_t *v, uint64_t len,
                                        const uint64_t *bitlen, uint64_t *out);
/* Key permutation; new_key is sorted ascending.  Returns the number of indices written. */
uint64_t csgn_oracle_permute_key(uint64_t n_bits, const uint64_t *perm,
                                 const uint64_t *key, uint64_t d, uint64_t *new_key);

/* ---- EXTENSION checker (not reference behaviour) ------------------------
 * Mod-2 compaction of a term list: a term that occurs an even number of times vanishes, one
 * that occurs an odd number of times is kept once, at the position of its first occurrence.
 * Decryption is unchanged for every key (XOR over terms).  Returns the number of terms kept;
 * out must hold terms*dL words. */
uint64_t csgn_oracle_compact(uint64_t dl, const uint64_t *v, uint64_t terms, uint64_t *out);

/* ---- harness helpers shared with the HIP side (definitions, not reference code) ---
 * Synthetic operand words: word idx of a flat term buffer is
 * splitmix64(seed + GOLDEN*(idx+1)); the last word of every term keeps only its top
 * N%64 bits. */
uint64_t csgn_oracle_synth_word(uint64_t seed, uint64_t idx);
void csgn_oracle_synth_fill(uint64_t seed, uint64_t n_bits, uint64_t first_word,
                            uint64_t n_words, uint64_t *out);
/* Order-sensitive 64-bit digest: sum over i of splitmix64(w[i] + GOLDEN*(first+i+1)). */
uint64_t csgn_oracle_digest(const uint64_t *w, uint64_t n_words, uint64_t first_index);

#ifdef __cplusplus
}
#endif
#endif
