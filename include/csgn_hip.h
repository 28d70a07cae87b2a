/*
 * csgn_hip.h -- C ABI of libcsgn_hip.so: the MI355X (gfx950) implementation of the
 * certFHE/CSGN ciphertext-arithmetic hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI layer; its seam is the set of
 * six private array functions behind the certFHE:: classes:
 *
 *   Ciphertext::defaultN_multiply   /root/reference/src/Ciphertext.cpp:124-131
 *   Ciphertext::multiply            /root/reference/src/Ciphertext.cpp:133-179
 *   Ciphertext::add                 /root/reference/src/Ciphertext.cpp:107-122
 *   SecretKey::encrypt(bit,n,d,s)   /root/reference/src/SecretKey.cpp:35-80   (+ packing :153-206)
 *   SecretKey::defaultN_decrypt     /root/reference/src/SecretKey.cpp:82-102
 *   SecretKey::decrypt(v,len,..)    /root/reference/src/SecretKey.cpp:104-147
 *
 * Each entry point below names the one(s) it replaces.  The certFHE:: C++ classes in
 * include/certfhe/ are implemented on top of exactly these symbols; INTEGRATION.md shows
 * the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - POD arguments only.  `const uint64_t *d_x` is a DEVICE pointer (HBM); `h_x` is a host
 *     pointer.  Nothing here allocates memory the caller must free with delete[].
 *   - Term buffers: a ciphertext of T terms at N bits is T*dL consecutive uint64 words,
 *     dL = ceil(N/64); bit j of a term is word j/64, bit 63-(j%64) (MSB first); the unused
 *     low bits of a term's last word are zero (src/Ciphertext.h:19-21, SecretKey.cpp:175-197).
 *     The reference's parallel `bitlen` array is a pure function of (N, T) for every
 *     ciphertext produced by encrypt/+/x and is NOT materialised on the device; see
 *     csgn_bitlen_canonical().
 *   - Batches: "uniform" = B ciphertexts of the same term count laid back to back;
 *     "ragged" = CSR: d_off[B+1] term offsets into one flat term buffer (empty ciphertexts
 *     allowed).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Compute calls
 *     are asynchronous on that stream; the caller synchronises.
 *   - Every function returns CSGN_OK (0) or a negative csgn_status; csgn_last_error() gives
 *     a thread-local message.  The reference has no error convention at all;
 *     misuse that is UB there is a reported error here.
 *   - Thread safety: no hidden global state except per-thread items (the error string and the
 *     ragged planner's small device scratch); one host thread (or process) per GPU may call
 *     concurrently.
 *   - There is NO CPU fallback: without a gfx950 device every compute call fails with
 *     CSGN_ERR_NO_DEVICE.
 */
#ifndef CSGN_HIP_H
#define CSGN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum csgn_status {
    CSGN_OK = 0,
    CSGN_ERR_INVALID = -1,      /* bad argument (null pointer, N==0, key index >= N, ...) */
    CSGN_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels handle (see each call) */
    CSGN_ERR_NO_DEVICE = -3,    /* no HIP device / not gfx950 */
    CSGN_ERR_HIP = -4           /* a HIP runtime call failed; message has the detail */
} csgn_status;

#define CSGN_ABI_VERSION 1

/* ------------------------------------------------------------------ runtime ---- */

int csgn_abi_version(void);
const char *csgn_last_error(void);

/* Select `device` for the calling thread and verify it is a gfx950 part. */
int csgn_init(int device);
int csgn_device_count(int *h_count);
/* Fills h_name (<= cap bytes) with the gcnArchName and reports CU count / HBM bytes. */
int csgn_device_info(int device, char *h_name, size_t cap, int *h_cu_count, uint64_t *h_hbm_bytes);

int csgn_malloc(void **d_ptr, size_t bytes);
int csgn_free(void *d_ptr);
/* Pinned host memory the GPU can address: *h_ptr for the host, *d_alias for kernels (results
 * a kernel writes there are visible to the host after csgn_stream_sync, no copy).  Used by the
 * C++ classes for the one-byte answer of SecretKey::decrypt. */
int csgn_host_alloc(void **h_ptr, void **d_alias, size_t bytes);
int csgn_host_free(void *h_ptr);
/* Host-buffer lifetime: h_src / h_dst belong to the call until `stream` has passed the copy.  From or to
 * pageable memory the runtime happens to stage the copy before it returns, but that is not part of this
 * contract: synchronise the stream (or use csgn_host_alloc memory that outlives it) before freeing or
 * reusing the buffer. */
int csgn_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int csgn_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int csgn_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes, void *stream);
int csgn_memset(void *d_dst, int value, size_t bytes, void *stream);
int csgn_stream_create(void **stream);
int csgn_stream_destroy(void *stream);
int csgn_stream_sync(void *stream);
int csgn_event_create(void **event);
int csgn_event_destroy(void *event);
int csgn_event_record(void *event, void *stream);
int csgn_event_elapsed_ms(void *start, void *stop, float *h_ms);   /* synchronises on stop */
int csgn_event_sync(void *event);                                  /* returns once the work recorded in front of the event has run */

/* ------------------------------------------------- host-side metadata helpers ---- */

/* Context::getDefaultN, src/Context.cpp:24-28. */
uint64_t csgn_default_len(uint64_t n_bits);
/* Context S = N/(2D), src/Context.cpp:22. */
uint64_t csgn_context_s(uint64_t n_bits, uint64_t d);
/* Result length in words of Ciphertext::multiply, src/Ciphertext.cpp:135-146. */
uint64_t csgn_mul_len(uint64_t n_bits, uint64_t len1, uint64_t len2);
/* The bitlen side-array the reference would hold for a T-term ciphertext
 * (src/SecretKey.cpp:171-173 per term): h_bitlen[T*dL]. */
int csgn_bitlen_canonical(uint64_t n_bits, uint64_t terms, uint64_t *h_bitlen);
/* Pack a secret key (D indices in [0,N), src/SecretKey.h:22) into the dL-word MSB-first
 * mask the decrypt/encrypt kernels consume.  Duplicate indices are allowed (setKey does
 * not forbid them, src/SecretKey.cpp:292-302); an index >= N is CSGN_ERR_INVALID. */
int csgn_key_mask(uint64_t n_bits, const uint64_t *h_key, uint64_t d, uint64_t *h_mask);

/* ------------------------------------------------------------------ multiply ---- */

/* Batched Ciphertext::multiply (src/Ciphertext.cpp:133-179) incl. the 1x1 fast path
 * defaultN_multiply (:124-131).  For every pair b < batch:
 *     out_b[(i*t2 + j)*dL + k] = L_b[i*dL + k] & R_b[j*dL + k]      i<t1, j<t2, k<dL
 * d_left: batch*t1*dL words, d_right: batch*t2*dL, d_out: batch*t1*t2*dL.
 * Pair p's product goes to slot (p % out_slots) of d_out when out_slots != 0 (streaming a
 * batch through a fixed arena, SURVEY 8d "streaming rule"); out_slots == 0 means one slot
 * per pair.  With out_slots < batch the call is split into launches of <= out_slots pairs
 * in stream order so later pairs overwrite earlier ones deterministically.
 * Limits: dL*8 <= 16384 bytes per term; t1*t2*dL < 2^32 per pair.
 * The kernel is chosen per shape (csgn_mul_uniform_kernel names it): calls with >= 4 MB of
 * operands whose output is >= 4x the operands are preceded by a read-only pass over the
 * operands that leaves them in the GPU's memory-side cache, so the operands are READ twice;
 * nothing but d_out is written. */
int csgn_mul_uniform(uint64_t n_bits, uint64_t batch, uint64_t t1, uint64_t t2,
                     const uint64_t *d_left, const uint64_t *d_right, uint64_t *d_out,
                     uint64_t out_slots, void *stream);

/* Ragged form.  Step 1 (plan): from the operand term offsets compute the product term
 * offsets d_off_out[batch+1] (exclusive scan of t1_b*t2_b) on the device and return
 * h_plan[0] = total output terms, h_plan[1] = max t1, h_plan[2] = max t2,
 * h_plan[3] = max t1*t2.  Synchronises `stream`.  d_off_out doubles as the per-pair result
 * term counts the multi-GPU driver gathers (count_b = off[b+1]-off[b]). */
int csgn_mul_ragged_plan(uint64_t batch, const uint64_t *d_off_left, const uint64_t *d_off_right,
                         uint64_t *d_off_out, uint64_t h_plan[4], void *stream);
/* Step 2: the products, into d_out[h_plan[0]*dL] at the planned offsets; pass the plan's
 * max_t1 = h_plan[1], max_t2 = h_plan[2], total_out_terms = h_plan[0].  Nearly uniform batches
 * of large products run the LDS-tiled kernel, and so do batches of small pairs whose largest shape is
 * small too (one narrow workgroup per pair); a batch whose pairs ALL have the largest shape runs
 * the uniform kernels; skewed or small ones a flat kernel whose grid is the real output (a workgroup
 * finds its first pair by a 64-ary search over d_off_out and stages the offsets it needs in LDS).
 * A pure function of its arguments: it keeps no state from the plan call (round 3 did, per host thread;
 * what a plan knows beyond its four numbers now lives in a csgn_mul_plan object, below).  Through this
 * entry a skewed batch's huge pairs take the CSR kernel like everything else and a product above 1 GiB is
 * written in 1 GiB slices. */
int csgn_mul_ragged(uint64_t n_bits, uint64_t batch,
                    const uint64_t *d_left, const uint64_t *d_off_left,
                    const uint64_t *d_right, const uint64_t *d_off_right,
                    uint64_t *d_out, const uint64_t *d_off_out,
                    uint64_t max_t1, uint64_t max_t2, uint64_t total_out_terms, void *stream);

/* The same two steps with the plan kept in an OBJECT of the caller's (round 4; supersedes the round-3 rule that
 * csgn_mul_ragged found the last plan of the calling thread by array addresses).  csgn_mul_plan_ragged is
 * csgn_mul_ragged_plan and additionally notes in *plan the batch's HUGE pairs (24 MB of output and more, up to
 * 32 of them), its operand size and a checksum of the three offset arrays; csgn_mul_planned multiplies by that
 * plan: a huge pair gets a uniform launch of its own when such pairs are most of the batch, a product above
 * 1 GiB is written in slices sized from the operand size.  The offset arrays belong to the plan until the next
 * csgn_mul_plan_ragged on it.  Because a huge pair's launch uses host copies of its offsets, csgn_mul_planned
 * first CHECKS the offset arrays against the plan's checksum whenever it is about to use such records (one
 * small kernel and a stream synchronise) and returns CSGN_ERR_INVALID if they changed;
 * csgn_mul_plan_trust(plan, 1) turns that check off for a caller who guarantees it.
 * csgn_mul_plan_validate does the check on demand.  A plan is used by one host thread at a time. */
typedef struct csgn_mul_plan csgn_mul_plan;
int csgn_mul_plan_create(csgn_mul_plan **plan);
void csgn_mul_plan_destroy(csgn_mul_plan *plan);
int csgn_mul_plan_ragged(csgn_mul_plan *plan, uint64_t batch, const uint64_t *d_off_left,
                         const uint64_t *d_off_right, uint64_t *d_off_out, uint64_t h_plan[4], void *stream);
int csgn_mul_planned(csgn_mul_plan *plan, uint64_t n_bits, const uint64_t *d_left, const uint64_t *d_right,
                     uint64_t *d_out, void *stream);
int csgn_mul_plan_validate(csgn_mul_plan *plan, void *stream);     /* CSGN_ERR_INVALID: offsets changed since the plan */
int csgn_mul_plan_trust(csgn_mul_plan *plan, int trust);

/* Ragged multiply with NO host round trip: the plan kernel and the multiply are enqueued back to back and
 * nothing is read back.  d_off_out[batch+1] is written as by csgn_mul_ragged_plan.  The caller gives the room:
 * out_capacity_terms = terms d_out can hold (an upper bound on the sum of t1_b*t2_b, e.g. from static shapes);
 * the launch is sized for it and stops at the real end, which only the device knows.  If the products do
 * not fit, NOTHING is written and the result's fifth word says so.  d_plan: device block of
 * csgn_mul_ragged_async_plan_words(batch) words, valid until the stream has passed the call.
 * Kernels: the CSR kernel, or -- decided on the device -- the plain AND stream when every pair is 1 x 1; the
 * LDS-tiled and per-huge-pair forms need shapes on the host and belong to csgn_mul_planned.
 * csgn_mul_ragged_async_result (optional, synchronises): h_result[0..3] as h_plan of csgn_mul_ragged_plan,
 * h_result[4] = 1 if the products did not fit. */
uint64_t csgn_mul_ragged_async_plan_words(uint64_t batch);
int csgn_mul_ragged_async(uint64_t n_bits, uint64_t batch,
                          const uint64_t *d_left, const uint64_t *d_off_left,
                          const uint64_t *d_right, const uint64_t *d_off_right,
                          uint64_t *d_out, uint64_t *d_off_out, uint64_t out_capacity_terms,
                          uint64_t *d_plan, void *stream);
int csgn_mul_ragged_async_result(const uint64_t *d_plan, uint64_t h_result[5], void *stream);

/* ----------------------------------------------------------------------- add ---- */

/* Batched Ciphertext::add (src/Ciphertext.cpp:107-122): out_b = L_b || R_b, t1+t2 terms.
 * No XOR, no de-duplication -- exactly the reference. */
int csgn_add_uniform(uint64_t n_bits, uint64_t batch, uint64_t t1, uint64_t t2,
                     const uint64_t *d_left, const uint64_t *d_right, uint64_t *d_out,
                     void *stream);
/* Ragged: d_off_out[b] = d_off_left[b] + d_off_right[b] is written by the call;
 * total_terms_out = d_off_left[batch] + d_off_right[batch] (the caller sized d_out with it). */
int csgn_add_ragged(uint64_t n_bits, uint64_t batch,
                    const uint64_t *d_left, const uint64_t *d_off_left,
                    const uint64_t *d_right, const uint64_t *d_off_right,
                    uint64_t *d_out, uint64_t *d_off_out,
                    uint64_t total_terms_out, void *stream);
/* The same with the caller's bounds on the term counts of one element of either operand (0, 0 = unknown).  The
 * offsets are device data; the bounds are how a caller that knows its shapes -- a class layer that built the
 * batch, a circuit -- tells the dispatch: with batch * (max_t1 + max_t2) == total_terms_out every pair has exactly
 * max_t1 + max_t2 terms, no lane has to find its pair and the uniform kernel runs (a million 1+1 sums given as CSR:
 * 73 % of the HBM peak against 55 % through the CSR kernel).  d_off_out is written either way.  Bounds too small
 * for total_terms_out are refused; bounds that do not hold (an element with more terms than stated) are the caller's
 * error, as wrong offsets would be. */
int csgn_add_ragged_bounded(uint64_t n_bits, uint64_t batch, uint64_t max_t1, uint64_t max_t2,
                            const uint64_t *d_left, const uint64_t *d_off_left,
                            const uint64_t *d_right, const uint64_t *d_off_right,
                            uint64_t *d_out, uint64_t *d_off_out,
                            uint64_t total_terms_out, void *stream);

/* A LIST of small, independent operations in one launch (round 5).  BASELINE config 1 is single operations on
 * one- and two-term ciphertexts behind a value-semantic API (tests/basic_operations.cpp:26-40): 480 bytes of traffic
 * and 2-3 us of launch each when issued one by one, 10-20 x the reference's 0.12-0.28 us.  A caller that can QUEUE
 * such operations (the class layer does: csgn_amd/csrc/certfhe/runtime.cpp) hands the queue over as records
 *     out = left + right (kind 0: concatenation, src/Ciphertext.cpp:107-122) or left * right (kind 1: all-pairs AND,
 *     :146-163), t1 / t2 terms a side, one workgroup per record.
 * d_ops must be readable by the device: device memory, or pinned host memory through its device alias
 * (csgn_host_alloc) -- the records are then read over the link, no copy is enqueued.  The operations of ONE call
 * must not read one another's outputs (split dependent ones over calls: the stream orders them); the record array
 * must stay untouched until the launch has run.  Meant for small shapes (a workgroup walks its output): use the
 * uniform / ragged calls for anything large.  Words are those of csgn_add_uniform / csgn_mul_uniform. */
typedef struct csgn_small_op {
    const uint64_t *left;
    const uint64_t *right;
    uint64_t *out;
    uint32_t t1, t2;
    uint32_t kind;           /* 0 add, 1 multiply */
    uint32_t reserved;
} csgn_small_op;
int csgn_small_ops(uint64_t n_bits, uint64_t count, const csgn_small_op *d_ops, void *stream);

/* ------------------------------------------------------------------- decrypt ---- */

/* Batched SecretKey::decrypt (src/SecretKey.cpp:104-147; single term :82-102):
 *     bit_b = XOR over terms k of ( AND over key indices s of term_k[s] )
 * d_mask is the dL-word key mask (csgn_key_mask) in device memory; d_bits receives one
 * byte (0/1) per ciphertext.  d_scratch must hold csgn_decrypt_scratch_bytes(batch, total
 * terms) bytes (one hit bit per term + one partial-parity word per ciphertext).  An empty
 * ciphertext decrypts to 0, as in the reference. */
size_t csgn_decrypt_scratch_bytes(uint64_t batch, uint64_t total_terms);
int csgn_decrypt_uniform(uint64_t n_bits, uint64_t batch, uint64_t terms,
                         const uint64_t *d_terms, const uint64_t *d_mask,
                         uint8_t *d_bits, void *d_scratch, void *stream);
int csgn_decrypt_ragged(uint64_t n_bits, uint64_t batch, uint64_t total_terms,
                        const uint64_t *d_terms, const uint64_t *d_off, const uint64_t *d_mask,
                        uint8_t *d_bits, void *d_scratch, void *stream);
/* The same with what the caller knows about the shapes: max_terms = an upper bound on the terms of any ONE
 * ciphertext (0 = unknown: csgn_decrypt_ragged).  The offsets live on the device, so the bound is the only way the
 * dispatch can learn what a class layer or a circuit knows anyway: with batch * max_terms == total_terms every
 * ciphertext has exactly max_terms terms and the uniform kernels run (a million single-term ciphertexts handed over as
 * CSR: one kernel that writes the plaintext bytes itself, 76 % of the HBM peak against 60 % through the CSR passes); with
 * max_terms <= 4096 the launch that folds long ciphertexts chunk by chunk is not made.  A bound that is too small for
 * total_terms is refused (CSGN_ERR_INVALID); one that is merely not tight costs nothing but the shortcut; one that
 * does not hold (a ciphertext with more terms than stated) is the caller's error, as wrong offsets would be. */
int csgn_decrypt_ragged_bounded(uint64_t n_bits, uint64_t batch, uint64_t total_terms, uint64_t max_terms,
                                const uint64_t *d_terms, const uint64_t *d_off, const uint64_t *d_mask,
                                uint8_t *d_bits, void *d_scratch, void *stream);

/* Fused forms: the plaintext of a product / sum WITHOUT materialising it.
 *     bit_b = Dec(L_b * R_b) = Dec(L_b) & Dec(R_b)        (product)
 *     bit_b = Dec(L_b + R_b) = Dec(L_b) ^ Dec(R_b)        (sum)
 * exact identities of the scheme: a product term L_i & R_j has all D secret positions set
 * iff both factors do, so the count of such terms is hits(L)*hits(R); concatenation adds the
 * counts (src/SecretKey.cpp:131-140 applied to src/Ciphertext.cpp:153-163 / :107-122).
 * Reads 8*dL*(t1+t2) bytes per pair instead of writing 8*dL*t1*t2.  d_scratch must hold
 * csgn_decrypt_combined_scratch_bytes(batch, t1, t2) bytes. */
size_t csgn_decrypt_combined_scratch_bytes(uint64_t batch, uint64_t t1, uint64_t t2);
int csgn_decrypt_product_uniform(uint64_t n_bits, uint64_t batch, uint64_t t1, uint64_t t2,
                                 const uint64_t *d_left, const uint64_t *d_right,
                                 const uint64_t *d_mask, uint8_t *d_bits, void *d_scratch, void *stream);
int csgn_decrypt_sum_uniform(uint64_t n_bits, uint64_t batch, uint64_t t1, uint64_t t2,
                             const uint64_t *d_left, const uint64_t *d_right,
                             const uint64_t *d_mask, uint8_t *d_bits, void *d_scratch, void *stream);

/* EXTENSION, not reference behaviour: mod-2 compaction of term lists.  The
 * reference's add never reduces (src/Ciphertext.cpp:107-122); because decryption XORs over
 * terms (src/SecretKey.cpp:139), identical terms cancel in pairs, so every ciphertext may be
 * replaced by its distinct odd-multiplicity terms without changing Dec under ANY key.  Output
 * keeps one copy of each such term at the position order of first occurrence; d_off_out
 * receives the compacted CSR offsets (d_off_out[batch] = terms kept).  d_out needs room for
 * total_terms*dL words and must not overlap d_terms; d_scratch needs
 * csgn_compact_scratch_bytes(n_bits, batch, total_terms) bytes.
 * max_terms: an upper bound on the term count of any ONE ciphertext of the batch when the caller
 * knows it, 0 = unknown.  It is a launch hint only: a ciphertext up to one workgroup's group (1024
 * terms at N=1247, 320 at N=4096) is read once and deduplicated in LDS; with a bound of up to 1792 / 768
 * terms a wide build of the same kernel (48 units per lane, one workgroup per CU) does the same for the
 * whole batch; larger ciphertexts are deduplicated by hash partitions (terms read twice; an exact table
 * in HBM behind a partition overflow or a hash collision) whose kernels are skipped when the bound rules
 * them out; a bound under half a group lets runs of small ciphertexts fill their groups.  A ciphertext
 * that exceeds a non-zero group-sized bound it was promised to respect is copied through uncompacted
 * (still a legal result); a broken small bound sends the call back to the plan it would have had without one.
 * Bit-exact for every input: terms are matched by a hash (48-bit tags inside a group, 64 bits between the
 * chunks of a larger ciphertext) first and then compared in full; a collision between unequal terms only
 * costs time.
 * Limits: fewer than 2^31 ciphertexts and terms per call (CSGN_ERR_UNSUPPORTED).
 * Never called on a parity path. */
size_t csgn_compact_scratch_bytes(uint64_t n_bits, uint64_t batch, uint64_t total_terms);
int csgn_compact_ragged(uint64_t n_bits, uint64_t batch, uint64_t total_terms, uint64_t max_terms,
                        const uint64_t *d_terms, const uint64_t *d_off,
                        uint64_t *d_out, uint64_t *d_off_out, void *d_scratch, void *stream);

/* ------------------------------------------------------------------- encrypt ---- */

/* Batched SecretKey::encrypt (bit vector src/SecretKey.cpp:35-80, MSB-first packing
 * :153-206) with the randomness made an explicit argument (hard part 3).
 * For ciphertext b:
 *   d_rnd[b*dL ..]  the value rand()%2 the reference would have stored at each position,
 *                   packed MSB-first (forced positions are ignored);
 *   d_chosen[b]     for plaintext 0: the secret POSITION s[rand()%d] picked at :51;
 *   d_last[b]       for plaintext 0: the final rand()%2 of :76 (used only when the other
 *                   secret positions are not all 1).
 * bit 1: out = rnd | mask.   bit 0: out = rnd with position `chosen` replaced by
 * (all other secret positions are 1) ? 0 : last -- for D==1 the reference never clears it
 * (its `v` stays 0), and neither does this.  Padding bits of the last word are cleared.
 * The certFHE::SecretKey class maps the glibc rand() stream onto these arrays, which makes
 * the device result bit-identical to the reference under the same srand(). */
int csgn_encrypt_explicit(uint64_t n_bits, uint64_t d, uint64_t batch,
                          const uint8_t *d_plain, const uint64_t *d_rnd,
                          const uint32_t *d_chosen, const uint8_t *d_last,
                          const uint64_t *d_mask, uint64_t *d_out, void *stream);
/* Throughput form: same construction, randomness generated on the device by a KEYED generator:
 * ChaCha (64-bit block counter, 64-bit nonce; 8, 12 or 20 rounds) in counter mode under a 256-bit
 * secret key.  Outputs do not reveal the key or one another (a ciphertext word that carries no
 * secret position IS raw generator output, so an invertible generator would leak the stream and
 * with it the secret positions).  Same distribution as the reference, not the same bits: every
 * position is drawn; plaintext 1 ORs the key mask in; for plaintext 0, if all D secret positions
 * came out 1, position s[draw % D] is cleared (equivalent to src/SecretKey.cpp:51-76: draw the
 * position first, force it to 0 when all the others are 1) -- unless the key has a single distinct
 * position, which the reference never clears either.  `draw` comes from a second stream of the same
 * (key, nonce) whose ChaCha constants are "csgn draw pos v1" instead of "expand 32-byte k": no nonce
 * makes it coincide with a keystream.
 *   h_rng              key/nonce/rounds (host struct; fill with csgn_rng_from_os for real use)
 *   first_ciphertext   GLOBAL index of d_plain[0] / d_out[0] in the (key, nonce) stream: ciphertext
 *                      c always draws the same words whatever batch or shard it is encrypted in
 *                      (csgn_shard.h).  Never encrypt two different plaintexts under the same
 *                      (key, nonce, index): advance first_ciphertext or change the nonce.
 *   d_key              the D secret indices (device), d_mask their dL-word mask (csgn_key_mask).
 * Keystream layout (csgn_encrypt_keyed_layout reports U, P, Gc): U = ceil(dL/2) 16-byte units per
 * ciphertext, P = U/gcd(U,256), Gc = 256*P/U; unit j of ciphertext c is words 4q..4q+3 of ChaCha
 * block (g*P + p)*64 + L with g = c/Gc, r = (c%Gc)*U + j, p = r/256, q = (r%256)/64, L = r%64. */
typedef struct csgn_rng {
    uint32_t key[8];      /* 256-bit generator key: SECRET */
    uint64_t nonce;       /* stream id */
    uint32_t rounds;      /* 8, 12 or 20 */
    uint32_t reserved;
} csgn_rng;
/* key and nonce from the operating system's entropy source (getrandom). */
int csgn_rng_from_os(csgn_rng *h_rng, uint32_t rounds);
/* REPRODUCIBLE stream for tests and benchmarks: key and nonce expanded from a 64-bit seed.  Sixty-four
 * bits of entropy at most -- not for ciphertexts that have to stay secret. */
int csgn_rng_from_seed(csgn_rng *h_rng, uint64_t seed, uint32_t rounds);
int csgn_encrypt_keyed_layout(uint64_t n_bits, uint32_t *h_units, uint32_t *h_passes, uint32_t *h_group);
int csgn_encrypt_keyed(uint64_t n_bits, uint64_t d, uint64_t batch, uint64_t first_ciphertext,
                       const uint8_t *d_plain, const uint64_t *d_key, const uint64_t *d_mask,
                       const csgn_rng *h_rng, uint64_t *d_out, void *stream);
/* FUSED FRESH CHAIN (the reference's canonical flow tests/basic_operations.cpp:26-40:
 * encrypt, encrypt, operator*, decrypt).  For every pair b < batch
 *     d_out_b = Enc_A(d_plain_a[b]) & Enc_B(d_plain_b[b])          one term, dL words
 * where Enc_A / Enc_B are EXACTLY the ciphertexts csgn_encrypt_keyed(..., h_rng_a / h_rng_b, ...)
 * writes for position first_ciphertext + b and & is Ciphertext::defaultN_multiply
 * (src/Ciphertext.cpp:124-131).  One kernel: both operands live in registers only and the product is
 * written once -- 8*dL bytes per pair instead of the five HBM passes of encrypt, encrypt, multiply.
 * d_bits (optional, may be NULL) receives Dec(d_out_b) under the same key, one byte per pair, computed
 * from the generated words (a 1x1 product decrypts to 1 iff both factors cover the key mask,
 * src/SecretKey.cpp:82-102).  The two generators must differ in key or nonce. */
int csgn_encrypt_mul_keyed(uint64_t n_bits, uint64_t d, uint64_t batch, uint64_t first_ciphertext,
                           const uint8_t *d_plain_a, const uint8_t *d_plain_b, const uint64_t *d_key,
                           const uint64_t *d_mask, const csgn_rng *h_rng_a, const csgn_rng *h_rng_b,
                           uint64_t *d_out, uint8_t *d_bits, void *stream);
/* = csgn_encrypt_keyed with csgn_rng_from_seed(seed, 8 rounds) and first_ciphertext 0: the
 * reproducible test/benchmark form (see csgn_rng_from_seed: NOT for secrets). */
int csgn_encrypt_device_rng(uint64_t n_bits, uint64_t d, uint64_t batch,
                            const uint8_t *d_plain, const uint64_t *d_key,
                            const uint64_t *d_mask, uint64_t seed, uint64_t *d_out, void *stream);

/* --------------------------------------------------------------- permutation ---- */

/* Batched Ciphertext::applyPermutation (src/Ciphertext.cpp:7-82): new bit j = old bit
 * perm[j].  d_perm: N uint32 indices.  The reference collapses a multi-term ciphertext to
 * its permuted FIRST term; `per_term` = 0 reproduces that (d_out: batch*dL
 * words, input stride terms_in*dL), `per_term` = 1 permutes every term (extension,
 * d_out: batch*terms_in*dL). */
int csgn_permute_uniform(uint64_t n_bits, uint64_t batch, uint64_t terms_in, int per_term,
                         const uint64_t *d_terms, const uint32_t *d_perm, uint64_t *d_out,
                         void *stream);

/* ------------------------------------------- explicit bitlen (one ciphertext) ---- */

/* A ciphertext built through the reference's 4-argument constructor / setBitlen may carry ANY
 * bitlen side array; the reference then reads (v, bitlen) as a bit stream -- word i contributes its
 * top bitlen[i] bits -- and addresses it at flat positions (src/SecretKey.cpp:104-147,
 * src/Ciphertext.cpp:16-69).  These two calls do exactly that for ONE ciphertext of len_words words,
 * with d_bitlen[len_words] on the device (values above 64 are read as 64; a position past the end of
 * the stream reads 0):
 *   decrypt:  *d_bit = XOR over k < len_words/dL of AND over i < d of stream[n*k + key[i]]
 *             (d_key: the D indices themselves, not the mask -- positions are stream positions);
 *   permute:  d_out[dL words]: new bit j = stream[perm[j]] for j < min(N, stream length), the rest 0
 *             (as in the reference the result is ONE term).
 * d_scratch: csgn_bitlen_scratch_bytes(len_words).  With the canonical pattern both agree with
 * csgn_decrypt_uniform / csgn_permute_uniform, which are the fast paths. */
size_t csgn_bitlen_scratch_bytes(uint64_t len_words);
int csgn_decrypt_bitlen(uint64_t n_bits, uint64_t d, uint64_t len_words, const uint64_t *d_v,
                        const uint64_t *d_bitlen, const uint64_t *d_key, uint8_t *d_bit, void *d_scratch,
                        void *stream);
int csgn_permute_bitlen(uint64_t n_bits, uint64_t len_words, const uint64_t *d_v, const uint64_t *d_bitlen,
                        const uint32_t *d_perm, uint64_t *d_out, void *d_scratch, void *stream);

/* ------------------------------------------------------------------- harness ---- */

/* Synthetic operand words: word idx = splitmix64(seed + GOLDEN*(idx+1)), the
 * last word of each term masked to its top N%64 bits (the test checker restates this
 * definition independently). */
int csgn_synth_fill(uint64_t seed, uint64_t n_bits, uint64_t first_word, uint64_t n_words,
                    uint64_t *d_out, void *stream);
/* 64-bit order-sensitive digest, ADDED into *d_digest (zero it first):
 * sum_i splitmix64(w[i] + GOLDEN*(first_index+i+1)). */
int csgn_digest(const uint64_t *d_words, uint64_t n_words, uint64_t first_index,
                uint64_t *d_digest, void *stream);

/* ------------------------------------------------------------------ circuits ---- */

/* A fixed circuit of adds, multiplies and decrypts over uniform batches of `batch` ciphertexts
 * (the "circuit runner" of SURVEY 8f-2): every value lives in one HBM block owned by the
 * circuit, and csgn_circuit_build captures all launches into a hipGraph, so a launch-bound
 * circuit (BASELINE config 5: 24 operations of a few microseconds each on single ciphertexts)
 * replays with one graph launch.  Values are numbered from 0 in creation order.  Usage:
 * create; input()*; add()/mul()/decrypt()*; build(); then, per evaluation, write the inputs to
 * csgn_circuit_value(c, id) (batch*terms*dL words each, e.g. csgn_memcpy_d2d or an encrypt
 * kernel writing there directly), csgn_circuit_run, and read results / csgn_circuit_bits after
 * synchronising the stream.  Results are the same words the one-by-one calls produce. */
typedef struct csgn_circuit csgn_circuit;
int csgn_circuit_create(uint64_t n_bits, uint64_t batch, csgn_circuit **circuit);
void csgn_circuit_destroy(csgn_circuit *circuit);
int csgn_circuit_input(csgn_circuit *circuit, uint64_t terms, uint32_t *value);
/* A RAGGED input: element i has h_terms[i] terms (batch entries on the host; 0 allowed).  Shapes are
 * static -- fixed when the circuit is described -- so every size downstream is known on the host, the
 * CSR offsets of every ragged value are uploaded once at build time and the graph needs no plan step.
 * add / mul with a ragged operand give a ragged result (the CSR kernels csgn_add_ragged /
 * csgn_mul_ragged); decrypt works on either kind; permute of a ragged value is not supported.  The
 * input's words go to csgn_circuit_value() in CSR order (element after element). */
int csgn_circuit_input_ragged(csgn_circuit *circuit, const uint64_t *h_terms, uint32_t *value);
int csgn_circuit_add(csgn_circuit *circuit, uint32_t a, uint32_t b, uint32_t *value);   /* a + b: concatenation */
int csgn_circuit_mul(csgn_circuit *circuit, uint32_t a, uint32_t b, uint32_t *value);   /* a * b: all-pairs AND */
/* Decrypt value `a` under the key whose dL-word mask is d_mask (must stay valid); *bits_id
 * names a `batch`-byte result buffer. */
int csgn_circuit_decrypt(csgn_circuit *circuit, uint32_t a, const uint64_t *d_mask, uint32_t *bits_id);
/* EXTENSION (as csgn_compact_ragged, never on a parity path): value = a with every element reduced to its
 * distinct terms of odd multiplicity -- the one node that bounds the growth of a long add/multiply chain.
 * Its sizes are data: the result (and everything computed from it) is a DYNAMIC ragged value -- the shapes the
 * circuit knows for it are upper bounds (they size buffers and launches: csgn_circuit_value_total_terms), the
 * real CSR offsets are written by the device in every run (csgn_circuit_value_offsets; element batch = the
 * real total).  add / mul / decrypt / compact accept dynamic values (mul through the kernels of
 * csgn_mul_ragged_async); permute does not. */
int csgn_circuit_compact(csgn_circuit *circuit, uint32_t a, uint32_t *value);
/* Ciphertext::applyPermutation on every element of value `a` (d_perm: N uint32 entries, must stay
 * valid): as in the reference the result is ONE term, the permuted first term. */
int csgn_circuit_permute(csgn_circuit *circuit, uint32_t a, const uint32_t *d_perm, uint32_t *value);
/* An INPUT produced inside the graph: batch fresh ciphertexts (one term each) of the plaintext bytes
 * at d_plain, encrypted by the keyed generator (csgn_encrypt_keyed) straight into the circuit's block:
 * no staging copy, and a fresh-ciphertext circuit Enc,Enc -> * / + -> Dec (BASELINE configs 2 and 4
 * end to end) replays as ONE graph launch.  d_plain (batch bytes), d_key (D indices) and d_mask stay
 * the caller's and must remain valid; rewrite d_plain between runs to encrypt other bits.  Element i
 * of run r (r = 1, 2, ...; csgn_circuit_epoch after csgn_circuit_run) draws stream position
 * first_ciphertext + i of the generator (node key, nonce = r), where the NODE KEY is derived here,
 * on the host, from (h_rng->key, h_rng->nonce) -- csgn_circuit_node_key -- and the nonce words carry
 * nothing but the run number: the graph's first node increments the run counter on the device, so no
 * replay re-uses a keystream, and no choice of nonce by the caller can make one node's run r the run
 * r' of another.  Give every encrypt node of a circuit its own first_ciphertext range (or its own
 * h_rng->nonce). */
int csgn_circuit_encrypt(csgn_circuit *circuit, uint64_t d, const uint8_t *d_plain, const uint64_t *d_key,
                         const uint64_t *d_mask, const csgn_rng *h_rng, uint64_t first_ciphertext,
                         uint32_t *value);
/* The fused fresh chain as ONE node (csgn_encrypt_mul_keyed inside the graph): value = Enc_A(d_plain_a) *
 * Enc_B(d_plain_b), one term per element, and -- when bits_id is not NULL -- its decryption under the same
 * key as a result buffer.  BASELINE configs 2 / 4 end to end (encrypt, encrypt, multiply, decrypt) are
 * then two graph nodes (run counter + this kernel) and one pass over 8*dL bytes per pair.  The operands
 * themselves are never materialised; a circuit that needs them as values uses csgn_circuit_encrypt and
 * csgn_circuit_mul instead (a tape circuit never rewrites what the caller described: every value it was asked
 * for stays addressable through csgn_circuit_value).  Keys, nonces and runs as for csgn_circuit_encrypt. */
int csgn_circuit_encrypt_mul(csgn_circuit *circuit, uint64_t d, const uint8_t *d_plain_a, const uint8_t *d_plain_b,
                             const uint64_t *d_key, const uint64_t *d_mask, const csgn_rng *h_rng_a,
                             const csgn_rng *h_rng_b, uint64_t first_ciphertext, uint32_t *value, uint32_t *bits_id);
uint64_t csgn_circuit_epoch(const csgn_circuit *circuit);      /* runs launched so far */
/* The key a circuit encrypt node built from *h_rng encrypts under: words 0..7 of the ChaCha20 block
 * (constants "csgn node key v1", key = h_rng->key, nonce = h_rng->nonce, counter 0).  Host only. */
int csgn_circuit_node_key(const csgn_rng *h_rng, uint32_t h_node_key[8]);
/* COMPILED circuits (round 5; SURVEY 8f-2 "never materialise").  By default a circuit is a TAPE: every value it was
 * asked for is written to a region of its own and stays addressable after every run, one kernel per node.
 * csgn_circuit_optimize(flags != 0) before csgn_circuit_build turns the build into a compiler: only inputs, the
 * values named by csgn_circuit_output and the decrypt bits survive a run (csgn_circuit_value returns NULL for
 * every other value), and the passes selected by `flags` arrange the rest:
 *   CSGN_CIRCUIT_REUSE         liveness -- the graph is a chain of kernel nodes, so the region of a value is handed
 *                              out again once its last reader has been emitted: the block is the peak live set
 *                              (csgn_circuit_block_bytes), not the sum of all values;
 *   CSGN_CIRCUIT_PLACE         add is concatenation (src/Ciphertext.cpp:107-122): a product or sum whose only
 *                              consumer is an add is written by its producer straight into its slice of the sum
 *                              (per-element output pitch) -- that operand's copy disappears; operands that cannot be
 *                              placed (inputs, shared values) are copied into their slice alone;
 *   CSGN_CIRCUIT_FUSE_DECRYPT  Dec(a*b) = Dec(a) & Dec(b), Dec(a+b) = Dec(a) ^ Dec(b) (src/SecretKey.cpp:131-140 XORs
 *                              hits over terms): a product or sum whose ONLY consumer is a decrypt is never computed,
 *                              its operands are decrypted and the bits combined -- ONE level;
 *   CSGN_CIRCUIT_PUSHDOWN      the same through every level of single-consumer values (a circuit that only asks for
 *                              bits then decrypts little more than its inputs); not part of CSGN_CIRCUIT_ALL;
 *   CSGN_CIRCUIT_HOIST         copies whose source is a circuit INPUT (an input added to a product, a sum of two
 *                              inputs) depend on nothing the graph computes: all of them go into ONE strided-copy
 *                              launch in front of the first node instead of a small launch each.
 * Nodes whose value nothing reads any more are dropped.  Shared sub-expressions are never fused or placed (a value
 * with two readers is materialised once).  Retained words and all bits are those of the tape, of the one-by-one
 * calls and of the reference.  The graph must be launched on the stream the inputs were written on, or after
 * synchronising with it. */
#define CSGN_CIRCUIT_REUSE 1u
#define CSGN_CIRCUIT_PLACE 2u
#define CSGN_CIRCUIT_FUSE_DECRYPT 4u
#define CSGN_CIRCUIT_PUSHDOWN 8u
#define CSGN_CIRCUIT_HOIST 16u
#define CSGN_CIRCUIT_ALL 23u
int csgn_circuit_optimize(csgn_circuit *circuit, uint32_t flags);
int csgn_circuit_output(csgn_circuit *circuit, uint32_t value);                /* keep this value materialised and addressable */
int csgn_circuit_build(csgn_circuit *circuit);
uint64_t csgn_circuit_block_bytes(const csgn_circuit *circuit);               /* size of the circuit's HBM block; 0 before build */
/* After build: {block bytes, algorithmic bytes of one run (reads + writes of every emitted kernel, SURVEY 8d's
 * per-operation figures), nodes described, kernels emitted, add operands placed (copies that disappeared),
 * decrypts fused, nodes dropped, input copies hoisted into the prologue launch}. */
int csgn_circuit_stats(const csgn_circuit *circuit, uint64_t h_stats[8]);
/* What csgn_circuit_build would do, without touching a device (host only, before build): a JSON object
 * {"bytes", "values": [{region, addressable, offset, pitch, parent, terms, total}], "ops": [{kind, a, b, out, elided,
 * placed_a, placed_b, expr}], "exprs": [{kind (-1 leaf, 0 xor, 1 and), value, l, r}], "regions": [{at, bytes, from, to}]}
 * -- `from`/`to` are the first node that writes and the last that reads a region (-1 = before the run, 2^31-1 = kept).
 * For tests of the compiler and for a caller who wants to see where the bytes of a circuit went. */
int csgn_circuit_plan_json(csgn_circuit *circuit, char *h_json, size_t cap);
uint64_t *csgn_circuit_value(csgn_circuit *circuit, uint32_t value);          /* device pointer; NULL before build, and for a value a compiled circuit did not retain */
uint64_t csgn_circuit_value_terms(csgn_circuit *circuit, uint32_t value);      /* per element; 0 for a ragged value */
uint64_t csgn_circuit_value_total_terms(csgn_circuit *circuit, uint32_t value); /* over the whole batch */
const uint64_t *csgn_circuit_value_offsets(csgn_circuit *circuit, uint32_t value);   /* device CSR offsets (batch+1) of a ragged value, NULL otherwise */
uint8_t *csgn_circuit_bits(csgn_circuit *circuit, uint32_t bits_id);          /* device pointer */
int csgn_circuit_run(csgn_circuit *circuit, void *stream);

/* Name of the kernel(s) a csgn_mul_uniform call of `pairs` pairs (its `batch` argument) of this
 * shape dispatches to ("k_and_stream", "k_mul_tiled", "k_mul_flat", "k_touch+k_mul_flat"); a
 * static string, no GPU needed.  Lets a profiler-driven harness (bench.py) label its roofline
 * with the kernel that really runs. */
const char *csgn_mul_uniform_kernel(uint64_t n_bits, uint64_t pairs, uint64_t t1, uint64_t t2);

/* ------------------------------------------------------------------- tuning ---- */

/* Kernel-choice and sweep knobs ("mul_flat", "mul_touch", "ragged_c", "perm_ballot", ...;
 * csgn_tuning_name(i) enumerates them, NULL past the end; csgn_amd/csrc/csgn_tuning.h documents
 * each).  A knob's start value is its built-in default or the environment variable
 * CSGN_<KEY IN CAPITALS>, sampled ONCE when the library is loaded: no compute entry point reads
 * the environment.  Knobs choose among kernels that produce the same words; results never depend
 * on them.  They are PER HOST THREAD: csgn_set_tuning changes the dispatch of the calling thread's own
 * later calls and of no other thread's (every thread starts from the defaults + the environment
 * snapshot), so concurrent one-thread-per-GPU callers cannot switch one another's kernels.
 * A circuit (csgn_circuit_build) bakes in the building thread's values at build time. */
/* Sharing the GPU.  The default dispatch of a large all-pairs multiply (operand touch pass + flat
 * kernel) is tuned for a caller that has the chip to itself: measured 7.3 TB/s alone, but 3.5 TB/s
 * beside a second stream of back-to-back 1 GiB device copies, where the LDS-tiled kernel holds 4.8
 * (6.9 alone; profiles/r03/cotenant_ab.json).  A host thread whose GPU also serves other streams or
 * processes should call csgn_set_tuning("shared_gpu", 1) (or start with CSGN_SHARED_GPU=1 in the
 * environment): its multiplies then take the kernel that does not depend on what the memory-side
 * cache holds.  Results are identical either way. */
int csgn_set_tuning(const char *key, int value);
int csgn_get_tuning(const char *key, int *h_value);
void csgn_reset_tuning(void);            /* defaults + the environment snapshot taken at load */
const char *csgn_tuning_name(int index);

/* Debug hook: quotient n/d computed by the same division-by-invariant helper the kernels
 * use (csgn_amd/csrc/csgn_common.h); lets the CPU tests pin it without a GPU. */
uint32_t csgn_debug_fastdiv(uint32_t n, uint32_t d);

#ifdef __cplusplus
}
#endif
#endif
