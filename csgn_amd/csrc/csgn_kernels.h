// csgn_kernels.h -- launchers for the gfx950 kernels (internal; the public surface is
// include/csgn_hip.h).
#pragma once

#include "csgn_common.h"

struct csgn_small_op;   // include/csgn_hip.h

namespace csgn {

// Tunables of the tiled all-pairs kernel; defaults chosen by measurement on MI355X
// ("all-pairs multiply").  Overridable through the environment for sweeps:
// CSGN_MUL_M (column units per lane), CSGN_MUL_TI (left terms per tile),
// CSGN_MUL_NT (1 = non-temporal stores).
struct MulTuning {
    int m;      // CSGN_MUL_M: column units per lane, 0 = auto
    int ti;
    int nt;
    int flat;   // CSGN_MUL_FLAT: 0 = choose per shape (mul_plan); k > 0 = flat kernel, k units per lane; -1 = LDS-tiled kernel
    int bs;     // CSGN_MUL_BS: override the tiled kernel's block size (0 = auto)
    int xcd;    // CSGN_MUL_XCD: XCD-contiguous block order: 0 off, 1 flat kernel only (default), 2 both kernels
};
MulTuning mul_tuning();
// name of the kernel(s) a mul_uniform call of `pairs` pairs of this shape dispatches to
// (16-byte aligned buffers assumed)
const char *mul_uniform_kernel_name(u64 n_bits, u64 pairs, u64 t1, u64 t2);

// out_pitch_words != 0 (circuit placement): element e's product is written at out + e * out_pitch_words instead of
// densely, e.g. into its slice of the sum that consumes it; out_slots is ignored then
hipError_t mul_uniform(u64 n_bits, u64 batch, u64 t1, u64 t2, const u64 *L, const u64 *R, u64 *out,
                       u64 out_slots, hipStream_t s, u64 out_pitch_words = 0);
u64 mul_ragged_plan_scratch_words(u64 batch);
u64 mul_ragged_plan_head_words();       // [plan4][huge-pair count][records][operand terms][offsets checksum]: what the host copies back
// What a plan learned beyond its four numbers; lives in the caller's csgn_mul_plan object.
struct MulPlanNotes {
    const u64 *offL = nullptr, *offR = nullptr, *offOut = nullptr;
    u64 batch = 0, total = 0, max_t1 = 0, max_t2 = 0;
    u64 operand_terms = 0;               // left + right terms of the whole batch
    u64 checksum = 0;                    // of the three offset arrays as planned
    u32 n = 0;                           // huge-pair records kept (sorted by pair)
    u64 rec[32][6];                      // {pair, offL, offR, t1, t2, offOut}
    // size classes of small pairs (csgn_mul.hip, class_of): pairs and product terms per class, and the device
    // list of every class's pairs (class c = lists[cls_base[c] .. cls_base[c + 1])); nullptr: no lists
    u64 cls_pairs[28] = {0}, cls_terms[28] = {0}, cls_base[29] = {0};
    const u32 *lists = nullptr;
};
// d_work: the plan's device block (mul_ragged_plan wrote the class lists there) when it stays alive with the
// notes -- a csgn_mul_plan owns one --, nullptr otherwise
void mul_plan_notes_from_head(MulPlanNotes &notes, const u64 *offL, const u64 *offR, const u64 *offOut, u64 batch,
                              const u64 *h_head, const u64 *d_work);
u32 offsets_checksum_words();          // d_sum of offsets_checksum: this many words, to be added up on the host
hipError_t offsets_checksum(u64 batch, const u64 *offL, const u64 *offR, const u64 *offOut, u64 *d_sum, hipStream_t s);
// gate (csgn_mul_ragged_async only): three device words the last plan kernel fills for the kernels behind it
hipError_t mul_ragged_plan(u64 batch, const u64 *offL, const u64 *offR, u64 *offOut, u64 *d_work,
                           hipStream_t s, u64 *gate = nullptr, u64 capacity_terms = 0, bool can_stream = false);
// notes: what the plan of exactly these offset arrays learned about huge pairs and operand size (nullptr: nothing;
// a circuit's offsets never came from a plan).  operand_terms: left + right terms of the whole batch when the
// caller knows them (a circuit does: its shapes are static), 0 = unknown; sizes the output slices of a large product.
// d_gate: csgn_mul_ragged_async -- {real output terms, ...} left on the DEVICE by the plan kernels; the grid is then
// sized by total_out_terms (the caller's bound) and only the CSR kernel is used.
hipError_t mul_ragged(u64 n_bits, u64 batch, const u64 *L, const u64 *offL, const u64 *R,
                      const u64 *offR, u64 *out, const u64 *offOut, u64 max_t1, u64 max_t2,
                      u64 total_out_terms, hipStream_t s, const MulPlanNotes *notes = nullptr, u64 operand_terms = 0,
                      const u64 *d_gate = nullptr, const u64 *d_huge = nullptr);
// plan + multiply enqueued back to back, nothing read back (csgn_mul_ragged_async); d_plan: mul_ragged_async_plan_words(batch)
u64 mul_ragged_async_plan_words(u64 batch);
hipError_t mul_ragged_async(u64 n_bits, u64 batch, const u64 *L, const u64 *offL, const u64 *R, const u64 *offR,
                            u64 *out, u64 *offOut, u64 capacity_terms, u64 *d_plan, hipStream_t s);
// out_pitch_words != 0 (circuit placement): element e's t1 + t2 terms go to out + e * out_pitch_words; with t1 or t2
// zero (that operand's pointer is then not read) this is the strided copy of one operand into its slice of a sum
hipError_t add_uniform(u64 n_bits, u64 batch, u64 t1, u64 t2, const u64 *L, const u64 *R, u64 *out,
                       hipStream_t s, u64 out_pitch_words = 0);
// device_end: total_terms_out is only an upper bound (sizes the launch); the real end is read from the offsets
// a list of strided copies in one launch (the prologue of a compiled circuit: copies whose sources are circuit inputs)
struct CopyEntry {
    const u64 *src;
    u64 *dst;
    u32 elem_words, src_pitch, dst_pitch, batch;   // words per element, words from element to element, elements
};
u32 copy_list_blocks(const CopyEntry &e);              // workgroups entry e needs
// d_first[e] = first workgroup of entry e (exclusive sums of copy_list_blocks), total_blocks = their sum
hipError_t copy_list(const CopyEntry *d_entries, const u32 *d_first, u32 n_entries, u32 total_blocks, hipStream_t s);
// max_t1 / max_t2: upper bounds on one element's terms (0, 0 = unknown); met with equality = a uniform batch
hipError_t add_ragged(u64 n_bits, u64 batch, const u64 *L, const u64 *offL, const u64 *R,
                      const u64 *offR, u64 *out, u64 *offOut, u64 total_terms_out, hipStream_t s,
                      bool device_end = false, u64 max_t1 = 0, u64 max_t2 = 0);
// max_terms (ragged batches): an upper bound on the terms of one ciphertext, 0 = unknown
hipError_t decrypt(u64 n_bits, u64 batch, u64 terms_uniform, u64 total_terms, const u64 *terms,
                   const u64 *off, const u64 *mask, uint8_t *bits, void *scratch, hipStream_t s, u64 max_terms = 0);
// mod-2 compaction (csgn_compact.hip).  max_terms: an upper bound on the terms of any one ciphertext when
// the caller knows it (0 = unknown); it only decides whether the kernels for ciphertexts larger than a
// workgroup's group are launched at all.
bool compact_supported(u64 n_bits);
size_t compact_scratch_bytes(u64 n_bits, u64 batch, u64 total_terms);
hipError_t compact(u64 n_bits, u64 batch, u64 total_terms, u64 max_terms, const u64 *terms, const u64 *off,
                   u64 *out, u64 *off_out, void *scratch, hipStream_t s);
hipError_t encrypt(u64 n_bits, u64 d, u64 batch, const uint8_t *plain, const u64 *rnd,
                   const u32 *chosen, const uint8_t *last, const u64 *mask, u64 *out, hipStream_t s);
// Keyed (ChaCha) device-RNG encrypt; keystream layout in csgn_encrypt.hip.
hipError_t bump_epoch(u64 *d_epoch, hipStream_t s);      // *d_epoch += 1 (head node of a circuit with encrypt inputs)
void encrypt_keyed_layout(u64 n_bits, u32 *U, u32 *P, u32 *Gc);
hipError_t encrypt_keyed(u64 n_bits, u64 d, u64 batch, u64 first_ct, const uint8_t *plain, const u64 *key_idx,
                         const u64 *mask, const u32 rng_key[8], u64 nonce, u32 rounds,
                         const u64 *d_epoch, u64 *out, hipStream_t s);
// Fused fresh chain: out_c = Enc_A(plain_a[c]) & Enc_B(plain_b[c]) in one kernel, optional Dec of it.
hipError_t encrypt_mul_keyed(u64 n_bits, u64 d, u64 batch, u64 first_ct, const uint8_t *plain_a,
                             const uint8_t *plain_b, const u64 *key_idx, const u64 *mask, const u32 key_a[8],
                             u64 nonce_a, const u32 key_b[8], u64 nonce_b, u32 rounds, const u64 *d_epoch, u64 *out,
                             uint8_t *bits, hipStream_t s);
hipError_t permute(u64 n_bits, u64 batch, u64 terms_in, bool per_term, const u64 *terms,
                   const u32 *perm, u64 *out, hipStream_t s);
// one ciphertext with an explicit bitlen side array (csgn_bitlen.hip)
size_t bitlen_scratch_bytes(u64 len);
hipError_t decrypt_bitlen(u64 n_bits, u64 d, u64 len, const u64 *v, const u64 *bitlen, const u64 *key,
                          uint8_t *bit, void *scratch, hipStream_t s);
hipError_t permute_bitlen(u64 n_bits, u64 len, const u64 *v, const u64 *bitlen, const u32 *perm, u64 *out,
                          void *scratch, hipStream_t s);
hipError_t circuit_zero_words(u64 *p, u64 n, hipStream_t s);   // zero-fill by a kernel (graph-safe, csgn_device.h)
hipError_t synth_fill(u64 seed, u64 n_bits, u64 first_word, u64 n_words, u64 *out, hipStream_t s);
hipError_t digest(const u64 *w, u64 n_words, u64 first_index, u64 *d_digest, hipStream_t s);

hipError_t small_ops(u64 n_bits, u64 count, const ::csgn_small_op *ops, hipStream_t s);
size_t decrypt_scratch_bytes(u64 batch, u64 total_terms);
// out[i] = a[i] & b[i] (is_product) or a[i] ^ b[i]: Dec(a*b) = Dec(a) & Dec(b), Dec(a+b) = Dec(a) ^ Dec(b)
hipError_t combine_bits(const uint8_t *a, const uint8_t *b, u64 n, bool is_product, uint8_t *out, hipStream_t s);
size_t decrypt_combined_scratch_bytes(u64 batch, u64 t1, u64 t2);
hipError_t decrypt_combined(u64 n_bits, u64 batch, u64 t1, u64 t2, const u64 *L, const u64 *R,
                            const u64 *mask, bool is_product, uint8_t *bits, void *scratch, hipStream_t s);

} // namespace csgn
