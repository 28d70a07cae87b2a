// certfhe/ShardedBatch.h -- EXTENSION (not in the reference): one logical batch of independent
// ciphertexts spread over the GPUs of a node, behind the certFHE:: class surface.
//
// The reference is single-device; its operators work on one Ciphertext at a time
// (src/Ciphertext.h:65-143).  Every element of a batch is independent, so a batch shards with no
// data-path exchange: element p of B goes to GPU p*G/B (contiguous ranges,
// csgn_shard_range), each GPU keeps its shard in its own HBM and runs the same kernels
// CiphertextBatch does, and the ONLY traffic between GPUs is
//   termCounts()  one uint64 per element  (newlen/dL of src/Ciphertext.cpp:146; ncclAllGather over xGMI)
//   decrypt()     one byte per element    (the plaintext bits; a second all-gather)
// Element i of `a * b` is `a[i] * b[i]` of the reference (src/Ciphertext.cpp:231-247), of `a + b`
// `a[i] + b[i]` (:204-229); results do not depend on the number of GPUs.
//
// A ShardGroup owns one host thread and one RCCL communicator per GPU (include/csgn_shard.h).  Every
// method of ShardedBatch posts one task per GPU and waits for all of them; if any GPU's task fails,
// that thread aborts every communicator of the group (ncclCommAbort) so that no peer stays blocked
// in a collective, and the call throws std::runtime_error naming the rank -- it never hangs.  After a
// failure the group is dead: further calls throw at once.
//
// Library: libcertFHE_shard.so (links libcsgn_shard.so and with it RCCL; libcertFHE.so does not).
#ifndef CERTFHE_SHARDED_BATCH_H
#define CERTFHE_SHARDED_BATCH_H

#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "Context.h"
#include "Permutation.h"
#include "SecretKey.h"

namespace certFHE {

namespace detail {
struct ShardGroupImpl;
struct ShardedData;
}

class ShardGroup {
    std::shared_ptr<detail::ShardGroupImpl> impl;
    friend class ShardedBatch;

  public:
    // devices: HIP device numbers, one rank each; empty = every visible device.
    // Throws std::runtime_error when there is no gfx950 device (no CPU fallback) or RCCL cannot
    // form the communicator.
    explicit ShardGroup(const std::vector<int> &devices = std::vector<int>());
    int size() const;                          // number of GPUs = ranks
    int device(int rank) const;
    bool healthy() const;                      // false once any rank has failed
    // "RCCL 2.27.7 (/opt/rocm/lib/librccl.so.1), header 2.27.7": which RCCL the process bound
    std::string collective() const;
    // Longest wait for peers in a collective, in milliseconds (default 120 000; 0 = for ever).
    void setTimeoutMs(uint64_t ms);
    // csgn_set_tuning(key, value) on EVERY worker thread of the group.  The library's knobs are per host thread
    // (include/csgn_hip.h, "tuning"), and a ShardedBatch's kernels are launched by the group's worker threads:
    // csgn_set_tuning from the application thread -- e.g. "shared_gpu" = 1 on co-tenant GPUs -- does not reach
    // them; this does.  (The CSGN_* environment snapshot reaches every thread either way.)
    void setTuning(const std::string &key, int value);
    // TEST HOOK: the next task of `rank` throws before doing anything (exercises the failure path).
    void injectFailure(int rank);
    // TEST HOOK: gathers take the uneven-shard (grouped ncclBroadcast) form even for equal shards.
    void forceGroupedBroadcast(bool on);
};

class ShardedBatch {
    std::shared_ptr<detail::ShardedData> data;
    explicit ShardedBatch(const std::shared_ptr<detail::ShardedData> &d) : data(d) {}
    std::vector<unsigned char> decryptWith(const ShardedBatch *rhs, bool product, const SecretKey &key) const;
    static const Context &keyContext(const SecretKey &key);
    static ShardedBatch encryptWith(ShardGroup &group, const SecretKey &key, const std::vector<unsigned char> &bits,
                                    const void *rng, uint64_t first_ciphertext);

  public:
    // Encrypts bits[i] under `key`: same distribution as SecretKey::encrypt (src/SecretKey.cpp:35-80),
    // randomness from the keyed ChaCha generator (csgn_encrypt_keyed) under a 256-bit key and nonce
    // drawn from the operating system once per call; element i draws stream position i on whichever
    // GPU owns it.
    static ShardedBatch encrypt(ShardGroup &group, const SecretKey &key, const std::vector<unsigned char> &bits);
    // REPRODUCIBLE form for tests and benchmarks (csgn_rng_from_seed -- not for secrets): element i draws
    // stream position first_ciphertext + i, so the words are the same for ANY number of GPUs and equal
    // CiphertextBatch::encrypt(key, bits, seed, first_ciphertext).
    static ShardedBatch encrypt(ShardGroup &group, const SecretKey &key, const std::vector<unsigned char> &bits,
                                uint64_t seed, uint64_t first_ciphertext = 0);
    // Enc(bits_a[i]) * Enc(bits_b[i]) for fresh ciphertexts in ONE kernel per GPU
    // (csgn_encrypt_mul_keyed): both operands are generated in registers and only the product is
    // written.  The words equal encrypt(.., seed_a) * encrypt(.., seed_b).
    static ShardedBatch encryptProduct(ShardGroup &group, const SecretKey &key,
                                       const std::vector<unsigned char> &bits_a,
                                       const std::vector<unsigned char> &bits_b, uint64_t seed_a, uint64_t seed_b,
                                       uint64_t first_ciphertext = 0);
    // Synthetic operands for benchmarks (csgn_synth_fill at the element's GLOBAL word position).
    static ShardedBatch synthetic(ShardGroup &group, const Context &context, uint64_t count, uint64_t terms,
                                  uint64_t seed);

    ShardedBatch operator*(const ShardedBatch &rhs) const;   // element-wise product, shard by shard
    ShardedBatch operator+(const ShardedBatch &rhs) const;   // element-wise sum

    // Ciphertext::applyPermutation on every element, shard by shard (as in the reference and in
    // CiphertextBatch the result has ONE term: the permuted first term of each element,
    // src/Ciphertext.cpp:7-89).  No exchange between GPUs.
    ShardedBatch applyPermutation(const Permutation &permutation) const;

    // One plaintext bit per element, in global order: every GPU decrypts its shard, the bytes are
    // all-gathered (csgn_comm_gather_bytes), rank 0's copy is returned.
    std::vector<unsigned char> decrypt(const SecretKey &key) const;
    // Dec(this[i] * rhs[i]) / Dec(this[i] + rhs[i]) without materialising the results
    // (csgn_decrypt_product_uniform / csgn_decrypt_sum_uniform on every shard, then the same byte gather).
    std::vector<unsigned char> decryptProduct(const ShardedBatch &rhs, const SecretKey &key) const;
    std::vector<unsigned char> decryptSum(const ShardedBatch &rhs, const SecretKey &key) const;
    // Per-element result term counts in global order = the gathered vector of
    // csgn_comm_gather_counts (ncclAllGather of one uint64 per element).  Every rank receives the
    // whole vector; they are compared with one another and rank 0's is returned.
    std::vector<uint64_t> termCounts() const;

    uint64_t size() const;                      // elements in the whole batch
    uint64_t terms() const;                     // terms per element
    int shards() const;                         // = group size
    std::pair<uint64_t, uint64_t> shardRange(int rank) const;      // [lo, hi) of the elements on that GPU
    const Context &context() const;
    // Host copy of element i's words (terms * dL), wherever it lives.
    std::vector<uint64_t> values(uint64_t i) const;
    // csgn_digest over the whole batch at global word positions (sum over shards): the same number
    // for any GPU count, equal to the digest of the same batch on one GPU.
    uint64_t digest() const;
    // Waits until every GPU has finished the work queued so far.
    void synchronize() const;
};

} // namespace certFHE

#endif
