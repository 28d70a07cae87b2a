// certfhe/Batch.h -- EXTENSION (not in the reference): a batch of ciphertexts that stays in
// MI355X HBM, for the throughput the value-per-object API cannot reach.
//
// A CiphertextBatch holds `count` independent ciphertexts laid back to back.  UNIFORM (every
// element `terms` terms): exactly what csgn_mul_uniform / csgn_add_uniform / csgn_decrypt_uniform
// expect (include/csgn_hip.h).  RAGGED (element i has termsOf(i) terms; what compact() returns when
// the elements end up with different sizes): CSR term offsets beside the words, the csgn_*_ragged
// entry points.  Element i of `a * b` is `a[i] * b[i]` of the reference
// (src/Ciphertext.cpp:231-247), element i of `a + b` is `a[i] + b[i]` (:204-229); one call does
// the whole batch.
#ifndef CERTFHE_BATCH_H
#define CERTFHE_BATCH_H

#include <memory>
#include <utility>
#include <vector>

#include "Ciphertext.h"
#include "Context.h"
#include "Permutation.h"
#include "SecretKey.h"

struct csgn_circuit;   // include/csgn_hip.h (opaque)

namespace certFHE {

class BatchCircuit;

class CiphertextBatch {
    friend class BatchCircuit;
    std::shared_ptr<detail::DevicePayload> payload;   // total terms * dL words, element after element
    uint64_t count_;
    uint64_t terms_;                                   // per element when uniform; 0 when ragged
    Context ctx;
    std::vector<uint64_t> offsets_;                    // ragged: count+1 term offsets (host copy)
    mutable std::shared_ptr<detail::DevicePayload> d_offsets_;   // the same in HBM (made on first use, also for a uniform batch)

    CiphertextBatch(const Context &c, uint64_t count, uint64_t terms);
    const uint64_t *deviceOffsets() const;             // CSR offsets in HBM
    uint64_t maxTerms() const;

  public:
    // Encrypts bits[i] under `key` on the device: same distribution as SecretKey::encrypt
    // (src/SecretKey.cpp:35-80), randomness from a keyed ChaCha generator (csgn_encrypt_keyed) whose
    // 256-bit key and nonce come fresh from the operating system for every call.
    static CiphertextBatch encrypt(const SecretKey &key, const std::vector<unsigned char> &bits);
    // REPRODUCIBLE form for tests and benchmarks: generator key expanded from a 64-bit seed
    // (csgn_rng_from_seed) -- not for ciphertexts that have to stay secret.  Element i draws stream
    // position first_ciphertext + i, so shards of one logical batch agree with the whole.
    static CiphertextBatch encrypt(const SecretKey &key, const std::vector<unsigned char> &bits,
                                   uint64_t seed, uint64_t first_ciphertext = 0);
    // Packs existing single ciphertexts (all with the same term count) into a batch.
    static CiphertextBatch pack(const std::vector<Ciphertext> &items);

    CiphertextBatch operator*(const CiphertextBatch &rhs) const;   // element-wise product
    CiphertextBatch operator+(const CiphertextBatch &rhs) const;   // element-wise sum

    // Ciphertext::applyPermutation on every element (as in the reference the result has ONE term:
    // the permuted first term of each element).
    CiphertextBatch applyPermutation(const Permutation &permutation) const;

    // EXTENSION beyond the reference's semantics (its add never reduces, src/Ciphertext.cpp:107-122):
    // every element rewritten as its distinct terms of odd multiplicity, in order of first occurrence
    // (csgn_compact_ragged).  Decryption XORs over terms (src/SecretKey.cpp:139), so identical terms
    // cancel in pairs: Dec of every element is unchanged under ANY key, while e.g. (a+b)*(a+b) shrinks
    // from 4 terms to 2 and a long add/multiply chain stops growing by its duplicates.  The words are
    // no longer the reference's; never part of a parity comparison.  The result is uniform again when
    // every element kept the same number of terms, ragged otherwise.
    CiphertextBatch compact() const;

    // One plaintext bit per element.
    std::vector<unsigned char> decrypt(const SecretKey &key) const;
    // Dec(this[i] * rhs[i]) / Dec(this[i] + rhs[i]) without materialising the results (uniform batches).
    std::vector<unsigned char> decryptProduct(const CiphertextBatch &rhs, const SecretKey &key) const;
    std::vector<unsigned char> decryptSum(const CiphertextBatch &rhs, const SecretKey &key) const;

    Ciphertext at(uint64_t i) const;          // copy of element i as an ordinary Ciphertext
    uint64_t size() const { return count_; }
    uint64_t terms() const { return terms_; }  // per element; 0 for a ragged batch (see termsOf)
    bool uniform() const { return offsets_.empty(); }
    uint64_t termsOf(uint64_t i) const;        // element i's term count
    uint64_t totalTerms() const { return uniform() ? count_ * terms_ : offsets_.back(); }
    const Context &context() const { return ctx; }
    const uint64_t *deviceValues() const;
};

// EXTENSION: a fixed add/multiply/decrypt circuit over uniform batches, captured once into a
// hipGraph (csgn_circuit_* in include/csgn_hip.h).  For circuits whose operations are too small to
// fill the GPU -- BASELINE config 5 on a handful of ciphertexts -- one graph launch replaces two
// dozen kernel launches (measured: 58 us instead of 116 us per depth-16 circuit).
//
//     BatchCircuit c(ctx, count);
//     unsigned a = c.input(1), b = c.input(1), k = c.input(1);
//     unsigned r = c.mul(c.add(a, b), k);
//     unsigned bits = c.decrypt(r, key);
//     c.build();
//     c.set(a, batchA); c.set(b, batchB); c.set(k, batchK);
//     c.run();
//     std::vector<unsigned char> plain = c.bits(bits);   CiphertextBatch out = c.value(r);
class BatchCircuit {
    ::csgn_circuit *handle;
    Context ctx;
    uint64_t count_;
    std::vector<std::shared_ptr<detail::DevicePayload> > masks;   // key masks / permutations / plaintext bytes the graph refers to
    std::vector<std::pair<unsigned, std::shared_ptr<detail::DevicePayload> > > plains;   // encrypt inputs: value id -> plaintext bytes
    std::vector<std::pair<unsigned, std::shared_ptr<detail::DevicePayload> > > pair_plains;   // fused products: value id -> both operands' plaintext bytes
    uint64_t next_first;                                           // stream range handed to the next encrypt input
    BatchCircuit(const BatchCircuit &);
    BatchCircuit &operator=(const BatchCircuit &);

  public:
    BatchCircuit(const Context &context, uint64_t count);
    ~BatchCircuit();
    unsigned input(uint64_t terms = 1);
    // An input ENCRYPTED INSIDE THE GRAPH (csgn_circuit_encrypt): `count` fresh ciphertexts of the bits
    // last given to setPlain(), under `key`, by the keyed generator with a key drawn from the OS here;
    // every run() draws a new keystream.  No staging copy: Enc,Enc -> * -> Dec is one graph launch.
    unsigned encryptInput(const SecretKey &key);
    void setPlain(unsigned encrypted_input, const std::vector<unsigned char> &bits);
    // The fused fresh chain as ONE node (csgn_circuit_encrypt_mul): value = Enc(a[i]) * Enc(b[i]) for the bits
    // last given to setPlainPair(), both operands generated in registers, only the product written; when
    // bits_id is given it receives the id for bits(): the product's decryption, computed by the same
    // kernel.  The reference's canonical flow (tests/basic_operations.cpp:26-40) is then one kernel.
    unsigned encryptProduct(const SecretKey &key, unsigned *bits_id = nullptr);
    void setPlainPair(unsigned product, const std::vector<unsigned char> &a, const std::vector<unsigned char> &b);
    unsigned add(unsigned a, unsigned b);
    unsigned mul(unsigned a, unsigned b);
    // EXTENSION (see CiphertextBatch::compact): every element reduced to its distinct terms of odd multiplicity.
    // The result and everything computed from it have data-dependent sizes: value() returns them as a ragged
    // batch; permute() does not accept them.
    unsigned compact(unsigned a);
    unsigned permute(unsigned a, const Permutation &p);   // applyPermutation: ONE term, the permuted first term
    unsigned decrypt(unsigned a, const SecretKey &key);   // returns the id for bits()
    // COMPILED circuits (csgn_circuit_optimize): call before build().  By default every value described above is
    // computed into a buffer of its own and value() answers for all of them.  optimize() lets build() arrange the
    // work -- a product whose only reader is an add is written straight into the sum, a product or sum whose only
    // reader is a decrypt is never computed (Dec(a*b) = Dec(a) & Dec(b), Dec(a+b) = Dec(a) ^ Dec(b)), buffers are
    // reused once their last reader has run -- and afterwards value() answers only for inputs and for values named by
    // keep().  Bits and kept values are the same words either way.  passes: CSGN_CIRCUIT_* of csgn_hip.h.
    void optimize(unsigned passes = 23u /* CSGN_CIRCUIT_ALL */);
    void keep(unsigned value);
    uint64_t blockBytes() const;                              // HBM held by the built circuit
    void build();
    void set(unsigned input, const CiphertextBatch &batch);   // copies the batch into the input's buffer
    void run();                                                // one graph launch (asynchronous)
    CiphertextBatch value(unsigned id) const;                  // copy of a value after run() (compiled circuits: inputs and keep()-ed values)
    std::vector<unsigned char> bits(unsigned bits_id) const;   // synchronises
};

} // namespace certFHE

#endif
