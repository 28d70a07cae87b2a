// certfhe/Batch.h -- EXTENSION (not in the reference): a uniform batch of ciphertexts that
// stays in MI355X HBM, for the throughput the value-per-object API cannot reach.
//
// A CiphertextBatch holds `count` independent ciphertexts of `terms` terms each, laid back to
// back exactly as csgn_mul_uniform / csgn_add_uniform / csgn_decrypt_uniform expect
// (include/csgn_hip.h).  Element i of `a * b` is `a[i] * b[i]` of the reference
// (src/Ciphertext.cpp:231-247), element i of `a + b` is `a[i] + b[i]` (:204-229); one kernel
// launch does the whole batch.
#ifndef CERTFHE_BATCH_H
#define CERTFHE_BATCH_H

#include <memory>
#include <vector>

#include "Ciphertext.h"
#include "Context.h"
#include "SecretKey.h"

namespace certFHE {

class CiphertextBatch {
    std::shared_ptr<detail::DevicePayload> payload;   // count * terms * dL words
    uint64_t count_;
    uint64_t terms_;
    Context ctx;

    CiphertextBatch(const Context &c, uint64_t count, uint64_t terms);

  public:
    // Encrypts bits[i] under `key` with the device's counter-based generator (same
    // distribution as SecretKey::encrypt, not the libc rand() stream); `seed` selects the stream.
    static CiphertextBatch encrypt(const SecretKey &key, const std::vector<unsigned char> &bits,
                                   uint64_t seed);
    // Packs existing single ciphertexts (all with the same term count) into a batch.
    static CiphertextBatch pack(const std::vector<Ciphertext> &items);

    CiphertextBatch operator*(const CiphertextBatch &rhs) const;   // element-wise product
    CiphertextBatch operator+(const CiphertextBatch &rhs) const;   // element-wise sum

    // One plaintext bit per element.
    std::vector<unsigned char> decrypt(const SecretKey &key) const;
    // Dec(this[i] * rhs[i]) / Dec(this[i] + rhs[i]) without materialising the results.
    std::vector<unsigned char> decryptProduct(const CiphertextBatch &rhs, const SecretKey &key) const;
    std::vector<unsigned char> decryptSum(const CiphertextBatch &rhs, const SecretKey &key) const;

    Ciphertext at(uint64_t i) const;          // copy of element i as an ordinary Ciphertext
    uint64_t size() const { return count_; }
    uint64_t terms() const { return terms_; }
    const Context &context() const { return ctx; }
    const uint64_t *deviceValues() const;
};

} // namespace certFHE

#endif
