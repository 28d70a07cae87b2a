// certfhe/SecretKey.h -- secret key, encrypt and decrypt on the MI355X.
// Same public interface as /root/reference/src/SecretKey.h:67-143.
//
// encrypt() consumes libc rand() in exactly the order the reference does
// (src/SecretKey.cpp:35-80) and hands the draws to csgn_encrypt_explicit, so under the
// same srand() the ciphertext bits equal the reference's.  decrypt() runs the
// AND-over-key / XOR-over-terms reduction on the device (csgn_decrypt_uniform) against a
// cached dL-word key mask.
#ifndef CERTFHE_SECRET_KEY_H
#define CERTFHE_SECRET_KEY_H

#include <memory>

#include "Ciphertext.h"
#include "Context.h"
#include "Helpers.h"
#include "Permutation.h"
#include "Plaintext.h"
#include "utils.h"

using namespace std;

namespace certFHE {

class CiphertextBatch;
class ShardedBatch;

class SecretKey {
    uint64_t *s;      // D secret positions in [0, N), in generation order
    long length;
    Context *certFHEContext;

    mutable std::shared_ptr<detail::DevicePayload> device_mask; // dL-word key mask in HBM
    mutable std::vector<uint64_t> host_mask;

    void invalidateMask();
    void ensureMask() const;
    friend class CiphertextBatch;      // extension (Batch.h): reads the device-resident key mask
    friend class BatchCircuit;         // extension (Batch.h)
    friend class ShardedBatch;         // extension (ShardedBatch.h): reads the key's Context

  public:
    SecretKey() = delete;
    SecretKey(const Context &context);      // key generation (seeds rand() from the clock)
    SecretKey(const SecretKey &secKey);
    virtual ~SecretKey();                   // zeroises the key

    Ciphertext encrypt(Plaintext &plaintext);
    Plaintext decrypt(Ciphertext &ciphertext);

    void applyPermutation_inplace(const Permutation &permutation);
    SecretKey applyPermutation(const Permutation &permutation);

    friend ostream &operator<<(ostream &out, const SecretKey &c);
    SecretKey &operator=(const SecretKey &secKey);

    uint64_t getLength() const;
    uint64_t *getKey() const;               // borrowed; do not delete
    void setKey(uint64_t *s, uint64_t len);

    long size();
};

} // namespace certFHE

#endif
