// certfhe/Ciphertext.h -- term-list ciphertext whose payload lives in MI355X HBM.
//
// Same public interface as /root/reference/src/Ciphertext.h:65-143, so user code compiles
// unchanged.  What differs is where the data is:
//   - the T*dL term words sit in device memory behind a reference-counted, immutable
//     payload; copies share it, every operator produces a fresh payload through
//     libcsgn_hip (csgn_mul_uniform / csgn_add_uniform / csgn_permute_uniform);
//   - getValues() / getBitlen() return BORROWED host mirrors that are filled on demand
//     and stay valid until the next mutating call on the object ("DO NOT DELETE", as in
//     the reference).  Writing through them does not change the ciphertext;
//   - the bitlen side-array is implied by (N, T) and only materialised when asked for;
//     a ciphertext constructed with some other Bitlen keeps it on the host and follows
//     the reference's propagation rules (left operand's term under *, concatenation
//     under +); decrypt and applyPermutation then read (values, Bitlen) as the bit stream the
//     reference reads (csgn_decrypt_bitlen / csgn_permute_bitlen), on the device as well.
#ifndef CERTFHE_CIPHERTEXT_H
#define CERTFHE_CIPHERTEXT_H

#include <memory>

#include "Context.h"
#include "Permutation.h"
#include "utils.h"

using namespace std;

namespace certFHE {

namespace detail {
struct DevicePayload;   // device buffer + word count (csgn_amd/csrc/certfhe/runtime.h)
struct LazyNode;        // a queued small + or * whose result does not exist yet (runtime.h)
}

class SecretKey;
class CiphertextBatch;

class Ciphertext {
    mutable std::shared_ptr<detail::DevicePayload> payload; // immutable once published
    mutable std::shared_ptr<detail::LazyNode> lazy; // set instead of payload while the producing operation is queued
    void resolve() const;                           // queue evaluated, payload set
    uint64_t len;                                   // words (T * dL)
    Context *certFHEcontext;                        // owned copy, may be null (default ctor)

    mutable uint64_t *host_v;          // lazy mirror for getValues()
    mutable uint64_t *host_bitlen;     // lazy canonical pattern, or the custom one
    bool custom_bitlen;                // true: host_bitlen is authoritative

    void dropMirrors();
    void publish(const std::shared_ptr<detail::DevicePayload> &p, uint64_t words);
    void combineBitlen(const Ciphertext &lhs, const Ciphertext &rhs, bool product);
    static Ciphertext combine(const Ciphertext &a, const Ciphertext &b, bool product);

    friend class SecretKey;
    friend class CiphertextBatch;      // extension (Batch.h)

  public:
    Ciphertext();
    Ciphertext(const uint64_t *V, const uint64_t *Bitlen, const uint64_t len, const Context &context);
    Ciphertext(const Ciphertext &ctxt);
    virtual ~Ciphertext();

    void setValues(const uint64_t *V, const uint64_t length);
    void setBitlen(const uint64_t *Bitlen, const uint64_t length);
    void setContext(const Context &context);
    uint64_t getLen() const;
    Context getContext() const;
    uint64_t *getValues() const;   // borrowed host mirror
    uint64_t *getBitlen() const;   // borrowed

    friend ostream &operator<<(ostream &out, const Ciphertext &c);

    Ciphertext operator+(const Ciphertext &c) const;
    Ciphertext &operator+=(const Ciphertext &c);
    Ciphertext operator*(const Ciphertext &c) const;
    Ciphertext &operator*=(const Ciphertext &c);
    Ciphertext &operator=(const Ciphertext &c);

    void applyPermutation_inplace(const Permutation &permutation);
    Ciphertext applyPermutation(const Permutation &permutation);

    long size();

    // --- extensions (not in the reference) ---
    uint64_t getTerms() const;               // len / dL
    const uint64_t *deviceValues() const;    // HBM pointer, valid while this object is unchanged
    bool hasCanonicalBitlen() const;

    // Wire format "CSGN" v1 (the reference has no save/load, only the raw
    // getValues()/4-arg-constructor round trip).  Little-endian:
    //   char[4] "CSGN" | u16 version=1 | u16 flags (bit0: explicit Bitlen array follows)
    //   u64 N | u64 D | u64 len (words) | len x u64 term words [| len x u64 Bitlen]
    // The bitlen side-array is implied by (N, len) and omitted unless it is non-canonical.
    void serialize(std::ostream &out) const;
    static Ciphertext deserialize(std::istream &in);
    // The same bytes to / from memory of the caller's, without a stream in between: the words go by ONE copy between
    // HBM and the buffer.  A std::ostream sink costs a memcpy on one host core (17-19 GB/s); a page-locked buffer
    // (csgn_host_alloc, hipHostMalloc, hipHostRegister) is written by the DMA engine at the link's rate.
    uint64_t serializedSize() const;                                    // bytes serialize() would write
    uint64_t serializeTo(void *buffer, uint64_t capacity) const;        // returns the bytes written; throws if they do not fit
    static Ciphertext deserializeFrom(const void *buffer, uint64_t bytes);
};

} // namespace certFHE

#endif
