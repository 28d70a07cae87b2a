// certfhe/Context.h -- scheme parameters (N, D, S, words per term).
// Public surface of /root/reference/src/Context.h:28-69.
#ifndef CERTFHE_CONTEXT_H
#define CERTFHE_CONTEXT_H

#include "Helpers.h"
#include "utils.h"

using namespace std;

namespace certFHE {

class Context {
    uint64_t N;          // bits per ciphertext term
    uint64_t D;          // secret positions
    uint64_t S;          // N / (2 D)
    uint64_t defaultLen; // 64-bit words per term, ceil(N / 64)

    void derive();

  public:
    Context() = delete;
    Context(const Context &context);
    Context(const uint64_t pN, const uint64_t pD);
    virtual ~Context();

    Context &operator=(const Context &context);
    friend ostream &operator<<(ostream &out, const Context &c);

    uint64_t getN() const;
    uint64_t getD() const;
    uint64_t getS() const;
    uint64_t getDefaultN() const;

    // Unlike the reference (src/Context.cpp:81-85, which leaves defaultLen stale) setN
    // re-derives the words-per-term count.
    void setN(uint64_t n);
    void setD(uint64_t d);
};

} // namespace certFHE

#endif
