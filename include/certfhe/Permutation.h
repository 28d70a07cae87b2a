// certfhe/Permutation.h -- permutations of the N bit positions (host-side object).
// Public surface of /root/reference/src/Permutation.h:27-89.
#ifndef CERTFHE_PERMUTATION_H
#define CERTFHE_PERMUTATION_H

#include "Context.h"
#include "Helpers.h"
#include "utils.h"

using namespace std;

namespace certFHE {

class Permutation {
    uint64_t *permutation;
    uint64_t length;

    void adopt(const uint64_t *src, uint64_t len);

  public:
    Permutation();                                        // empty
    Permutation(const uint64_t *perm, const uint64_t len); // copy of a given table
    Permutation(const Context &context);                  // random, size N
    Permutation(const uint64_t len);                      // random, size len
    Permutation(const Permutation &perm);
    virtual ~Permutation();

    uint64_t getLength() const;
    void setLength(uint64_t len);
    void setPermutation(uint64_t *perm, uint64_t len);
    uint64_t *getPermutation() const;                     // borrowed; do not delete

    friend ostream &operator<<(ostream &out, const Permutation &c);
    Permutation &operator=(const Permutation &perm);

    Permutation getInverse();
    Permutation operator+(const Permutation &permB) const; // this o permB
    Permutation &operator+=(const Permutation &permB);
};

} // namespace certFHE

#endif
