// certfhe/Plaintext.h -- one bit of F2 (drop-in for the reference's Plaintext class,
// /root/reference/src/Plaintext.h:24-45).  Trivial enough to live in the header; only the
// stream operator is out of line.
#ifndef CERTFHE_PLAINTEXT_H
#define CERTFHE_PLAINTEXT_H

#include "utils.h"

using namespace std;

namespace certFHE {

class Plaintext {
    unsigned char value;   // 0 or 1

  public:
    Plaintext() : value(0) {}
    Plaintext(const int v) : value(lowBit(v)) {}   // keeps only the low bit, like BIT()
    virtual ~Plaintext() {}

    unsigned char getValue() const { return value; }
    void setValue(unsigned char v) { value = v & 0x01; }

    // prints '0' or '1' followed by a newline (src/Plaintext.cpp:10-19)
    friend ostream &operator<<(ostream &out, const Plaintext &c);
};

} // namespace certFHE

#endif
