// certfhe/Plaintext.h -- one bit of F2.  Public surface of /root/reference/src/Plaintext.h:24-45.
#ifndef CERTFHE_PLAINTEXT_H
#define CERTFHE_PLAINTEXT_H

#include "utils.h"

using namespace std;

namespace certFHE {

class Plaintext {
    unsigned char value;

  public:
    Plaintext();
    Plaintext(const int value);
    virtual ~Plaintext();

    unsigned char getValue() const;
    void setValue(unsigned char value);

    friend ostream &operator<<(ostream &out, const Plaintext &c);
};

} // namespace certFHE

#endif
