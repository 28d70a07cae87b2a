// Plaintext.cpp -- one bit.  Behaviour per /root/reference/src/Plaintext.cpp:10-52.
#include "Plaintext.h"

namespace certFHE {

Plaintext::Plaintext() : value(0) {}

Plaintext::Plaintext(const int v) : value(lowBit(v)) {}

Plaintext::~Plaintext() {}

unsigned char Plaintext::getValue() const { return value; }

void Plaintext::setValue(unsigned char v) { value = v & 0x01; }

ostream &operator<<(ostream &out, const Plaintext &c)
{
    out << static_cast<char>('0' + (c.getValue() & 1)) << endl;   // '0' / '1' and a newline
    return out;
}

} // namespace certFHE
