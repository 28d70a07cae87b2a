// Plaintext.cpp -- stream output of a plaintext bit: the character '0' or '1' and a newline,
// the format of /root/reference/src/Plaintext.cpp:10-19.
#include "Plaintext.h"

namespace certFHE {

ostream &operator<<(ostream &out, const Plaintext &c)
{
    out << static_cast<char>('0' + (c.getValue() & 1)) << endl;
    return out;
}

} // namespace certFHE
