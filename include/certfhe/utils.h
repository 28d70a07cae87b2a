// certfhe/utils.h -- common includes of the drop-in certFHE API (MI355X build).
// Mirrors what /root/reference/src/utils.h makes visible to user code (the std headers and
// the `using namespace std` that the reference's public headers rely on); the reference's
// unparenthesised BIT() macro is replaced by a function so it cannot mis-expand.
#ifndef CERTFHE_UTILS_H
#define CERTFHE_UTILS_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <bitset>
#include <chrono>
#include <iostream>
#include <string>
#include <vector>

namespace certFHE {
inline unsigned char lowBit(int x) { return static_cast<unsigned char>(x & 0x01); }
}

#endif
