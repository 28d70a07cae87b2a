// Context.cpp -- scheme parameters.  Behaviour per /root/reference/src/Context.cpp:20-91.
#include "Context.h"

#include "csgn_hip.h"

namespace certFHE {

void Context::derive()
{
    S = csgn_context_s(N, D);
    defaultLen = csgn_default_len(N);
}

Context::Context(const uint64_t pN, const uint64_t pD) : N(pN), D(pD), S(0), defaultLen(0) { derive(); }

Context::Context(const Context &c) : N(c.N), D(c.D), S(c.S), defaultLen(c.defaultLen) {}

Context::~Context() {}

Context &Context::operator=(const Context &c)
{
    N = c.N;
    D = c.D;
    S = c.S;
    defaultLen = c.defaultLen;
    return *this;
}

ostream &operator<<(ostream &out, const Context &c)
{
    out << "N= " << c.getN() << endl << "D= " << c.getD() << endl << "S= " << c.getS() << endl;
    return out;
}

uint64_t Context::getN() const { return N; }
uint64_t Context::getD() const { return D; }
uint64_t Context::getS() const { return S; }
uint64_t Context::getDefaultN() const { return defaultLen; }

void Context::setN(uint64_t n)
{
    N = n;
    derive();
}

void Context::setD(uint64_t d)
{
    D = d;
    derive();
}

} // namespace certFHE
