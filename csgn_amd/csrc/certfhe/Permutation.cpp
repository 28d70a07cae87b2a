// Permutation.cpp -- host-side permutations of [0, N).
// Results per /root/reference/src/Permutation.cpp (random ctor :139-157, inverse :8-27,
// compose :63-96); the O(N^2) scans are replaced by O(N) tables, the rand() draws consumed
// are the same, so a seeded run yields the same permutation.
#include "Permutation.h"

namespace certFHE {

void Permutation::adopt(const uint64_t *src, uint64_t len)
{
    uint64_t *fresh = len ? new uint64_t[len] : nullptr;
    for (uint64_t i = 0; i < len; ++i)
        fresh[i] = src[i];
    delete[] permutation;
    permutation = fresh;
    length = len;
}

Permutation::Permutation() : permutation(nullptr), length(0) {}

Permutation::Permutation(const uint64_t *perm, const uint64_t len) : permutation(nullptr), length(0)
{
    adopt(perm, len);
}

Permutation::Permutation(const uint64_t size) : permutation(nullptr), length(0)
{
    // slot i takes the first draw (mod size) that no earlier slot took
    permutation = size ? new uint64_t[size] : nullptr;
    length = size;
    std::vector<bool> taken(size, false);
    for (uint64_t i = 0; i < size; ++i) {
        uint64_t cand = (uint64_t)rand() % size;
        while (taken[cand])
            cand = (uint64_t)rand() % size;
        taken[cand] = true;
        permutation[i] = cand;
    }
}

Permutation::Permutation(const Context &context) : Permutation(context.getN()) {}

Permutation::Permutation(const Permutation &perm) : permutation(nullptr), length(0)
{
    adopt(perm.permutation, perm.length);
}

Permutation::~Permutation()
{
    delete[] permutation;
    permutation = nullptr;
    length = 0;
}

uint64_t Permutation::getLength() const { return length; }

uint64_t *Permutation::getPermutation() const { return permutation; }

void Permutation::setLength(uint64_t len) { length = len; }

void Permutation::setPermutation(uint64_t *perm, uint64_t len) { adopt(perm, len); }

Permutation &Permutation::operator=(const Permutation &perm)
{
    if (this != &perm)
        adopt(perm.permutation, perm.length);
    return *this;
}

ostream &operator<<(ostream &out, const Permutation &p)
{
    out << "(";
    for (uint64_t i = 0; i < p.length; ++i)
        out << i << " ";
    out << ")" << endl << "(";
    for (uint64_t i = 0; i < p.length; ++i)
        out << p.permutation[i] << " ";
    out << ")" << endl;
    return out;
}

Permutation Permutation::getInverse()
{
    // inverse[i] = smallest j with permutation[j] == i
    std::vector<uint64_t> inv(length, 0);
    for (uint64_t j = length; j-- > 0;)
        if (permutation[j] < length)
            inv[permutation[j]] = j;
    return Permutation(inv.data(), length);
}

Permutation Permutation::operator+(const Permutation &permB) const
{
    if (length != permB.length)
        return Permutation();          // the reference's silent empty result
    std::vector<uint64_t> out(length);
    for (uint64_t i = 0; i < length; ++i)
        out[i] = permutation[permB.permutation[i]];
    return Permutation(out.data(), length);
}

Permutation &Permutation::operator+=(const Permutation &permB)
{
    if (length != permB.length)
        return *this;
    std::vector<uint64_t> out(length);
    for (uint64_t i = 0; i < length; ++i)
        out[i] = permutation[permB.permutation[i]];
    adopt(out.data(), length);
    return *this;
}

} // namespace certFHE
