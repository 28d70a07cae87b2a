/*
 * ref_driver.cpp -- extern "C" entry points around the REAL reference library.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together with the
 * reference's own sources *where they lie* (/root/reference/src/*.cpp, never copied) by
 * oracle/Makefile into oracle/_ref/libcsgn_ref.so.  It drives the reference purely
 * through its public certFHE:: API (src/certFHE.h) so that
 *   - tests can pin oracle/csgn_oracle.c against the genuine implementation, and
 *   - tests/golden/gen_golden.py can produce the committed known-answer vectors, and
 *   - bench.py can time the genuine Ciphertext::operator* as cpu_baseline.kind="reference".
 *
 * Determinism recipe: construct SecretKey (re-seeds from the clock),
 * overwrite the key with setKey(), THEN srand(seed), then encrypt.
 */
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <sstream>
#include <string>
#include <vector>

#include "certFHE.h"

using certFHE::Ciphertext;
using certFHE::Context;
using certFHE::Permutation;
using certFHE::Plaintext;
using certFHE::SecretKey;

namespace {

void copy_out(const Ciphertext &c, uint64_t *out_v, uint64_t *out_bitlen)
{
    uint64_t n = c.getLen();
    if (out_v)
        std::memcpy(out_v, c.getValues(), n * sizeof(uint64_t));
    if (out_bitlen)
        std::memcpy(out_bitlen, c.getBitlen(), n * sizeof(uint64_t));
}

} // namespace

extern "C" {

uint64_t ref_default_len(uint64_t n, uint64_t d)
{
    Context ctx(n, d);
    return ctx.getDefaultN();
}

uint64_t ref_context_s(uint64_t n, uint64_t d)
{
    Context ctx(n, d);
    return ctx.getS();
}

/* Key generation as the reference does it (seeded from the clock inside the ctor).
 * Returns time(NULL) sampled just before and just after, so a test can recover the
 * seed that was used. */
void ref_keygen(uint64_t n, uint64_t d, uint64_t *key_out, int64_t *t_before, int64_t *t_after)
{
    Context ctx(n, d);
    /* The reference's sampler tests membership against the WHOLE new uint64_t[d] array, whose
     * tail is still uninitialised (SecretKey.cpp:318-327 + Helpers.cpp:18-24): stale heap words
     * that happen to be < n reject draws at random.  Hand the allocator a few recycled chunks
     * of that size filled with values no draw can equal, so the run is reproducible. */
    {
        uint64_t *pad[8];
        for (int i = 0; i < 8; ++i) {
            pad[i] = new uint64_t[d];
            volatile uint64_t *q = pad[i]; /* keep the fill: it is dead to the optimiser */
            for (uint64_t j = 0; j < d; ++j)
                q[j] = ~0ull;
        }
        for (int i = 0; i < 8; ++i)
            delete[] pad[i];
    }
    *t_before = (int64_t)time(NULL);
    SecretKey sk(ctx);
    *t_after = (int64_t)time(NULL);
    std::memcpy(key_out, sk.getKey(), d * sizeof(uint64_t));
}

/* Encrypt `count` bits in sequence from ONE srand(seed) stream.  out_v receives
 * count*dL words, out_bitlen (optional) likewise. */
void ref_encrypt_seq(uint64_t n, uint64_t d, const uint64_t *key, unsigned seed,
                     const uint8_t *bits, uint64_t count, uint64_t *out_v, uint64_t *out_bitlen)
{
    Context ctx(n, d);
    SecretKey sk(ctx);
    sk.setKey(const_cast<uint64_t *>(key), d);
    uint64_t dl = ctx.getDefaultN();
    srand(seed);
    for (uint64_t i = 0; i < count; ++i) {
        Plaintext p(bits[i]);
        Ciphertext c = sk.encrypt(p);
        copy_out(c, out_v + i * dl, out_bitlen ? out_bitlen + i * dl : nullptr);
    }
}

uint64_t ref_mul(uint64_t n, uint64_t d,
                 const uint64_t *v1, const uint64_t *bl1, uint64_t len1,
                 const uint64_t *v2, const uint64_t *bl2, uint64_t len2,
                 uint64_t *out_v, uint64_t *out_bitlen)
{
    Context ctx(n, d);
    Ciphertext a(v1, bl1, len1, ctx);
    Ciphertext b(v2, bl2, len2, ctx);
    Ciphertext c = a * b;
    copy_out(c, out_v, out_bitlen);
    return c.getLen();
}

/* operator*= variant (src/Ciphertext.cpp:283-304). */
uint64_t ref_mul_inplace(uint64_t n, uint64_t d,
                         const uint64_t *v1, const uint64_t *bl1, uint64_t len1,
                         const uint64_t *v2, const uint64_t *bl2, uint64_t len2,
                         uint64_t *out_v, uint64_t *out_bitlen)
{
    Context ctx(n, d);
    Ciphertext a(v1, bl1, len1, ctx);
    Ciphertext b(v2, bl2, len2, ctx);
    a *= b;
    copy_out(a, out_v, out_bitlen);
    return a.getLen();
}

uint64_t ref_add(uint64_t n, uint64_t d,
                 const uint64_t *v1, const uint64_t *bl1, uint64_t len1,
                 const uint64_t *v2, const uint64_t *bl2, uint64_t len2,
                 uint64_t *out_v, uint64_t *out_bitlen)
{
    Context ctx(n, d);
    Ciphertext a(v1, bl1, len1, ctx);
    Ciphertext b(v2, bl2, len2, ctx);
    Ciphertext c = a + b;
    copy_out(c, out_v, out_bitlen);
    return c.getLen();
}

uint64_t ref_add_inplace(uint64_t n, uint64_t d,
                         const uint64_t *v1, const uint64_t *bl1, uint64_t len1,
                         const uint64_t *v2, const uint64_t *bl2, uint64_t len2,
                         uint64_t *out_v, uint64_t *out_bitlen)
{
    Context ctx(n, d);
    Ciphertext a(v1, bl1, len1, ctx);
    Ciphertext b(v2, bl2, len2, ctx);
    a += b;
    copy_out(a, out_v, out_bitlen);
    return a.getLen();
}

unsigned ref_decrypt(uint64_t n, uint64_t d, const uint64_t *key,
                     const uint64_t *v, const uint64_t *bl, uint64_t len)
{
    Context ctx(n, d);
    SecretKey sk(ctx);
    sk.setKey(const_cast<uint64_t *>(key), d);
    Ciphertext c(v, bl, len, ctx);
    Plaintext p = sk.decrypt(c);
    return p.getValue();
}

void ref_perm_random(uint64_t size, unsigned seed, uint64_t *perm_out)
{
    srand(seed);
    Permutation p(size);
    std::memcpy(perm_out, p.getPermutation(), size * sizeof(uint64_t));
}

void ref_perm_inverse(const uint64_t *perm, uint64_t size, uint64_t *inv_out)
{
    Permutation p(perm, size);
    Permutation q = p.getInverse();
    std::memcpy(inv_out, q.getPermutation(), size * sizeof(uint64_t));
}

/* Returns the length of the composed permutation (0 on length mismatch). */
uint64_t ref_perm_compose(const uint64_t *a, uint64_t len_a, const uint64_t *b, uint64_t len_b,
                          uint64_t *out)
{
    Permutation pa(a, len_a), pb(b, len_b);
    Permutation pc = pa + pb;
    if (pc.getLength())
        std::memcpy(out, pc.getPermutation(), pc.getLength() * sizeof(uint64_t));
    return pc.getLength();
}

uint64_t ref_permute_ciphertext(uint64_t n, uint64_t d, const uint64_t *perm,
                                const uint64_t *v, const uint64_t *bl, uint64_t len,
                                uint64_t *out_v, uint64_t *out_bitlen)
{
    Context ctx(n, d);
    Permutation p(perm, n);
    Ciphertext c(v, bl, len, ctx);
    Ciphertext r = c.applyPermutation(p);
    copy_out(r, out_v, nullptr);
    /* the reference writes result_bitlen[div] even when N%64==0 (one word past the end,
     * src/Ciphertext.cpp:45); only hand back the in-bounds part */
    if (out_bitlen)
        std::memcpy(out_bitlen, r.getBitlen(), r.getLen() * sizeof(uint64_t));
    return r.getLen();
}

void ref_permute_key(uint64_t n, uint64_t d, const uint64_t *perm, const uint64_t *key,
                     uint64_t *key_out)
{
    Context ctx(n, d);
    SecretKey sk(ctx);
    sk.setKey(const_cast<uint64_t *>(key), d);
    Permutation p(perm, n);
    SecretKey r = sk.applyPermutation(p);
    std::memcpy(key_out, r.getKey(), d * sizeof(uint64_t));
}

/* Time `iters` evaluations of the genuine Ciphertext::operator* on fixed operands.
 * Returns seconds (wall, steady clock).  sink receives a value derived from the last
 * product so the work is observable. */
double ref_time_mul(uint64_t n, uint64_t d,
                    const uint64_t *v1, uint64_t len1, const uint64_t *v2, uint64_t len2,
                    uint64_t iters, uint64_t *sink)
{
    Context ctx(n, d);
    uint64_t dl = ctx.getDefaultN();
    uint64_t rem = n % 64;
    uint64_t *bl1 = new uint64_t[len1 ? len1 : 1];
    uint64_t *bl2 = new uint64_t[len2 ? len2 : 1];
    for (uint64_t i = 0; i < len1; ++i)
        bl1[i] = (rem && (i % dl) == dl - 1) ? rem : 64;
    for (uint64_t i = 0; i < len2; ++i)
        bl2[i] = (rem && (i % dl) == dl - 1) ? rem : 64;
    Ciphertext a(v1, bl1, len1, ctx);
    Ciphertext b(v2, bl2, len2, ctx);
    uint64_t acc = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t it = 0; it < iters; ++it) {
        Ciphertext c = a * b;
        acc += c.getValues()[c.getLen() - 1] + c.getLen();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (sink)
        *sink = acc;
    delete[] bl1;
    delete[] bl2;
    return std::chrono::duration<double>(t1 - t0).count();
}

/* Time `iters` evaluations of the genuine SecretKey::decrypt on one ciphertext. */
double ref_time_decrypt(uint64_t n, uint64_t d, const uint64_t *key,
                        const uint64_t *v, uint64_t len, uint64_t iters, uint64_t *sink)
{
    Context ctx(n, d);
    uint64_t dl = ctx.getDefaultN();
    uint64_t rem = n % 64;
    uint64_t *bl = new uint64_t[len ? len : 1];
    for (uint64_t i = 0; i < len; ++i)
        bl[i] = (rem && (i % dl) == dl - 1) ? rem : 64;
    SecretKey sk(ctx);
    sk.setKey(const_cast<uint64_t *>(key), d);
    Ciphertext c(v, bl, len, ctx);
    uint64_t acc = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t it = 0; it < iters; ++it)
        acc += sk.decrypt(c).getValue();
    auto t1 = std::chrono::steady_clock::now();
    if (sink)
        *sink = acc;
    delete[] bl;
    return std::chrono::duration<double>(t1 - t0).count();
}

/* BASELINE config 5 through the public class API: x0 = Enc(b0); level l odd: x = x + Enc(b),
 * even: x = x * (Enc(b) + Enc(b')); `levels` levels, `iters` repetitions on the same fresh
 * inputs.  Returns the seconds spent in the arithmetic plus one final decrypt per repetition
 * (encryption of the inputs is outside the timed region); *terms_out = terms of the result,
 * *bit_out = its decryption. */
double ref_time_circuit(uint64_t n, uint64_t d, unsigned levels, uint64_t iters, unsigned seed,
                        uint64_t *terms_out, unsigned *bit_out)
{
    Context ctx(n, d);
    SecretKey sk(ctx);
    srand(seed);
    std::vector<Ciphertext> in;
    for (unsigned i = 0; i < 2 * levels + 2; ++i) {
        Plaintext p((int)((i * 7 + 3) % 5 < 3));
        in.push_back(sk.encrypt(p));
    }
    uint64_t terms = 0;
    unsigned bit = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t it = 0; it < iters; ++it) {
        Ciphertext x = in[0];
        unsigned k = 1;
        for (unsigned level = 1; level <= levels; ++level) {
            // compound forms: the reference's copy assignment leaves the target without a
            // Context, so `x = x * y` would crash it
            if (level % 2) {
                x += in[k];
                k += 1;
            } else {
                x *= (in[k] + in[k + 1]);
                k += 2;
            }
        }
        bit = (unsigned)sk.decrypt(x).getValue();
        terms = x.getLen() / ctx.getDefaultN();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (terms_out)
        *terms_out = terms;
    if (bit_out)
        *bit_out = bit;
    return std::chrono::duration<double>(t1 - t0).count();
}

/* Text form (operator<<) of one object.  kind: 0 Ciphertext (a=v, b=bitlen, len words),
 * 1 SecretKey (a=key, len=d), 2 Context, 3 Plaintext (len = the bit), 4 Permutation (a=perm,
 * len=size).  Copies at most cap bytes into buf and returns the full length. */
uint64_t ref_text(int kind, uint64_t n, uint64_t d, const uint64_t *a, const uint64_t *b,
                  uint64_t len, char *buf, uint64_t cap)
{
    std::ostringstream os;
    Context ctx(n, d);
    if (kind == 0) {
        Ciphertext c(a, b, len, ctx);
        os << c;
    } else if (kind == 1) {
        SecretKey sk(ctx);
        sk.setKey(const_cast<uint64_t *>(a), len);
        os << sk;
    } else if (kind == 2) {
        os << ctx;
    } else if (kind == 3) {
        Plaintext p((int)len);
        os << p;
    } else if (kind == 4) {
        Permutation p(a, len);
        os << p;
    }
    const std::string t = os.str();
    if (buf && cap)
        std::memcpy(buf, t.data(), t.size() < cap ? t.size() : cap);
    return t.size();
}

} // extern "C"
