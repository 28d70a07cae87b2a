// Ciphertext.cpp -- device-resident ciphertext behind the reference's Ciphertext API.
//
// Results follow /root/reference/src/Ciphertext.cpp: operator* = all-pairs AND with the
// left operand as the slow index (:133-179, :231-247), operator+ = concatenation
// (:107-122, :204-229), applyPermutation = permuted FIRST term (:7-82).  The arithmetic runs
// in libcsgn_hip (csgn_mul_uniform / csgn_add_uniform / csgn_permute_uniform); this file only
// marshals handles and the host-side bitlen bookkeeping.
//
// Deliberate differences: operator= re-creates the Context instead of leaving
// it deleted; operator*= does not mismatch delete/delete[]; 64-bit indices throughout.
#include "Ciphertext.h"

#include "runtime.h"


namespace certFHE {

namespace {
// ------------------------------------------------------------------ host mirror storage
// The host mirror behind getValues() is library-owned ("DO NOT DELETE", src/Ciphertext.h:93-102).  A LARGE one
// (1 MiB and up) is a PINNED block out of the runtime's pool (detail::pinnedTake): the copy from HBM is then one DMA
// at the link's rate.  (Round 4's pageable mirror -- 2 MiB-aligned, on transparent huge pages, filled piece by piece
// through the staging buffers -- reached 8.2 GB/s on a 168 MB product, a sixth of the link: the host's memcpy and
// first-touch faults, not the copy.)  A 64-byte head in front of the words says which kind the block is.
const size_t kPinnedMirrorBytes = (size_t)1 << 20;
struct MirrorHead {
    uint64_t pinned_capacity;                      // 0: malloc
    uint64_t pad[7];
};

uint64_t *allocMirror(uint64_t words)
{
    const size_t bytes = (size_t)(words ? words : 1) * 8 + sizeof(MirrorHead);
    MirrorHead *h = nullptr;
    if (bytes >= kPinnedMirrorBytes) {
        size_t capacity = 0;
        h = static_cast<MirrorHead *>(detail::pinnedTake(bytes, &capacity));
        h->pinned_capacity = capacity;
    } else {
        h = static_cast<MirrorHead *>(malloc(bytes));
        if (!h)
            throw std::bad_alloc();
        h->pinned_capacity = 0;
    }
    return reinterpret_cast<uint64_t *>(h + 1);
}

void freeMirror(uint64_t *p)
{
    if (!p)
        return;
    MirrorHead *h = reinterpret_cast<MirrorHead *>(p) - 1;
    if (h->pinned_capacity)
        detail::pinnedGive(h, (size_t)h->pinned_capacity);
    else
        free(h);
}
} // namespace


using detail::DevicePayload;

namespace {

const Context &requireContext(const Context *ctx)
{
    if (!ctx)
        throw std::logic_error("certFHE::Ciphertext: operation needs a Context (object was default-constructed)");
    return *ctx;
}

bool isCanonicalBitlen(const uint64_t *bl, uint64_t len, uint64_t n)
{
    const uint64_t dl = csgn_default_len(n), rem = n % 64;
    if (dl == 0)
        return len == 0;
    for (uint64_t i = 0; i < len; ++i) {
        const uint64_t want = (rem && (i % dl) == dl - 1) ? rem : 64;
        if (bl[i] != want)
            return false;
    }
    return true;
}

} // namespace

// ------------------------------------------------------------------ life cycle

Ciphertext::Ciphertext()
    : len(0), certFHEcontext(nullptr), host_v(nullptr), host_bitlen(nullptr), custom_bitlen(false)
{
}

Ciphertext::Ciphertext(const uint64_t *V, const uint64_t *Bitlen, const uint64_t length,
                       const Context &context)
    : Ciphertext()
{
    certFHEcontext = new Context(context);
    len = length;
    payload = detail::uploadWords(V, length);
    if (Bitlen && length && !isCanonicalBitlen(Bitlen, length, context.getN())) {
        host_bitlen = new uint64_t[length];
        memcpy(host_bitlen, Bitlen, length * sizeof(uint64_t));
        custom_bitlen = true;
    }
}

Ciphertext::Ciphertext(const Ciphertext &c) : Ciphertext() { *this = c; }

Ciphertext::~Ciphertext()
{
    dropMirrors();
    delete certFHEcontext;
    certFHEcontext = nullptr;
    len = 0;
}

void Ciphertext::dropMirrors()
{
    freeMirror(host_v);
    host_v = nullptr;
    delete[] host_bitlen;
    host_bitlen = nullptr;
    custom_bitlen = false;
}

void Ciphertext::publish(const std::shared_ptr<DevicePayload> &p, uint64_t words)
{
    dropMirrors();
    lazy.reset();
    payload = p;
    len = words;
}

// A ciphertext made by a queued operation (detail::deferSmallOp) has no payload yet: the first look at its words
// evaluates the queue.
void Ciphertext::resolve() const
{
    if (lazy) {
        payload = detail::valueOf(lazy);
        lazy.reset();
    }
}

Ciphertext &Ciphertext::operator=(const Ciphertext &c)
{
    if (this == &c)
        return *this;
    dropMirrors();
    payload = c.payload;                       // immutable payload: sharing == deep copy
    lazy = c.lazy;                             // ... and so is sharing the node of a queued operation
    len = c.len;
    if (c.custom_bitlen && c.host_bitlen) {
        host_bitlen = new uint64_t[len ? len : 1];
        memcpy(host_bitlen, c.host_bitlen, len * sizeof(uint64_t));
        custom_bitlen = true;
    }
    Context *fresh = c.certFHEcontext ? new Context(*c.certFHEcontext) : nullptr;
    delete certFHEcontext;
    certFHEcontext = fresh;
    return *this;
}

// ------------------------------------------------------------------ accessors

void Ciphertext::setValues(const uint64_t *V, const uint64_t length)
{
    std::shared_ptr<DevicePayload> p = detail::uploadWords(V, length);
    freeMirror(host_v);
    host_v = nullptr;
    lazy.reset();
    payload = p;
    len = length;
}

void Ciphertext::setBitlen(const uint64_t *Bitlen, const uint64_t length)
{
    delete[] host_bitlen;
    host_bitlen = nullptr;
    custom_bitlen = false;
    len = length;
    const bool canonical = certFHEcontext && isCanonicalBitlen(Bitlen, length, certFHEcontext->getN());
    if (!canonical && length) {
        host_bitlen = new uint64_t[length];
        memcpy(host_bitlen, Bitlen, length * sizeof(uint64_t));
        custom_bitlen = true;
    }
}

void Ciphertext::setContext(const Context &context)
{
    Context *fresh = new Context(context);
    delete certFHEcontext;
    certFHEcontext = fresh;
}

uint64_t Ciphertext::getLen() const { return len; }

Context Ciphertext::getContext() const { return requireContext(certFHEcontext); }

uint64_t *Ciphertext::getValues() const
{
    resolve();
    if (!host_v && len && payload) {
        host_v = allocMirror(len);                                 // pinned when large: the copy is one DMA
        detail::downloadBytes(host_v, payload->ptr, (size_t)len * 8);
    }
    return host_v;
}

uint64_t *Ciphertext::getBitlen() const
{
    if (!host_bitlen && len && certFHEcontext) {
        const uint64_t n = certFHEcontext->getN(), dl = certFHEcontext->getDefaultN(), rem = n % 64;
        host_bitlen = new uint64_t[len];
        for (uint64_t i = 0; i < len; ++i)
            host_bitlen[i] = (rem && (i % dl) == dl - 1) ? rem : 64;   // src/SecretKey.cpp:171-173
    }
    return host_bitlen;
}

uint64_t Ciphertext::getTerms() const
{
    const uint64_t dl = certFHEcontext ? certFHEcontext->getDefaultN() : 0;
    return dl ? len / dl : 0;
}

const uint64_t *Ciphertext::deviceValues() const
{
    resolve();
    return payload ? payload->data() : nullptr;
}

bool Ciphertext::hasCanonicalBitlen() const { return !custom_bitlen; }

long Ciphertext::size()
{
    // same arithmetic as src/Ciphertext.cpp:91-101: three pointers, one length, and the
    // two len-word arrays of the reference's in-memory form
    long bytes = 0;
    bytes += sizeof(Context *) + sizeof(uint64_t) + 2 * sizeof(uint64_t *);
    bytes += (long)(len * 2 * sizeof(uint64_t));
    return bytes;
}

ostream &operator<<(ostream &out, const Ciphertext &c)
{
    const uint64_t *v = c.getValues();
    const uint64_t *bl = c.getBitlen();
    for (uint64_t w = 0; w < c.getLen(); ++w) {
        const uint64_t nbits = bl ? bl[w] : 64;
        for (uint64_t s = 0; s < nbits && s < 64; ++s)
            out << ((v[w] >> (63 - s)) & 1ull);
    }
    out << std::endl;
    return out;
}

// ------------------------------------------------------------------ arithmetic

void Ciphertext::combineBitlen(const Ciphertext &lhs, const Ciphertext &rhs, bool product)
{
    // Only reached when an operand carries a non-canonical Bitlen (4-arg ctor / setBitlen).
    const uint64_t dl = certFHEcontext->getDefaultN();
    if (product) {
        if (!lhs.custom_bitlen)
            return;                                   // pattern comes from the LEFT term only
        const uint64_t *src = lhs.getBitlen();
        host_bitlen = new uint64_t[len ? len : 1];
        if (lhs.len == dl && rhs.len == dl) {
            memcpy(host_bitlen, src, dl * sizeof(uint64_t));            // src/Ciphertext.cpp:140-142
        } else {
            const uint64_t t1 = lhs.len / dl, t2 = rhs.len / dl;
            uint64_t *dst = host_bitlen;
            for (uint64_t i = 0; i < t1; ++i)
                for (uint64_t j = 0; j < t2; ++j, dst += dl)
                    memcpy(dst, src + i * dl, dl * sizeof(uint64_t));   // src/Ciphertext.cpp:165-176
        }
    } else {
        host_bitlen = new uint64_t[len ? len : 1];
        if (lhs.len)
            memcpy(host_bitlen, lhs.getBitlen(), lhs.len * sizeof(uint64_t));          // :215-223
        if (rhs.len)
            memcpy(host_bitlen + lhs.len, rhs.getBitlen(), rhs.len * sizeof(uint64_t));
    }
    custom_bitlen = true;
}

Ciphertext Ciphertext::combine(const Ciphertext &a, const Ciphertext &b, bool product)
{
    const Context &ctx = requireContext(a.certFHEcontext);
    const uint64_t n = ctx.getN(), dl = ctx.getDefaultN();
    Ciphertext out;
    out.certFHEcontext = new Context(ctx);

    // small operands (fresh ciphertexts, short sums): the operation is QUEUED, not launched (runtime.h, "deferred small
    // operations") -- BASELINE config 1 through this API was one 2-3 us launch per 480-byte product
    if (dl && !a.custom_bitlen && !b.custom_bitlen && a.len % dl == 0 && b.len % dl == 0 && a.len && b.len &&
        a.len / dl <= detail::kDeferMaxTerms && b.len / dl <= detail::kDeferMaxTerms && (a.payload || a.lazy) &&
        (b.payload || b.lazy) && requireContext(b.certFHEcontext).getN() == n) {
        std::shared_ptr<detail::LazyNode> node =
            detail::deferSmallOp(product, n, dl, a.len / dl, b.len / dl, a.lazy ? std::shared_ptr<DevicePayload>() : a.payload,
                                 a.lazy, b.lazy ? std::shared_ptr<DevicePayload>() : b.payload, b.lazy);
        if (node) {
            out.lazy = node;
            out.len = product ? (a.len / dl) * (b.len / dl) * dl : a.len + b.len;
            return out;
        }
    }

    if (product) {
        const uint64_t t1 = dl ? a.len / dl : 0, t2 = dl ? b.len / dl : 0;
        const uint64_t newlen = csgn_mul_len(n, a.len, b.len);
        std::shared_ptr<DevicePayload> p = detail::allocWords(newlen);
        if (newlen > t1 * t2 * dl)      // ragged tail the reference leaves unwritten
            detail::check(csgn_memset(p->ptr, 0, (size_t)newlen * 8, detail::stream()), "csgn_memset");
        if (t1 && t2)
            detail::check(csgn_mul_uniform(n, 1, t1, t2, a.deviceValues(), b.deviceValues(), p->data(),
                                           0, detail::stream()),
                          "csgn_mul_uniform");
        out.publish(p, newlen);
    } else {
        const uint64_t newlen = a.len + b.len;
        std::shared_ptr<DevicePayload> p = detail::allocWords(newlen);
        if (dl && a.len % dl == 0 && b.len % dl == 0) {
            if (newlen)
                detail::check(csgn_add_uniform(n, 1, a.len / dl, b.len / dl, a.deviceValues(),
                                               b.deviceValues(), p->data(), detail::stream()),
                              "csgn_add_uniform");
        } else {
            // lengths that are not whole terms: plain device-to-device concatenation
            detail::check(csgn_memcpy_d2d(p->data(), a.deviceValues(), (size_t)a.len * 8, detail::stream()),
                          "csgn_memcpy_d2d");
            detail::check(csgn_memcpy_d2d(p->data() + a.len, b.deviceValues(), (size_t)b.len * 8,
                                          detail::stream()),
                          "csgn_memcpy_d2d");
        }
        out.publish(p, newlen);
    }
    if (a.custom_bitlen || b.custom_bitlen)
        out.combineBitlen(a, b, product);
    return out;
}

Ciphertext Ciphertext::operator+(const Ciphertext &c) const { return combine(*this, c, false); }

Ciphertext Ciphertext::operator*(const Ciphertext &c) const { return combine(*this, c, true); }

Ciphertext &Ciphertext::operator+=(const Ciphertext &c)
{
    Ciphertext r = combine(*this, c, false);
    *this = r;
    return *this;
}

Ciphertext &Ciphertext::operator*=(const Ciphertext &c)
{
    Ciphertext r = combine(*this, c, true);
    *this = r;
    return *this;
}

// ------------------------------------------------------------------ wire format

namespace {
const char kWireMagic[4] = {'C', 'S', 'G', 'N'};
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
const bool kHostIsLittleEndian = true;
#else
const bool kHostIsLittleEndian = false;
#endif

void putU64(std::ostream &out, uint64_t v)
{
    unsigned char b[8];
    for (int i = 0; i < 8; ++i)
        b[i] = (unsigned char)(v >> (8 * i));
    out.write(reinterpret_cast<const char *>(b), 8);
}

uint64_t getU64(std::istream &in)
{
    unsigned char b[8];
    in.read(reinterpret_cast<char *>(b), 8);
    if (!in)
        throw std::runtime_error("certFHE::Ciphertext::deserialize: truncated stream");
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i)
        v |= (uint64_t)b[i] << (8 * i);
    return v;
}
} // namespace

void Ciphertext::serialize(std::ostream &out) const
{
    const Context &ctx = requireContext(certFHEcontext);
    resolve();
    out.write(kWireMagic, 4);
    const unsigned char ver_flags[4] = {1, 0, (unsigned char)(custom_bitlen ? 1 : 0), 0};
    out.write(reinterpret_cast<const char *>(ver_flags), 4);
    putU64(out, ctx.getN());
    putU64(out, ctx.getD());
    putU64(out, len);
    // the words: little-endian on the wire, which is the host's (and the device's) own order, so a
    // piece of HBM goes out as it is -- through the two pinned staging buffers, the DMA of one piece
    // behind the stream write of the previous one.  A ciphertext whose host mirror already exists
    // (getValues() was called) is written from it.
    if (host_v || !payload || !kHostIsLittleEndian) {
        const uint64_t *v = getValues();
        if (kHostIsLittleEndian)
            out.write(reinterpret_cast<const char *>(v), (std::streamsize)(len * 8));
        else
            for (uint64_t i = 0; i < len; ++i)
                putU64(out, v[i]);
    } else {
        struct Sink {
            std::ostream *out;
            static void take(void *ctx, const void *piece, size_t n)
            {
                static_cast<Sink *>(ctx)->out->write(static_cast<const char *>(piece), (std::streamsize)n);
            }
        } sink = {&out};
        detail::downloadStaged(payload->ptr, (size_t)len * 8, &Sink::take, &sink);
    }
    if (custom_bitlen)
        for (uint64_t i = 0; i < len; ++i)
            putU64(out, host_bitlen[i]);
    if (!out)
        throw std::runtime_error("certFHE::Ciphertext::serialize: write failed");
}

uint64_t Ciphertext::serializedSize() const
{
    requireContext(certFHEcontext);
    resolve();
    return 32u + len * 8u * (custom_bitlen ? 2u : 1u);
}

namespace {
void storeU64(unsigned char *p, uint64_t v)
{
    for (int i = 0; i < 8; ++i)
        p[i] = (unsigned char)(v >> (8 * i));
}
uint64_t loadU64(const unsigned char *p)
{
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i)
        v |= (uint64_t)p[i] << (8 * i);
    return v;
}
} // namespace

uint64_t Ciphertext::serializeTo(void *buffer, uint64_t capacity) const
{
    const Context &ctx = requireContext(certFHEcontext);
    const uint64_t need = serializedSize();
    if (!buffer || capacity < need)
        throw std::runtime_error("certFHE::Ciphertext::serializeTo: the buffer is too small");
    unsigned char *b = static_cast<unsigned char *>(buffer);
    memcpy(b, kWireMagic, 4);
    const unsigned char ver_flags[4] = {1, 0, (unsigned char)(custom_bitlen ? 1 : 0), 0};
    memcpy(b + 4, ver_flags, 4);
    storeU64(b + 8, ctx.getN());
    storeU64(b + 16, ctx.getD());
    storeU64(b + 24, len);
    unsigned char *w = b + 32;
    if (host_v || !payload || !kHostIsLittleEndian) {
        const uint64_t *v = getValues();
        if (kHostIsLittleEndian)
            memcpy(w, v, (size_t)len * 8);
        else
            for (uint64_t i = 0; i < len; ++i)
                storeU64(w + 8 * i, v[i]);
    } else if (len) {
        // straight from HBM into the caller's memory (a page-locked buffer: the DMA engine writes it)
        detail::downloadBytes(w, payload->ptr, (size_t)len * 8);    // (returns when the bytes are there, as for getValues())
    }
    if (custom_bitlen)
        for (uint64_t i = 0; i < len; ++i)
            storeU64(w + (len + i) * 8, host_bitlen[i]);
    return need;
}

Ciphertext Ciphertext::deserializeFrom(const void *buffer, uint64_t bytes)
{
    const unsigned char *b = static_cast<const unsigned char *>(buffer);
    if (!b || bytes < 32 || memcmp(b, kWireMagic, 4) != 0)
        throw std::runtime_error("certFHE::Ciphertext::deserializeFrom: not a CSGN stream");
    if (b[4] != 1 || b[5] != 0)
        throw std::runtime_error("certFHE::Ciphertext::deserializeFrom: unsupported version");
    const uint64_t n = loadU64(b + 8), d = loadU64(b + 16), words = loadU64(b + 24);
    const bool with_bitlen = (b[6] & 1) != 0;
    if (n == 0 || d == 0 || words > (1ull << 40) || bytes < 32 + words * 8 * (with_bitlen ? 2u : 1u))
        throw std::runtime_error("certFHE::Ciphertext::deserializeFrom: implausible header or truncated buffer");
    Context ctx(n, d);
    const unsigned char *w = b + 32;
    if (!with_bitlen && kHostIsLittleEndian && words % ctx.getDefaultN() == 0) {
        Ciphertext c;
        c.certFHEcontext = new Context(ctx);
        // (uploadWords copies from the caller's memory and waits: the buffer may be reused at once)
        c.publish(detail::uploadWords(reinterpret_cast<const uint64_t *>(w), words), words);
        return c;
    }
    std::vector<uint64_t> v(words), bl;
    for (uint64_t i = 0; i < words; ++i)
        v[i] = loadU64(w + 8 * i);
    if (with_bitlen) {
        bl.resize(words);
        for (uint64_t i = 0; i < words; ++i)
            bl[i] = loadU64(w + (words + i) * 8);
    }
    return Ciphertext(v.data(), bl.empty() ? nullptr : bl.data(), words, ctx);
}

Ciphertext Ciphertext::deserialize(std::istream &in)
{
    char magic[4];
    unsigned char ver_flags[4];
    in.read(magic, 4);
    in.read(reinterpret_cast<char *>(ver_flags), 4);
    if (!in || memcmp(magic, kWireMagic, 4) != 0)
        throw std::runtime_error("certFHE::Ciphertext::deserialize: not a CSGN stream");
    if (ver_flags[0] != 1 || ver_flags[1] != 0)
        throw std::runtime_error("certFHE::Ciphertext::deserialize: unsupported version");
    const uint64_t n = getU64(in), d = getU64(in), words = getU64(in);
    if (n == 0 || d == 0 || words > (1ull << 40))
        throw std::runtime_error("certFHE::Ciphertext::deserialize: implausible header");
    Context ctx(n, d);
    if (!(ver_flags[2] & 1) && kHostIsLittleEndian && words % ctx.getDefaultN() == 0) {
        // canonical bitlen: the words go from the stream into pinned staging and from there to HBM, the
        // read of one piece behind the DMA of the previous one; no host copy of the whole ciphertext is made
        struct Source {
            std::istream *in;
            static void fill(void *ctx, void *piece, size_t n)
            {
                std::istream &is = *static_cast<Source *>(ctx)->in;
                is.read(static_cast<char *>(piece), (std::streamsize)n);
                if (!is)
                    throw std::runtime_error("certFHE::Ciphertext::deserialize: truncated stream");
            }
        } source = {&in};
        std::shared_ptr<DevicePayload> p = detail::allocWords(words);
        detail::uploadStaged(p->ptr, (size_t)words * 8, &Source::fill, &source);
        Ciphertext c;
        c.certFHEcontext = new Context(ctx);
        c.publish(p, words);
        return c;
    }
    std::vector<uint64_t> v(words), bl;
    for (uint64_t i = 0; i < words; ++i)
        v[i] = getU64(in);
    if (ver_flags[2] & 1) {
        bl.resize(words);
        for (uint64_t i = 0; i < words; ++i)
            bl[i] = getU64(in);
    }
    return Ciphertext(v.data(), bl.empty() ? nullptr : bl.data(), words, ctx);
}

// ------------------------------------------------------------------ permutation

void Ciphertext::applyPermutation_inplace(const Permutation &permutation)
{
    const Context &ctx = requireContext(certFHEcontext);
    const uint64_t n = ctx.getN(), dl = ctx.getDefaultN();
    if (permutation.getLength() < n)
        throw std::invalid_argument("certFHE::Ciphertext::applyPermutation: permutation shorter than N");
    std::vector<uint32_t> p32(n);
    const uint64_t *p = permutation.getPermutation();
    for (uint64_t i = 0; i < n; ++i)
        p32[i] = (uint32_t)p[i];
    if (custom_bitlen) {
        // a Bitlen other than the canonical pattern: (v, bitlen) is a bit stream (src/Ciphertext.cpp:16-31);
        // the result is one term with the canonical Bitlen (:36-47)
        // staging: [bitlen (len words)][perm as u32]
        std::shared_ptr<DevicePayload> stage = detail::allocBytes((size_t)len * 8 + (((size_t)n * 4 + 7) & ~(size_t)7));
        detail::check(csgn_memcpy_h2d(stage->ptr, host_bitlen, (size_t)len * 8, detail::stream()), "csgn_memcpy_h2d");
        detail::check(csgn_memcpy_h2d(stage->data() + len, p32.data(), (size_t)n * 4, detail::stream()),
                      "csgn_memcpy_h2d");
        std::shared_ptr<DevicePayload> work = detail::allocBytes((csgn_bitlen_scratch_bytes(len) + 7) & ~(size_t)7);
        std::shared_ptr<DevicePayload> out = detail::allocWords(dl);
        detail::check(csgn_permute_bitlen(n, len, deviceValues(), stage->data(),
                                          reinterpret_cast<const uint32_t *>(stage->data() + len), out->data(),
                                          work->ptr, detail::stream()),
                      "csgn_permute_bitlen");
        detail::check(csgn_stream_sync(detail::stream()), "csgn_stream_sync");
        publish(out, dl);
        return;
    }
    // one staging block: [perm as u32, padded to 8 bytes]
    std::shared_ptr<DevicePayload> dperm = detail::allocBytes(((size_t)n * 4 + 7) & ~(size_t)7);
    detail::check(csgn_memcpy_h2d(dperm->ptr, p32.data(), (size_t)n * 4, detail::stream()), "csgn_memcpy_h2d");
    std::shared_ptr<DevicePayload> out = detail::allocWords(dl);
    // per_term = 0: the reference keeps only the permuted FIRST term (src/Ciphertext.cpp:33-47)
    detail::check(csgn_permute_uniform(n, 1, dl ? len / dl : 0, 0, deviceValues(),
                                       static_cast<const uint32_t *>(dperm->ptr), out->data(),
                                       detail::stream()),
                  "csgn_permute_uniform");
    detail::check(csgn_stream_sync(detail::stream()), "csgn_stream_sync");   // p32 goes out of scope
    publish(out, dl);
}

Ciphertext Ciphertext::applyPermutation(const Permutation &permutation)
{
    Ciphertext copy(*this);
    copy.applyPermutation_inplace(permutation);
    return copy;
}

} // namespace certFHE
