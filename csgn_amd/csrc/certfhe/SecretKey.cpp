// SecretKey.cpp -- key generation, encrypt and decrypt behind the reference's SecretKey API.
//
// encrypt(): the host walks the positions exactly as /root/reference/src/SecretKey.cpp:35-80
// does, drawing rand() once per non-forced position (plus the slot choice and, when needed,
// the spare bit), and ships the draws to csgn_encrypt_explicit, which applies the forcing
// rules and the MSB-first packing (:175-197) on the device.  Same srand() => same bits.
// decrypt(): csgn_decrypt_uniform, i.e. XOR over terms of AND over the secret positions
// (:104-147), against the key packed into a dL-word mask.
#include "SecretKey.h"

#include <ctime>

#include "runtime.h"

#include <cstring>
#include <vector>

namespace certFHE {

using detail::DevicePayload;

namespace {
const Context &requireContext(const Context *ctx)
{
    if (!ctx)
        throw std::logic_error("certFHE::SecretKey: missing Context");
    return *ctx;
}
} // namespace

// ------------------------------------------------------------------ life cycle

SecretKey::SecretKey(const Context &context) : s(nullptr), length(0), certFHEContext(nullptr)
{
    // Bring the GPU up BEFORE touching libc's generator: HIP runtime start-up draws from
    // rand() itself, which would otherwise disturb a caller's srand(seed) ... encrypt sequence.
    detail::ensureDevice();
    srand((unsigned)time(NULL));               // the reference re-seeds here (src/SecretKey.cpp:311-312)
    certFHEContext = new Context(context);
    const uint64_t d = context.getD(), n = context.getN();
    s = new uint64_t[d ? d : 1];
    length = (long)d;
    uint64_t count = 0;
    while (count < d) {                        // rejection-sample D distinct positions (:322-335)
        const uint64_t cand = (uint64_t)rand() % n;
        if (Helper::exists(s, count, cand))
            continue;
        s[count++] = cand;
    }
}

SecretKey::SecretKey(const SecretKey &other) : s(nullptr), length(0), certFHEContext(nullptr)
{
    certFHEContext = new Context(requireContext(other.certFHEContext));
    if (other.length < 0)
        return;
    length = other.length;
    s = new uint64_t[length ? length : 1];
    for (long i = 0; i < length; ++i)
        s[i] = other.s[i];
}

SecretKey &SecretKey::operator=(const SecretKey &other)
{
    if (this == &other)
        return *this;
    setKey(other.s, (uint64_t)(other.length < 0 ? 0 : other.length));
    if (other.certFHEContext) {
        Context *fresh = new Context(*other.certFHEContext);
        delete certFHEContext;
        certFHEContext = fresh;
    }
    return *this;
}

SecretKey::~SecretKey()
{
    for (long i = 0; i < length; ++i)
        s[i] = 0;                              // zeroise (src/SecretKey.cpp:356-357)
    for (size_t i = 0; i < host_mask.size(); ++i)
        host_mask[i] = 0;
    if (device_mask && device_mask.use_count() == 1 && device_mask->ptr)
        csgn_memset(device_mask->ptr, 0, (size_t)device_mask->words * 8, detail::stream());
    delete[] s;
    s = nullptr;
    length = -1;
    delete certFHEContext;
    certFHEContext = nullptr;
}

// ------------------------------------------------------------------ key access

uint64_t SecretKey::getLength() const { return (uint64_t)length; }

uint64_t *SecretKey::getKey() const { return s; }

void SecretKey::setKey(uint64_t *key, uint64_t len)
{
    uint64_t *fresh = new uint64_t[len ? len : 1];
    for (uint64_t i = 0; i < len; ++i)
        fresh[i] = key[i];
    if (s) {
        for (long i = 0; i < length; ++i)
            s[i] = 0;
        delete[] s;
    }
    s = fresh;
    length = (long)len;
    invalidateMask();
}

long SecretKey::size()
{
    // src/SecretKey.cpp:269-276
    return (long)(sizeof(Context *) + sizeof(long) + sizeof(uint64_t) * (size_t)length);
}

ostream &operator<<(ostream &out, const SecretKey &k)
{
    for (long i = 0; i < k.length; ++i)
        out << k.s[i] << " ";
    out << endl;
    return out;
}

void SecretKey::invalidateMask()
{
    device_mask.reset();
    host_mask.clear();
}

void SecretKey::ensureMask() const
{
    if (device_mask)
        return;
    const Context &ctx = requireContext(certFHEContext);
    host_mask.assign(ctx.getDefaultN(), 0);
    detail::check(csgn_key_mask(ctx.getN(), s, (uint64_t)length, host_mask.data()), "csgn_key_mask");
    device_mask = detail::uploadWords(host_mask.data(), host_mask.size());
}

// ------------------------------------------------------------------ encrypt / decrypt

Ciphertext SecretKey::encrypt(Plaintext &plaintext)
{
    const Context &ctx = requireContext(certFHEContext);
    const uint64_t n = ctx.getN(), d = ctx.getD(), dl = ctx.getDefaultN();
    if ((uint64_t)length != d || d == 0)
        throw std::logic_error("certFHE::SecretKey::encrypt: key length does not match Context D");
    ensureMask();

    // host staging block, mirrored 1:1 on the device:
    //   [0, dl)      per-position random bits, packed MSB-first
    //   word dl      chosen position (u32) | plaintext (u8) << 32 | spare bit (u8) << 40
    std::vector<uint64_t> stage(dl + 1, 0);
    uint64_t *rnd = stage.data();
    const unsigned char bit = plaintext.getValue() & 1;
    uint32_t chosen = 0;
    unsigned char spare = 0;

#define SECRET_AT(i) ((host_mask[(i) >> 6] >> (63 - ((i) & 63))) & 1ull)
#define SET_RND(i) (rnd[(i) >> 6] |= 1ull << (63 - ((i) & 63)))
    if (bit) {
        for (uint64_t i = 0; i < n; ++i)
            if (!SECRET_AT(i) && (rand() % 2))
                SET_RND(i);
    } else {
        chosen = (uint32_t)s[(uint64_t)rand() % d];
        bool others = false, all_one = true;
        for (uint64_t i = 0; i < n; ++i) {
            if (i == chosen)
                continue;
            const int r = rand() % 2;
            if (r)
                SET_RND(i);
            if (SECRET_AT(i)) {
                others = true;
                if (!r)
                    all_one = false;
            }
        }
        if (!(others && all_one))
            spare = (unsigned char)(rand() % 2);
    }
#undef SECRET_AT
#undef SET_RND
    stage[dl] = (uint64_t)chosen | ((uint64_t)bit << 32) | ((uint64_t)spare << 40);

    std::shared_ptr<DevicePayload> dstage = detail::uploadWords(stage.data(), stage.size());
    const unsigned char *meta = reinterpret_cast<const unsigned char *>(dstage->data() + dl);
    std::shared_ptr<DevicePayload> out = detail::allocWords(dl);
    detail::check(csgn_encrypt_explicit(n, d, 1, meta + 4, dstage->data(),
                                        reinterpret_cast<const uint32_t *>(meta), meta + 5,
                                        device_mask->data(), out->data(), detail::stream()),
                  "csgn_encrypt_explicit");
    detail::check(csgn_stream_sync(detail::stream()), "csgn_stream_sync");

    Ciphertext c;
    c.certFHEcontext = new Context(ctx);
    c.publish(out, dl);
    return c;
}

Plaintext SecretKey::decrypt(Ciphertext &ciphertext)
{
    const Context &ctx = requireContext(certFHEContext);
    const uint64_t n = ctx.getN(), dl = ctx.getDefaultN();
    if (!ciphertext.hasCanonicalBitlen()) {
        // a Bitlen other than 64,...,64,N%64 (4-argument constructor / setBitlen): the reference reads
        // (v, bitlen) as a bit stream (src/SecretKey.cpp:110-140); so does csgn_decrypt_bitlen
        const uint64_t words = ciphertext.getLen();
        if (words == 0)
            return Plaintext(0);
        // staging: [bitlen (words)][key indices (length)]
        std::vector<uint64_t> stage(words + (uint64_t)length);
        memcpy(stage.data(), ciphertext.getBitlen(), words * sizeof(uint64_t));
        for (long i = 0; i < length; ++i)
            stage[words + i] = s[i];
        std::shared_ptr<DevicePayload> dstage = detail::uploadWords(stage.data(), stage.size());
        std::shared_ptr<DevicePayload> work = detail::allocBytes((csgn_bitlen_scratch_bytes(words) + 7) & ~(size_t)7);
        void *d_bit = nullptr;
        volatile unsigned char *h_bit = detail::resultSlot(&d_bit);
        detail::check(csgn_decrypt_bitlen(n, (uint64_t)length, words, ciphertext.deviceValues(), dstage->data(),
                                          dstage->data() + words, static_cast<uint8_t *>(d_bit), work->ptr,
                                          detail::stream()),
                      "csgn_decrypt_bitlen");
        // the staging block held the secret indices: wipe it before it returns to the block cache
        detail::check(csgn_memset(dstage->data() + words, 0, (size_t)length * 8, detail::stream()), "csgn_memset");
        detail::syncDevice();
        volatile uint64_t *wipe = stage.data() + words;
        for (long i = 0; i < length; ++i)
            wipe[i] = 0;
        return Plaintext((int)(*h_bit & 1u));
    }
    const uint64_t terms = dl ? ciphertext.getLen() / dl : 0;
    if (terms == 0)
        return Plaintext(0);                   // empty term list XORs to 0
    ensureMask();
    const size_t scratch = (csgn_decrypt_scratch_bytes(1, terms) + 7) & ~(size_t)7;
    std::shared_ptr<DevicePayload> work = detail::allocBytes(scratch);
    // the answer lands in pinned host memory the kernel writes itself: no device-to-host copy
    void *d_bit = nullptr;
    const uint64_t *d_terms = ciphertext.deviceValues();          // (evaluates a queued operation: before the slot is armed)
    volatile unsigned char *h_bit = detail::resultSlot(&d_bit);
    *h_bit = 0xFF;                                                // neither 0 nor 1: the kernel's byte replaces it
    detail::check(csgn_decrypt_uniform(n, 1, terms, d_terms, device_mask->data(),
                                       static_cast<uint8_t *>(d_bit), work->ptr, detail::stream()),
                  "csgn_decrypt_uniform");
    // The byte arrives in host memory with the end of the last kernel; watching it costs a microsecond or two where
    // hipStreamSynchronize costs eight (round 5: 12.8 -> x us per one-term decrypt).  A launch that failed or a byte that
    // does not come falls back to the synchronise, which reports the error.
    detail::awaitByte(h_bit, 0xFF);
    return Plaintext((int)(*h_bit & 1u));
}

// ------------------------------------------------------------------ permutation

void SecretKey::applyPermutation_inplace(const Permutation &permutation)
{
    // position i belongs to the new key iff perm[i] belonged to the old one; the new key
    // comes out sorted (src/SecretKey.cpp:231-250)
    const Context &ctx = requireContext(certFHEContext);
    const uint64_t n = ctx.getN();
    if (permutation.getLength() < n)
        throw std::invalid_argument("certFHE::SecretKey::applyPermutation: permutation shorter than N");
    std::vector<bool> member(n, false);
    for (long i = 0; i < length; ++i)
        if (s[i] < n)
            member[s[i]] = true;
    const uint64_t *p = permutation.getPermutation();
    uint64_t *fresh = new uint64_t[length ? length : 1];
    long count = 0;
    for (uint64_t i = 0; i < n && count < length; ++i)
        if (p[i] < n && member[p[i]])
            fresh[count++] = i;
    for (long i = count; i < length; ++i)
        fresh[i] = 0;
    for (long i = 0; i < length; ++i)
        s[i] = 0;
    delete[] s;
    s = fresh;
    invalidateMask();
}

SecretKey SecretKey::applyPermutation(const Permutation &permutation)
{
    SecretKey copy(*this);
    copy.applyPermutation_inplace(permutation);
    return copy;
}

} // namespace certFHE
