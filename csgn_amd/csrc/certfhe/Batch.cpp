// Batch.cpp -- uniform device-resident batches over the C ABI (extension, see Batch.h).
#include "Batch.h"

#include "runtime.h"

#include <algorithm>
#include <cstring>

namespace certFHE {

using detail::DevicePayload;

namespace {
void requireSameShape(const Context &a, const Context &b, uint64_t ca, uint64_t cb)
{
    if (a.getN() != b.getN() || ca != cb)
        throw std::invalid_argument("certFHE::CiphertextBatch: operands differ in N or element count");
}
} // namespace

CiphertextBatch::CiphertextBatch(const Context &c, uint64_t count, uint64_t terms)
    : count_(count), terms_(terms), ctx(c)
{
    payload = detail::allocWords(count * terms * c.getDefaultN());
}

const uint64_t *CiphertextBatch::deviceValues() const { return payload ? payload->data() : nullptr; }

uint64_t CiphertextBatch::termsOf(uint64_t i) const
{
    if (i >= count_)
        throw std::out_of_range("certFHE::CiphertextBatch::termsOf");
    return uniform() ? terms_ : offsets_[i + 1] - offsets_[i];
}

uint64_t CiphertextBatch::maxTerms() const
{
    if (uniform())
        return terms_;
    uint64_t m = 0;
    for (uint64_t i = 0; i < count_; ++i)
        m = std::max(m, offsets_[i + 1] - offsets_[i]);
    return m;
}

const uint64_t *CiphertextBatch::deviceOffsets() const
{
    if (!d_offsets_) {
        if (uniform()) {
            std::vector<uint64_t> off(count_ + 1);
            for (uint64_t i = 0; i <= count_; ++i)
                off[i] = i * terms_;
            d_offsets_ = detail::uploadWords(off.data(), off.size());
        } else {
            d_offsets_ = detail::uploadWords(offsets_.data(), offsets_.size());
        }
    }
    return d_offsets_->data();
}

CiphertextBatch CiphertextBatch::compact() const
{
    const uint64_t total = totalTerms(), n = ctx.getN(), dl = ctx.getDefaultN();
    CiphertextBatch out(ctx, count_, 0);
    out.terms_ = terms_;
    if (count_ == 0 || total == 0) {
        out.offsets_ = offsets_;
        return out;
    }
    out.payload = detail::allocWords(total * dl);                  // room for "nothing cancels"
    std::shared_ptr<DevicePayload> off_out = detail::allocWords(count_ + 1);
    const size_t need = csgn_compact_scratch_bytes(n, count_, total);
    if (need == 0)
        throw std::runtime_error("certFHE::CiphertextBatch::compact: context not supported by csgn_compact_ragged");
    std::shared_ptr<DevicePayload> work = detail::allocBytes(need);
    detail::check(csgn_compact_ragged(n, count_, total, maxTerms(), deviceValues(), deviceOffsets(),
                                      out.payload->data(), off_out->data(), work->ptr, detail::stream()),
                  "csgn_compact_ragged");
    std::vector<uint64_t> off(count_ + 1);
    detail::downloadBytes(off.data(), off_out->data(), (count_ + 1) * 8);      // synchronises
    bool same = true;
    for (uint64_t i = 1; i < count_ && same; ++i)
        same = off[i + 1] - off[i] == off[1] - off[0];
    if (same) {                                                    // dense CSR of equal sizes IS the uniform layout
        out.terms_ = off[1] - off[0];
    } else {
        out.terms_ = 0;
        out.offsets_.swap(off);
        out.d_offsets_ = off_out;
    }
    return out;
}

namespace {
// encrypt `bits` under `key` with the keyed device generator (csgn_encrypt_keyed)
void encryptInto(uint64_t n_bits, uint64_t d, const uint64_t *key_indices, const uint64_t *device_mask,
                 const std::vector<unsigned char> &bits, const csgn_rng &rng, uint64_t first, uint64_t *d_out)
{
    const uint64_t count = bits.size();
    // staging: [key indices (d words)][plaintext bytes]
    std::vector<uint64_t> stage(d + (count + 7) / 8, 0);
    for (uint64_t i = 0; i < d; ++i)
        stage[i] = key_indices[i];
    memcpy(stage.data() + d, bits.data(), count);
    std::shared_ptr<DevicePayload> dstage = detail::uploadWords(stage.data(), stage.size());
    detail::check(csgn_encrypt_keyed(n_bits, d, count, first,
                                     reinterpret_cast<const uint8_t *>(dstage->data() + d), dstage->data(),
                                     device_mask, &rng, d_out, detail::stream()),
                  "csgn_encrypt_keyed");
    // the staging block held the secret indices: wipe it before it returns to the block cache, and
    // the host copy too (~SecretKey zeroises its own copies the same way, src/SecretKey.cpp:354-372)
    detail::check(csgn_memset(dstage->ptr, 0, stage.size() * 8, detail::stream()), "csgn_memset");
    detail::check(csgn_stream_sync(detail::stream()), "csgn_stream_sync");
    volatile uint64_t *wipe = stage.data();
    for (size_t i = 0; i < stage.size(); ++i)
        wipe[i] = 0;
}
} // namespace

CiphertextBatch CiphertextBatch::encrypt(const SecretKey &key, const std::vector<unsigned char> &bits)
{
    if (!key.certFHEContext)
        throw std::logic_error("certFHE::CiphertextBatch::encrypt: key has no Context");
    key.ensureMask();
    CiphertextBatch out(*key.certFHEContext, bits.size(), 1);
    if (bits.empty())
        return out;
    csgn_rng rng;                      // fresh 256-bit generator key + nonce from the OS for every batch
    detail::check(csgn_rng_from_os(&rng, 8), "csgn_rng_from_os");
    encryptInto(key.certFHEContext->getN(), (uint64_t)key.length, key.s, key.device_mask->data(), bits, rng, 0,
                out.payload->data());
    volatile uint32_t *wipe = rng.key;
    for (int i = 0; i < 8; ++i)
        wipe[i] = 0;
    return out;
}

CiphertextBatch CiphertextBatch::encrypt(const SecretKey &key, const std::vector<unsigned char> &bits,
                                         uint64_t seed, uint64_t first_ciphertext)
{
    if (!key.certFHEContext)
        throw std::logic_error("certFHE::CiphertextBatch::encrypt: key has no Context");
    key.ensureMask();
    CiphertextBatch out(*key.certFHEContext, bits.size(), 1);
    if (bits.empty())
        return out;
    csgn_rng rng;
    detail::check(csgn_rng_from_seed(&rng, seed, 8), "csgn_rng_from_seed");
    encryptInto(key.certFHEContext->getN(), (uint64_t)key.length, key.s, key.device_mask->data(), bits, rng,
                first_ciphertext, out.payload->data());
    return out;
}

CiphertextBatch CiphertextBatch::pack(const std::vector<Ciphertext> &items)
{
    if (items.empty())
        throw std::invalid_argument("certFHE::CiphertextBatch::pack: empty list");
    const Context c = items[0].getContext();
    const uint64_t dl = c.getDefaultN(), terms = items[0].getTerms();
    CiphertextBatch out(c, items.size(), terms);
    for (size_t i = 0; i < items.size(); ++i) {
        if (items[i].getTerms() != terms || items[i].getLen() != terms * dl || !items[i].hasCanonicalBitlen())
            throw std::invalid_argument("certFHE::CiphertextBatch::pack: elements must share one shape");
        detail::check(csgn_memcpy_d2d(out.payload->data() + i * terms * dl, items[i].deviceValues(),
                                      (size_t)terms * dl * 8, detail::stream()),
                      "csgn_memcpy_d2d");
    }
    return out;
}

CiphertextBatch CiphertextBatch::operator*(const CiphertextBatch &rhs) const
{
    requireSameShape(ctx, rhs.ctx, count_, rhs.count_);
    if (!uniform() || !rhs.uniform()) {                            // CSR kernels: plan, then multiply
        CiphertextBatch out(ctx, count_, 0);
        if (count_ == 0)
            return out;
        std::shared_ptr<DevicePayload> off_out = detail::allocWords(count_ + 1);
        uint64_t plan[4] = {0, 0, 0, 0};
        detail::check(csgn_mul_ragged_plan(count_, deviceOffsets(), rhs.deviceOffsets(), off_out->data(), plan,
                                           detail::stream()),
                      "csgn_mul_ragged_plan");
        out.payload = detail::allocWords(plan[0] * ctx.getDefaultN());
        if (plan[0])
            detail::check(csgn_mul_ragged(ctx.getN(), count_, deviceValues(), deviceOffsets(), rhs.deviceValues(),
                                          rhs.deviceOffsets(), out.payload->data(), off_out->data(), plan[1], plan[2],
                                          plan[0], detail::stream()),
                          "csgn_mul_ragged");
        out.offsets_.resize(count_ + 1);
        for (uint64_t i = 0, run = 0; i <= count_; ++i) {
            out.offsets_[i] = run;
            if (i < count_)
                run += termsOf(i) * rhs.termsOf(i);
        }
        out.d_offsets_ = off_out;
        return out;
    }
    CiphertextBatch out(ctx, count_, terms_ * rhs.terms_);
    detail::check(csgn_mul_uniform(ctx.getN(), count_, terms_, rhs.terms_, deviceValues(),
                                   rhs.deviceValues(), out.payload->data(), 0, detail::stream()),
                  "csgn_mul_uniform");
    return out;
}

CiphertextBatch CiphertextBatch::operator+(const CiphertextBatch &rhs) const
{
    requireSameShape(ctx, rhs.ctx, count_, rhs.count_);
    if (!uniform() || !rhs.uniform()) {
        CiphertextBatch out(ctx, count_, 0);
        if (count_ == 0)
            return out;
        const uint64_t total = totalTerms() + rhs.totalTerms();
        std::shared_ptr<DevicePayload> off_out = detail::allocWords(count_ + 1);
        out.payload = detail::allocWords(total * ctx.getDefaultN());
        detail::check(csgn_add_ragged_bounded(ctx.getN(), count_, maxTerms(), rhs.maxTerms(), deviceValues(), deviceOffsets(),
                                              rhs.deviceValues(), rhs.deviceOffsets(), out.payload->data(), off_out->data(), total,
                                              detail::stream()),
                      "csgn_add_ragged_bounded");
        out.offsets_.resize(count_ + 1);
        for (uint64_t i = 0, run = 0; i <= count_; ++i) {
            out.offsets_[i] = run;
            if (i < count_)
                run += termsOf(i) + rhs.termsOf(i);
        }
        out.d_offsets_ = off_out;
        return out;
    }
    CiphertextBatch out(ctx, count_, terms_ + rhs.terms_);
    detail::check(csgn_add_uniform(ctx.getN(), count_, terms_, rhs.terms_, deviceValues(),
                                   rhs.deviceValues(), out.payload->data(), detail::stream()),
                  "csgn_add_uniform");
    return out;
}

namespace {
// N uint32 permutation entries in HBM (two per uploaded word)
std::shared_ptr<DevicePayload> uploadPermutation(const Permutation &p, uint64_t n, const char *who)
{
    if (p.getLength() < n)
        throw std::invalid_argument(std::string(who) + ": permutation shorter than N");
    std::vector<uint64_t> packed((n + 1) / 2, 0);
    const uint64_t *src = p.getPermutation();
    for (uint64_t i = 0; i < n; ++i)
        packed[i / 2] |= (uint64_t)(uint32_t)src[i] << (32 * (i % 2));
    return detail::uploadWords(packed.data(), packed.size());
}
} // namespace

CiphertextBatch CiphertextBatch::applyPermutation(const Permutation &permutation) const
{
    if (!uniform())
        throw std::logic_error("certFHE::CiphertextBatch::applyPermutation: uniform batches only");
    CiphertextBatch out(ctx, count_, 1);
    if (count_ == 0)
        return out;
    std::shared_ptr<DevicePayload> d = uploadPermutation(permutation, ctx.getN(), "certFHE::CiphertextBatch::applyPermutation");
    detail::check(csgn_permute_uniform(ctx.getN(), count_, terms_, 0, deviceValues(),
                                       reinterpret_cast<const uint32_t *>(d->data()), out.payload->data(),
                                       detail::stream()),
                  "csgn_permute_uniform");
    detail::syncDevice();               // the permutation table is released on return
    return out;
}

std::vector<unsigned char> CiphertextBatch::decrypt(const SecretKey &key) const
{
    std::vector<unsigned char> bits(count_, 0);
    if (count_ == 0)
        return bits;
    key.ensureMask();
    const size_t scratch = (csgn_decrypt_scratch_bytes(count_, totalTerms()) + 255) & ~(size_t)255;
    std::shared_ptr<DevicePayload> work = detail::allocBytes(scratch + count_);
    uint8_t *d_bits = static_cast<uint8_t *>(work->ptr) + scratch;
    if (uniform())
        detail::check(csgn_decrypt_uniform(ctx.getN(), count_, terms_, deviceValues(), key.device_mask->data(),
                                           d_bits, work->ptr, detail::stream()),
                      "csgn_decrypt_uniform");
    else
        detail::check(csgn_decrypt_ragged_bounded(ctx.getN(), count_, totalTerms(), maxTerms(), deviceValues(), deviceOffsets(),
                                                  key.device_mask->data(), d_bits, work->ptr, detail::stream()),
                      "csgn_decrypt_ragged_bounded");
    detail::downloadBytes(bits.data(), d_bits, count_);
    return bits;
}

static std::vector<unsigned char> fusedDecrypt(const CiphertextBatch &a, const CiphertextBatch &b,
                                               const uint64_t *mask, bool product)
{
    std::vector<unsigned char> bits(a.size(), 0);
    if (a.size() == 0)
        return bits;
    const size_t scratch =
        (csgn_decrypt_combined_scratch_bytes(a.size(), a.terms(), b.terms()) + 255) & ~(size_t)255;
    std::shared_ptr<DevicePayload> work = detail::allocBytes(scratch + a.size());
    uint8_t *d_bits = static_cast<uint8_t *>(work->ptr) + scratch;
    const uint64_t n = a.context().getN();
    detail::check(product ? csgn_decrypt_product_uniform(n, a.size(), a.terms(), b.terms(), a.deviceValues(),
                                                         b.deviceValues(), mask, d_bits, work->ptr,
                                                         detail::stream())
                          : csgn_decrypt_sum_uniform(n, a.size(), a.terms(), b.terms(), a.deviceValues(),
                                                     b.deviceValues(), mask, d_bits, work->ptr,
                                                     detail::stream()),
                  "csgn_decrypt_{product,sum}_uniform");
    detail::downloadBytes(bits.data(), d_bits, a.size());
    return bits;
}

std::vector<unsigned char> CiphertextBatch::decryptProduct(const CiphertextBatch &rhs,
                                                           const SecretKey &key) const
{
    requireSameShape(ctx, rhs.ctx, count_, rhs.count_);
    if (!uniform() || !rhs.uniform())
        throw std::logic_error("certFHE::CiphertextBatch::decryptProduct: uniform batches only");
    key.ensureMask();
    return fusedDecrypt(*this, rhs, key.device_mask->data(), true);
}

std::vector<unsigned char> CiphertextBatch::decryptSum(const CiphertextBatch &rhs, const SecretKey &key) const
{
    requireSameShape(ctx, rhs.ctx, count_, rhs.count_);
    if (!uniform() || !rhs.uniform())
        throw std::logic_error("certFHE::CiphertextBatch::decryptSum: uniform batches only");
    key.ensureMask();
    return fusedDecrypt(*this, rhs, key.device_mask->data(), false);
}

Ciphertext CiphertextBatch::at(uint64_t i) const
{
    if (i >= count_)
        throw std::out_of_range("certFHE::CiphertextBatch::at");
    const uint64_t dl = ctx.getDefaultN(), words = termsOf(i) * dl;
    const uint64_t first = (uniform() ? i * terms_ : offsets_[i]) * dl;
    std::shared_ptr<DevicePayload> p = detail::allocWords(words);
    if (words)
        detail::check(csgn_memcpy_d2d(p->data(), deviceValues() + first, (size_t)words * 8, detail::stream()),
                      "csgn_memcpy_d2d");
    Ciphertext c;
    c.certFHEcontext = new Context(ctx);
    c.publish(p, words);
    return c;
}

// ------------------------------------------------------------------ BatchCircuit

BatchCircuit::BatchCircuit(const Context &context, uint64_t count)
    : handle(nullptr), ctx(context), count_(count), next_first(0)
{
    detail::ensureDevice();
    detail::check(csgn_circuit_create(ctx.getN(), count, &handle), "csgn_circuit_create");
}

BatchCircuit::~BatchCircuit()
{
    if (handle) {
        detail::syncDevice();           // a launch of the graph may still be running
        csgn_circuit_destroy(handle);
    }
}

unsigned BatchCircuit::input(uint64_t terms)
{
    uint32_t id = 0;
    detail::check(csgn_circuit_input(handle, terms, &id), "csgn_circuit_input");
    return id;
}

unsigned BatchCircuit::encryptInput(const SecretKey &key)
{
    if (!key.certFHEContext || key.certFHEContext->getN() != ctx.getN())
        throw std::invalid_argument("certFHE::BatchCircuit::encryptInput: key and circuit differ in N");
    key.ensureMask();
    masks.push_back(key.device_mask);
    const uint64_t d = (uint64_t)key.length;
    // one block: [key indices (d words)][plaintext bytes, zero until setPlain()]
    std::vector<uint64_t> stage(d + (count_ + 7) / 8, 0);
    for (uint64_t i = 0; i < d; ++i)
        stage[i] = key.s[i];
    std::shared_ptr<DevicePayload> blk = detail::uploadWords(stage.data(), stage.size());
    volatile uint64_t *wipe = stage.data();
    for (uint64_t i = 0; i < d; ++i)
        wipe[i] = 0;
    masks.push_back(blk);
    csgn_rng rng;
    detail::check(csgn_rng_from_os(&rng, 8), "csgn_rng_from_os");
    uint32_t id = 0;
    detail::check(csgn_circuit_encrypt(handle, d, reinterpret_cast<const uint8_t *>(blk->data() + d), blk->data(),
                                       key.device_mask->data(), &rng, next_first, &id),
                  "csgn_circuit_encrypt");
    volatile uint32_t *wk = rng.key;
    for (int i = 0; i < 8; ++i)
        wk[i] = 0;
    next_first += (uint64_t)1 << 40;                // every encrypt input draws from its own range
    plains.push_back(std::make_pair((unsigned)id, blk));
    return id;
}

unsigned BatchCircuit::encryptProduct(const SecretKey &key, unsigned *bits_id)
{
    if (!key.certFHEContext || key.certFHEContext->getN() != ctx.getN())
        throw std::invalid_argument("certFHE::BatchCircuit::encryptProduct: key and circuit differ in N");
    key.ensureMask();
    masks.push_back(key.device_mask);
    const uint64_t d = (uint64_t)key.length, pw = (count_ + 7) / 8;
    // one block: [key indices (d words)][plaintext bytes of a][plaintext bytes of b], zero until setPlainPair()
    std::vector<uint64_t> stage(d + 2 * pw, 0);
    for (uint64_t i = 0; i < d; ++i)
        stage[i] = key.s[i];
    std::shared_ptr<DevicePayload> blk = detail::uploadWords(stage.data(), stage.size());
    volatile uint64_t *wipe = stage.data();
    for (uint64_t i = 0; i < d; ++i)
        wipe[i] = 0;
    masks.push_back(blk);
    csgn_rng ra, rb;                                  // two independent generator keys from the OS
    detail::check(csgn_rng_from_os(&ra, 8), "csgn_rng_from_os");
    detail::check(csgn_rng_from_os(&rb, 8), "csgn_rng_from_os");
    uint32_t id = 0, bid = 0;
    const uint8_t *pa = reinterpret_cast<const uint8_t *>(blk->data() + d);
    detail::check(csgn_circuit_encrypt_mul(handle, d, pa, pa + pw * 8, blk->data(), key.device_mask->data(), &ra, &rb,
                                           next_first, &id, bits_id ? &bid : nullptr),
                  "csgn_circuit_encrypt_mul");
    volatile uint32_t *wa = ra.key, *wb = rb.key;
    for (int i = 0; i < 8; ++i)
        wa[i] = wb[i] = 0;
    next_first += (uint64_t)1 << 40;
    pair_plains.push_back(std::make_pair((unsigned)id, blk));
    if (bits_id)
        *bits_id = bid;
    return id;
}

void BatchCircuit::setPlainPair(unsigned product, const std::vector<unsigned char> &a, const std::vector<unsigned char> &b)
{
    if (a.size() != count_ || b.size() != count_)
        throw std::invalid_argument("certFHE::BatchCircuit::setPlainPair: one bit of each operand per element expected");
    const uint64_t pw = (count_ + 7) / 8;
    for (size_t i = 0; i < pair_plains.size(); ++i)
        if (pair_plains[i].first == product) {
            const uint64_t d_words = pair_plains[i].second->words - 2 * pw;
            detail::check(csgn_memcpy_h2d(pair_plains[i].second->data() + d_words, a.data(), (size_t)count_, detail::stream()),
                          "csgn_memcpy_h2d");
            detail::check(csgn_memcpy_h2d(pair_plains[i].second->data() + d_words + pw, b.data(), (size_t)count_, detail::stream()),
                          "csgn_memcpy_h2d");
            detail::check(csgn_stream_sync(detail::stream()), "csgn_stream_sync");
            return;
        }
    throw std::invalid_argument("certFHE::BatchCircuit::setPlainPair: not a fused product of this circuit");
}

void BatchCircuit::setPlain(unsigned encrypted_input, const std::vector<unsigned char> &bits)
{
    if (bits.size() != count_)
        throw std::invalid_argument("certFHE::BatchCircuit::setPlain: one bit per element expected");
    for (size_t i = 0; i < plains.size(); ++i)
        if (plains[i].first == encrypted_input) {
            // the key indices occupy the first words of the block; the bytes follow
            const uint64_t d_words = plains[i].second->words - (count_ + 7) / 8;
            detail::check(csgn_memcpy_h2d(plains[i].second->data() + d_words, bits.data(), (size_t)count_,
                                          detail::stream()),
                          "csgn_memcpy_h2d");
            detail::check(csgn_stream_sync(detail::stream()), "csgn_stream_sync");
            return;
        }
    throw std::invalid_argument("certFHE::BatchCircuit::setPlain: not an encrypted input of this circuit");
}

unsigned BatchCircuit::add(unsigned a, unsigned b)
{
    uint32_t id = 0;
    detail::check(csgn_circuit_add(handle, a, b, &id), "csgn_circuit_add");
    return id;
}

unsigned BatchCircuit::mul(unsigned a, unsigned b)
{
    uint32_t id = 0;
    detail::check(csgn_circuit_mul(handle, a, b, &id), "csgn_circuit_mul");
    return id;
}

unsigned BatchCircuit::compact(unsigned a)
{
    uint32_t id = 0;
    detail::check(csgn_circuit_compact(handle, a, &id), "csgn_circuit_compact");
    return id;
}

unsigned BatchCircuit::permute(unsigned a, const Permutation &p)
{
    std::shared_ptr<DevicePayload> d = uploadPermutation(p, ctx.getN(), "certFHE::BatchCircuit::permute");
    masks.push_back(d);                 // the graph holds the pointer: keep the block alive
    uint32_t id = 0;
    detail::check(csgn_circuit_permute(handle, a, reinterpret_cast<const uint32_t *>(d->data()), &id),
                  "csgn_circuit_permute");
    return id;
}

unsigned BatchCircuit::decrypt(unsigned a, const SecretKey &key)
{
    key.ensureMask();
    masks.push_back(key.device_mask);   // the graph holds the pointer: keep the block alive
    uint32_t id = 0;
    detail::check(csgn_circuit_decrypt(handle, a, key.device_mask->data(), &id), "csgn_circuit_decrypt");
    return id;
}

void BatchCircuit::optimize(unsigned passes)
{
    detail::check(csgn_circuit_optimize(handle, passes), "csgn_circuit_optimize");
}

void BatchCircuit::keep(unsigned value) { detail::check(csgn_circuit_output(handle, value), "csgn_circuit_output"); }

uint64_t BatchCircuit::blockBytes() const { return csgn_circuit_block_bytes(handle); }

void BatchCircuit::build() { detail::check(csgn_circuit_build(handle), "csgn_circuit_build"); }

void BatchCircuit::set(unsigned input, const CiphertextBatch &batch)
{
    const uint64_t terms = csgn_circuit_value_terms(handle, input);
    uint64_t *dst = csgn_circuit_value(handle, input);
    if (!dst || batch.size() != count_ || batch.terms() != terms || batch.context().getN() != ctx.getN())
        throw std::invalid_argument("certFHE::BatchCircuit::set: shape mismatch or circuit not built");
    detail::check(csgn_memcpy_d2d(dst, batch.deviceValues(), (size_t)(count_ * terms * ctx.getDefaultN() * 8),
                                  detail::stream()),
                  "csgn_memcpy_d2d");
}

void BatchCircuit::run() { detail::check(csgn_circuit_run(handle, detail::stream()), "csgn_circuit_run"); }

CiphertextBatch BatchCircuit::value(unsigned id) const
{
    const uint64_t terms = csgn_circuit_value_terms(handle, id);
    const uint64_t *src = csgn_circuit_value(handle, id);
    if (!src)
        throw std::invalid_argument("certFHE::BatchCircuit::value: no such value, circuit not built, or a value a compiled circuit did not keep()");
    const uint64_t *d_off = csgn_circuit_value_offsets(handle, id);
    if (terms == 0 && !d_off)
        throw std::invalid_argument("certFHE::BatchCircuit::value: no such value, circuit not built, or a value a compiled circuit did not keep()");
    if (d_off) {                                    // ragged (static shapes, or data-dependent ones behind compact())
        CiphertextBatch out(ctx, count_, 0);
        out.offsets_.resize(count_ + 1);
        detail::downloadBytes(out.offsets_.data(), d_off, (count_ + 1) * 8);      // synchronises
        const uint64_t words = out.offsets_.back() * ctx.getDefaultN();
        out.payload = detail::allocWords(words);
        if (words)
            detail::check(csgn_memcpy_d2d(out.payload->ptr, src, (size_t)words * 8, detail::stream()), "csgn_memcpy_d2d");
        return out;
    }
    CiphertextBatch out(ctx, count_, terms);
    detail::check(csgn_memcpy_d2d(out.payload->ptr, src, (size_t)(count_ * terms * ctx.getDefaultN() * 8),
                                  detail::stream()),
                  "csgn_memcpy_d2d");
    return out;
}

std::vector<unsigned char> BatchCircuit::bits(unsigned bits_id) const
{
    const uint8_t *src = csgn_circuit_bits(handle, bits_id);
    if (!src)
        throw std::invalid_argument("certFHE::BatchCircuit::bits: no such result or circuit not built");
    std::vector<unsigned char> out(count_, 0);
    detail::downloadBytes(out.data(), src, count_);
    return out;
}

} // namespace certFHE
