// csgn_decrypt.hip -- decrypt: per-term hit bitmap (__ballot) + parity (__popcll), fused product/sum decrypt.
// Hand-written CDNA4 (gfx950) HIP; shared helpers in csgn_device.h, design notes in DESIGN.md.
#include "csgn_device.h"

namespace csgn {

namespace {

// ---------------------------------------------------------------------------------------
// decrypt, pass 1: one hit bit per term.  Replaces the unpack-everything loops of
// SecretKey::decrypt (src/SecretKey.cpp:110-137): a term "hits" iff all D secret positions
// are 1, i.e. (term & mask) == mask over its dL words.  A workgroup streams 256 consecutive
// terms with coalesced 16-B loads; a unit that misses the mask flags its term in LDS; the
// 256 verdicts leave as four __ballot words, so the bitmap needs no atomics.
// ---------------------------------------------------------------------------------------
template <typename Unit>
__global__ void __launch_bounds__(256) k_term_hits(const Unit *__restrict__ terms,
                                                   const Unit *__restrict__ mask, u64 total_terms,
                                                   u32 U, FastDiv dU, u64 *__restrict__ hits,
                                                   u32 *__restrict__ zero4)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    if (zero4 && blockIdx.x == 0 && threadIdx.x < 4)
        zero4[threadIdx.x] = 0;                     // the ragged pass 2's counters (see LongWork)
    Unit *lmask = reinterpret_cast<Unit *>(smem_raw);                  // U units
    u32 *fail = reinterpret_cast<u32 *>(smem_raw + (size_t)U * sizeof(Unit));   // 256 flags

    const u32 tid = threadIdx.x;
    const u64 term0 = (u64)blockIdx.x * 256u;
    const u32 nterms = (u32)min((u64)256, total_terms - term0);
    const u32 nunits = nterms * U;
    for (u32 k = tid; k < U; k += 256u)
        lmask[k] = mask[k];
    fail[tid] = 0;
    __syncthreads();

    const Unit *base = terms + term0 * U;
#pragma unroll 4
    for (u32 u = tid; u < nunits; u += 256u) {
        const Unit x = base[u];
        const u32 t = csgn_fastdiv(u, dU);
        const u32 k = u - t * U;
        if (!unit_covers(x, lmask[k]))
            fail[t] = 1;                        // benign race: every writer stores 1
    }
    __syncthreads();

    const bool hit = (tid < nterms) && (fail[tid] == 0);
    const u64 b = __ballot(hit);
    if ((tid & (kWave - 1)) == 0)
        hits[(u64)blockIdx.x * 4u + (tid >> 6)] = b;
}

// decrypt, pass 1, fast form.  A 256-thread workgroup makes K passes over K consecutive
// 4 KiB segments (K*256 units = TB whole terms, TB a multiple of 8), all K loads of a lane in
// flight at once.  Every wave ballots "my unit covers the mask" per pass; the K*4 ballots form
// a bit string in LDS in which term t owns bits [t*U, t*U+U); lane t < TB tests them and the
// TB verdicts leave as TB/8 bytes of the hit bitmap.  Segments stay 4 KiB-aligned whatever U
// is (320-thread / 5 KiB workgroups measured 15 % slower at N=1247), and workgroups stay
// short-lived and in address order (see k_and_stream).
template <typename Unit, int K>
__global__ void __launch_bounds__(256) k_term_hits_seg(const Unit *__restrict__ terms,
                                                       const Unit *__restrict__ mask,
                                                       u64 total_units, u32 U, FastDiv dU, u32 TB,
                                                       unsigned char *__restrict__ hits,
                                                       uint8_t *__restrict__ direct_bits,
                                                       u32 *__restrict__ zero4)
{
    __shared__ u64 ok_bits[K * 4 + 1];
    if (zero4 && blockIdx.x == 0 && threadIdx.x < 4)
        zero4[threadIdx.x] = 0;                     // the ragged pass 2's counters (see LongWork)
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    // XCD-contiguous block order: +4.5 % on a pure read stream (tools/rbench.hip)
    const u32 bid = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const u64 g0 = (u64)bid * (256u * K) + tid;

    // Loads are unconditional (addresses clamped into range) so that all K of them, and the K
    // mask units, are in flight together: a load under a divergent `if` makes hipcc wait
    // vmcnt(0) right behind it.
    Unit x[K], mk[K];
    const u64 last = total_units - 1;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const u64 g = g0 + (u32)j * 256u;
        x[j] = terms[g < last ? g : last];
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const u32 local = (u32)j * 256u + tid;
        mk[j] = mask[local - csgn_fastdiv(local, dU) * U];
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const bool ok = (g0 + (u32)j * 256u <= last) && unit_covers(x[j], mk[j]);
        const u64 b = __ballot(ok);
        if (lane == 0)
            ok_bits[j * 4 + wave] = b;
    }
    if (tid == 0)
        ok_bits[K * 4] = 0;                       // pad word for the straddling shift below
    __syncthreads();

    const u64 need = (U >= 64u) ? ~0ull : ((1ull << U) - 1ull);
    for (u32 t = tid; t < TB; t += 256u) {        // TB <= 256 unless U == 1 (TB = 256*K)
        const u32 start = t * U, w = start >> 6, sh = start & 63u;
        u64 v = ok_bits[w] >> sh;
        if (sh)
            v |= ok_bits[w + 1] << (64u - sh);
        const bool hit = (v & need) == need;
        if (direct_bits) {                        // single-term ciphertexts: the verdict IS the plaintext
            const u64 term = (u64)bid * TB + t;
            if (term * U < total_units)
                direct_bits[term] = hit ? 1 : 0;
            continue;
        }
        const u64 hb = __ballot(hit);             // lanes past TB are not in this iteration
        if (lane == 0) {
            const u32 first = t;                  // first term of this wave's group
            const u32 nbits = min(64u, TB - first);
            unsigned char *dst = hits + (u64)bid * (TB >> 3) + (first >> 3);
            if (nbits == 64u)
                *reinterpret_cast<u64 *>(dst) = hb;
            else if (nbits == 32u)
                *reinterpret_cast<u32 *>(dst) = (u32)hb;
            else if (nbits == 16u)
                *reinterpret_cast<unsigned short *>(dst) = (unsigned short)hb;
            else
                *dst = (unsigned char)hb;
        }
    }
}

// decrypt, pass 2: XOR over the terms of each ciphertext = parity of the popcount of its
// bit range (src/SecretKey.cpp:139, `_dec = (dec + _dec) % 2`).  G lanes per ciphertext:
// 1 for small term counts, a whole wave (with a __ballot/__popcll fold) for large ones.
// MODE 0: every ciphertext; 1 (ragged batches): only those of at most kLongTerms terms; a longer one
// gets a slot (its index, a parity word) and one work-list entry per 65 536-term chunk of its bit
// range, which k_hits_parity_chunks folds -- one huge ciphertext among many small ones costs neither
// a wave per small one nor one lonely workgroup for the big one (round 2: a single workgroup walked
// the 128 KB bitmap of a 1 M-term ciphertext, 3.1 TB/s for the batch).
constexpr u64 kLongTerms = 4096;
constexpr u64 kChunkTerms = 65536;

// work area of the ragged pass 2 (after the hit bitmap): [n_long, n_entries, pad, pad][slot -> ciphertext:
// max_long u32][slot parity: max_long u32][slot chunks left: max_long u32][entries: (slot << 32) | chunk,
// max_entries u64].  The counters are zeroed by pass 1 (one launch less than a memset).
struct LongWork {
    u32 *counters;
    u32 *slot_ct;
    u32 *slot_par;
    u32 *slot_left;
    u64 *entries;
};

template <int G, int MODE>
__global__ void __launch_bounds__(256) k_hits_parity(const u64 *__restrict__ hits,
                                                     const u64 *__restrict__ off, u64 T, u64 batch,
                                                     uint8_t *__restrict__ bits, LongWork work)
{
    const u64 gid = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 b = gid / G;
    const u32 lane = (u32)(gid % G);
    if (b >= batch)
        return;
    const u64 s = off ? off[b] : b * T;
    const u64 e = off ? off[b + 1] : s + T;
    if (MODE == 1 && e - s > kLongTerms) {
        const u32 slot = atomicAdd(work.counters, 1u);
        work.slot_ct[slot] = (u32)b;
        work.slot_par[slot] = 0;
        const u32 nch = (u32)((e - s + kChunkTerms - 1) / kChunkTerms);
        work.slot_left[slot] = nch;
        const u32 base = atomicAdd(work.counters + 1, nch);
        for (u32 c = 0; c < nch; ++c)
            work.entries[base + c] = ((u64)slot << 32) | c;
        return;
    }
    u32 par = 0;
    if (e > s) {
        const u64 w0 = s >> 6, w1 = (e - 1) >> 6;
        for (u64 w = w0 + lane; w <= w1; w += G) {
            u64 x = hits[w];
            if (w == w0)
                x &= ~0ull << (s & 63);
            if (w == w1 && (e & 63))
                x &= (1ull << (e & 63)) - 1;
            par ^= (u32)__popcll(x);
        }
    }
    if (G == 1) {
        bits[b] = (uint8_t)(par & 1u);
    } else {
        const u64 odd = __ballot(par & 1u);
        if (lane == 0)
            bits[b] = (uint8_t)(__popcll(odd) & 1);
    }
}

// The long ciphertexts of a ragged batch, chunk by chunk: workgroups stride over the work list (its
// length is on the device), fold 1024 bitmap words each (five independent loads per lane) and XOR one
// bit into the ciphertext's slot; the workgroup that folds a ciphertext's LAST chunk writes its bit.
__global__ void __launch_bounds__(256) k_hits_parity_chunks(const u64 *__restrict__ hits,
                                                            const u64 *__restrict__ off, LongWork work,
                                                            uint8_t *__restrict__ bits)
{
    __shared__ u32 wave_par[4];
    const u32 n = work.counters[1];
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        const u64 ent = work.entries[i];
        const u32 slot = (u32)(ent >> 32), c = (u32)ent;
        const u32 b = work.slot_ct[slot];
        const u64 s = off[b], e = off[b + 1];
        const u64 cs = s + (u64)c * kChunkTerms, ce = min(e, cs + kChunkTerms);
        const u64 w0 = cs >> 6, w1 = (ce - 1) >> 6;            // at most 1025 words
        u32 par = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const u64 w = w0 + threadIdx.x + (u32)k * 256u;
            u64 x = hits[min(w, w1)];                           // unconditional: all five in flight
            if (w > w1)
                x = 0;
            if (w == w0)
                x &= ~0ull << (cs & 63);
            if (w == w1 && (ce & 63))
                x &= (1ull << (ce & 63)) - 1;
            par ^= (u32)__popcll(x);
        }
        const u64 odd = __ballot(par & 1u);
        if ((threadIdx.x & (kWave - 1)) == 0)
            wave_par[threadIdx.x >> 6] = (u32)__popcll(odd) & 1u;
        __syncthreads();
        if (threadIdx.x == 0) {
            if (wave_par[0] ^ wave_par[1] ^ wave_par[2] ^ wave_par[3])
                atomicXor(work.slot_par + slot, 1u);
            __threadfence();                                    // my parity is in before I count myself out
            if (atomicSub(work.slot_left + slot, 1u) == 1u)     // the last chunk of this ciphertext
                bits[b] = (uint8_t)(atomicOr(work.slot_par + slot, 0u) & 1u);
        }
        __syncthreads();
    }
}

// decrypt, pass 2 for LONG uniform ciphertexts: a ciphertext's bit range is cut into chunks of
// 65536 terms (1024 bitmap words); one workgroup per (ciphertext, chunk) folds its chunk with
// __popcll / __ballot and XORs one bit into a per-ciphertext word, so a single 1M-term
// ciphertext is reduced by 16 workgroups instead of one wave.
__global__ void __launch_bounds__(256) k_hits_parity_chunked(const u64 *__restrict__ hits, u64 T,
                                                             u32 chunks, u32 *__restrict__ partial)
{
    __shared__ u32 wave_par[4];
    const u32 b = blockIdx.x / chunks, c = blockIdx.x - b * chunks;
    const u64 s = (u64)b * T, e = s + T;
    const u64 cs = s + (u64)c * 65536u;
    const u64 ce = min(e, cs + 65536u);
    u32 par = 0;
    if (ce > cs) {
        const u64 w0 = cs >> 6, w1 = (ce - 1) >> 6;
        for (u64 w = w0 + threadIdx.x; w <= w1; w += 256u) {
            u64 x = hits[w];
            if (w == w0)
                x &= ~0ull << (cs & 63);
            if (w == w1 && (ce & 63))
                x &= (1ull << (ce & 63)) - 1;
            par ^= (u32)__popcll(x);
        }
    }
    const u64 odd = __ballot(par & 1u);
    if ((threadIdx.x & (kWave - 1)) == 0)
        wave_par[threadIdx.x >> 6] = (u32)__popcll(odd) & 1u;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 p = wave_par[0] ^ wave_par[1] ^ wave_par[2] ^ wave_par[3];
        if (p)
            atomicXor(partial + b, 1u);
    }
}

__global__ void __launch_bounds__(256) k_partial_to_bits(const u32 *__restrict__ partial, u64 batch,
                                                         uint8_t *__restrict__ bits)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < batch)
        bits[i] = (uint8_t)(partial[i] & 1u);
}

} // namespace

// ------------------------------------------------------------------------------ public

static size_t decrypt_bitmap_bytes(u64 total_terms)
{
    return (size_t)((total_terms + 255) / 256) * 32u + 64u;
}

// ragged batches: at most this many ciphertexts are "long" (> kLongTerms terms) ...
static u64 decrypt_max_long(u64 batch, u64 total_terms)
{
    const u64 by_terms = total_terms / (kLongTerms + 1) + 1;
    return batch < by_terms ? batch : by_terms;
}
// ... and their chunk lists have at most this many entries
static u64 decrypt_max_entries(u64 batch, u64 total_terms)
{
    return decrypt_max_long(batch, total_terms) + total_terms / kChunkTerms + 1;
}

size_t decrypt_scratch_bytes(u64 batch, u64 total_terms)
{
    // [hit bitmap, one bit per term | pad] then the larger of
    //   uniform long ciphertexts: one u32 partial parity per ciphertext
    //   ragged batches: LongWork (4 counters, 2 u32 per long slot, one u64 per chunk entry)
    const size_t uniform = (size_t)batch * 4u + 16u;
    const size_t ragged = 16u + (size_t)decrypt_max_long(batch, total_terms) * 12u + 8u +
                          (size_t)decrypt_max_entries(batch, total_terms) * 8u;
    return decrypt_bitmap_bytes(total_terms) + (uniform > ragged ? uniform : ragged);
}

hipError_t decrypt(u64 n_bits, u64 batch, u64 terms_uniform, u64 total_terms, const u64 *terms,
                   const u64 *off, const u64 *mask, uint8_t *bits, void *scratch, hipStream_t s, u64 max_terms)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0)
        return hipSuccess;
    // a CSR batch in which every ciphertext has exactly max_terms terms (the caller's bound is met with equality) IS a
    // uniform batch: no offsets are read, single-term ciphertexts get their plaintext bytes from pass 1 itself
    if (off && max_terms != 0 && batch * max_terms == total_terms) {
        off = nullptr;
        terms_uniform = max_terms;
    }
    u64 *hits = reinterpret_cast<u64 *>(scratch);
    LongWork work = {};
    if (off) {
        unsigned char *wbase = reinterpret_cast<unsigned char *>(scratch) + decrypt_bitmap_bytes(total_terms);
        const u64 max_long = decrypt_max_long(batch, total_terms);
        work.counters = reinterpret_cast<u32 *>(wbase);
        work.slot_ct = work.counters + 4;
        work.slot_par = work.slot_ct + max_long;
        work.slot_left = work.slot_par + max_long;
        work.entries = reinterpret_cast<u64 *>(wbase + ((16u + max_long * 12u + 7u) & ~(size_t)7u));
    }
    u32 *zero4 = work.counters;                     // pass 1 zeroes them; nullptr for uniform batches
    if (total_terms) {
        const bool wide = (dL % 2 == 0) && aligned16(terms) && aligned16(mask);
        const u32 U = (u32)(wide ? dL / 2 : dL);
        const u64 blocks64 = (total_terms + 255) / 256;
        if (blocks64 > kMaxBlocks256)
            return hipErrorInvalidValue;
        const u32 blocks = (u32)blocks64;
        const FastDiv dU = csgn_fastdiv_make(U);
        // fast form: K 4-KiB segments per workgroup holding TB whole terms (TB % 8 == 0);
        // term sizes that need K > 8 (or U > 64) use the looping form
        int k_seg = 0;
        if (U <= 64u)
            for (int k = 1; k <= 8; ++k)
                if ((256u * k) % U == 0 && ((256u * k) / U) % 8u == 0) {
                    k_seg = k;
                    break;
                }
        if (k_seg && tune(TUNE_DEC_LOOP) == 0) {
            const u32 tb = 256u * k_seg / U;
            const u64 nblk = (total_terms + tb - 1) / tb;
            if (nblk > kMaxBlocks256)
                return hipErrorInvalidValue;
            unsigned char *hb = reinterpret_cast<unsigned char *>(scratch);
            const u64 tu = total_terms * U;
            // fresh (single-term) ciphertexts in a uniform batch: pass 1 writes the plaintext
            // bytes itself and pass 2 is skipped
            uint8_t *direct = (!off && terms_uniform == 1) ? bits : nullptr;
#define CSGN_HITS_SEG(K)                                                                              \
    do {                                                                                              \
        if (wide)                                                                                     \
            k_term_hits_seg<unit16, K><<<(u32)nblk, 256, 0, s>>>(                                     \
                reinterpret_cast<const unit16 *>(terms), reinterpret_cast<const unit16 *>(mask), tu, U, \
                dU, tb, hb, direct, zero4);                                                           \
        else                                                                                          \
            k_term_hits_seg<unit8, K><<<(u32)nblk, 256, 0, s>>>(terms, mask, tu, U, dU, tb, hb, direct, zero4); \
    } while (0)
            switch (k_seg) {
            case 1: CSGN_HITS_SEG(1); break;
            case 2: CSGN_HITS_SEG(2); break;
            case 3: CSGN_HITS_SEG(3); break;
            case 4: CSGN_HITS_SEG(4); break;
            case 5: CSGN_HITS_SEG(5); break;
            case 6: CSGN_HITS_SEG(6); break;
            case 7: CSGN_HITS_SEG(7); break;
            default: CSGN_HITS_SEG(8); break;
            }
#undef CSGN_HITS_SEG
            if (direct)
                return hipGetLastError();
        } else if (wide)
            k_term_hits<unit16><<<blocks, 256, (size_t)U * 16 + 1024, s>>>(
                reinterpret_cast<const unit16 *>(terms), reinterpret_cast<const unit16 *>(mask),
                total_terms, U, dU, hits, zero4);
        else
            k_term_hits<unit8><<<blocks, 256, (size_t)U * 8 + 1024, s>>>(terms, mask, total_terms, U,
                                                                        dU, hits, zero4);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return e;
    }
    if (off) {
        // ragged: short ciphertexts one lane each; the long ones are queued chunk by chunk
        if (batch >= (1ull << 32))
            return hipErrorInvalidValue;
        // (the work area's counters were zeroed by pass 1; with no terms at all nothing is queued)
        k_hits_parity<1, 1><<<ceil_div_u64(batch, 256), 256, 0, s>>>(hits, off, 0, batch, bits, work);
        if (total_terms > kLongTerms && (max_terms == 0 || max_terms > kLongTerms)) {       // (a bound says: nothing is queued)
            const u64 max_entries = decrypt_max_entries(batch, total_terms);
            k_hits_parity_chunks<<<(u32)std::min<u64>(8192, max_entries), 256, 0, s>>>(hits, off, work, bits);
        }
    } else if (terms_uniform <= kLongTerms) {
        k_hits_parity<1, 0><<<ceil_div_u64(batch, 256), 256, 0, s>>>(hits, nullptr, terms_uniform, batch, bits, LongWork{});
    } else if (batch * ((terms_uniform + 65535) / 65536) <= kMaxBlocks256) {
        // long uniform ciphertexts: chunked fold + one atomicXor per (ciphertext, chunk)
        u32 *partial = reinterpret_cast<u32 *>(reinterpret_cast<unsigned char *>(scratch) +
                                               decrypt_bitmap_bytes(total_terms));
        // (a kernel, as every zero fill a circuit may capture: csgn_device.h, zero_words; the bitmap is 32-byte
        // granular + 64, so `partial` is 8-byte aligned and (batch + 1) / 2 words cover batch u32 and stay inside
        // the scratch, which ends 16 bytes past them)
        hipError_t e = zero_words(reinterpret_cast<u64 *>(partial), (batch + 1) / 2, s);
        if (e != hipSuccess)
            return e;
        const u32 chunks = (u32)((terms_uniform + 65535) / 65536);
        k_hits_parity_chunked<<<(u32)(batch * chunks), 256, 0, s>>>(hits, terms_uniform, chunks, partial);
        k_partial_to_bits<<<ceil_div_u64(batch, 256), 256, 0, s>>>(partial, batch, bits);
    } else {
        if (batch * 64 > kMaxBlocks256 * 256u)
            return hipErrorInvalidValue;
        k_hits_parity<64, 0><<<ceil_div_u64(batch * 64, 256), 256, 0, s>>>(hits, nullptr, terms_uniform, batch, bits, LongWork{});
    }
    return hipGetLastError();
}

// dec(a*b) = dec(a) & dec(b) and dec(a+b) = dec(a) ^ dec(b): a product term L_i & R_j covers
// the key mask iff both factors do, so the number of hitting product terms is
// hits(L)*hits(R) and its parity the AND of the parities; concatenation adds the counts.
// The 168 MB product of a 1024x1024 pair is therefore never materialised when only its
// plaintext is wanted: 2 x 160 KB are read instead.
__global__ void __launch_bounds__(256) k_combine_bits(const uint8_t *__restrict__ a,
                                                      const uint8_t *__restrict__ b, u64 n, int is_product,
                                                      uint8_t *__restrict__ out)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n)
        out[i] = is_product ? (a[i] & b[i] & 1u) : ((a[i] ^ b[i]) & 1u);
}

hipError_t combine_bits(const uint8_t *a, const uint8_t *b, u64 n, bool is_product, uint8_t *out, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    if (n > kMaxBlocks256 * 256u)
        return hipErrorInvalidValue;
    k_combine_bits<<<ceil_div_u64(n, 256), 256, 0, s>>>(a, b, n, is_product ? 1 : 0, out);
    return hipGetLastError();
}

size_t decrypt_combined_scratch_bytes(u64 batch, u64 t1, u64 t2)
{
    const size_t pad = 256;
    return decrypt_scratch_bytes(batch, batch * t1) + decrypt_scratch_bytes(batch, batch * t2) +
           2 * (((size_t)batch + pad - 1) / pad * pad) + 4 * pad;
}

hipError_t decrypt_combined(u64 n_bits, u64 batch, u64 t1, u64 t2, const u64 *L, const u64 *R,
                            const u64 *mask, bool is_product, uint8_t *bits, void *scratch, hipStream_t s)
{
    if (batch == 0)
        return hipSuccess;
    const size_t pad = 256;
    auto up = [&](size_t x) { return (x + pad - 1) / pad * pad; };
    unsigned char *base = reinterpret_cast<unsigned char *>(scratch);
    base = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(base)));
    unsigned char *s1 = base;
    unsigned char *s2 = s1 + up(decrypt_scratch_bytes(batch, batch * t1));
    uint8_t *b1 = s2 + up(decrypt_scratch_bytes(batch, batch * t2));
    uint8_t *b2 = b1 + up(batch);
    hipError_t e = decrypt(n_bits, batch, t1, batch * t1, L, nullptr, mask, b1, s1, s);
    if (e != hipSuccess)
        return e;
    e = decrypt(n_bits, batch, t2, batch * t2, R, nullptr, mask, b2, s2, s);
    if (e != hipSuccess)
        return e;
    k_combine_bits<<<ceil_div_u64(batch, 256), 256, 0, s>>>(b1, b2, batch, is_product ? 1 : 0, bits);
    return hipGetLastError();
}

} // namespace csgn
