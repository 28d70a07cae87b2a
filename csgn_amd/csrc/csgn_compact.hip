// csgn_compact.hip -- EXTENSION: mod-2 compaction of term lists (not reference behaviour).
// Hand-written CDNA4 (gfx950) HIP; shared helpers in csgn_device.h, design notes in DESIGN.md.
#include "csgn_device.h"

namespace csgn {

// ------------------------------------------------------------------------------ public

// ---------------------------------------------------------------------------------------
// EXTENSION (not reference behaviour): mod-2 compaction of term lists.
// Decryption XORs over terms, so a term occurring an even number of times contributes
// nothing and one occurring an odd number of times contributes once: each ciphertext is
// rewritten as its distinct odd-multiplicity terms, in order of first occurrence.  The
// reference never does this (add is pure concatenation), so it is opt-in and never runs on a
// parity path.  Method: per-ciphertext open-addressing hash table in HBM keyed by a 64-bit
// hash of the term; representative = smallest index with that key; every term is compared
// in full against its representative (a hash collision between different terms just keeps
// the colliding term unmerged, it can never merge unequal terms); parity by atomicXor;
// survivors are compacted with a prefix sum.  Deterministic output.
// ---------------------------------------------------------------------------------------
struct CompactView {
    u64 *keys;      // 2 slots per term
    u32 *rep;       // per slot: smallest term index holding the key
    u32 *parity;    // per slot: multiplicity mod 2 of the representative's value
    u32 *slot_of;   // per term: its slot (global slot index)
    u32 *keep;      // per term: 1 = survives
    u64 *scan;      // per term: exclusive prefix sum of keep (+1 total at [total])
};

__device__ inline u64 term_hash(const u64 *t, u32 dL)
{
    u64 h = 0x243F6A8885A308D3ull;
    for (u32 k = 0; k < dL; ++k)
        h = csgn_splitmix64(h ^ t[k]);
    return h | 1ull;                                 // 0 is the empty-slot marker
}

// which ciphertext owns global term index g (CSR offsets, batch >= 1)
__device__ inline u32 owner_of_term(const u64 *off, u32 batch, u64 g)
{
    u32 lo = 0, hi = batch;                          // off[lo] <= g < off[hi]
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (off[mid] <= g)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256) k_compact_insert(const u64 *__restrict__ terms,
                                                        const u64 *__restrict__ off, u32 batch,
                                                        u64 total, u32 dL, CompactView v)
{
    const u64 g = (u64)blockIdx.x * 256u + threadIdx.x;
    if (g >= total)
        return;
    const u32 b = owner_of_term(off, batch, g);
    const u64 base = 2 * off[b], nslots = 2 * (off[b + 1] - off[b]);
    const u64 key = term_hash(terms + g * dL, dL);
    u64 slot = key % nslots;
    for (;;) {
        const u64 old = atomicCAS(reinterpret_cast<unsigned long long *>(v.keys + base + slot), 0ull, key);
        if (old == 0ull || old == key)
            break;
        slot = slot + 1 == nslots ? 0 : slot + 1;    // table is at most half full: terminates
    }
    v.slot_of[g] = (u32)(base + slot);
    atomicMin(v.rep + base + slot, (u32)(g - off[b]));
}

__global__ void __launch_bounds__(256) k_compact_match(const u64 *__restrict__ terms,
                                                       const u64 *__restrict__ off, u32 batch,
                                                       u64 total, u32 dL, CompactView v)
{
    const u64 g = (u64)blockIdx.x * 256u + threadIdx.x;
    if (g >= total)
        return;
    const u32 b = owner_of_term(off, batch, g);
    const u32 slot = v.slot_of[g];
    const u64 r = off[b] + v.rep[slot];              // global index of the representative
    bool same = true;
    if (r != g) {
        const u64 *x = terms + g * dL, *y = terms + r * dL;
        for (u32 k = 0; k < dL; ++k)
            same = same && (x[k] == y[k]);
    }
    if (same)
        atomicXor(v.parity + slot, 1u);
    v.keep[g] = same ? 2u : 1u;                      // 2 = decided by the slot parity, 1 = collision survivor
}

__global__ void __launch_bounds__(256) k_compact_decide(const u64 *__restrict__ off, u32 batch, u64 total,
                                                        CompactView v)
{
    const u64 g = (u64)blockIdx.x * 256u + threadIdx.x;
    if (g >= total)
        return;
    const u32 b = owner_of_term(off, batch, g);
    const u32 slot = v.slot_of[g];
    u32 k = v.keep[g];
    if (k == 2u)
        k = (off[b] + v.rep[slot] == g && (v.parity[slot] & 1u)) ? 1u : 0u;
    v.keep[g] = k;
}

// exclusive scan of keep[] (one workgroup, chunked) + compacted CSR offsets
__global__ void __launch_bounds__(1024) k_compact_scan(u64 total, CompactView v)
{
    __shared__ u64 part[1024];
    const u32 tid = threadIdx.x;
    const u64 chunk = (total + 1023) / 1024;
    const u64 g0 = min(total, (u64)tid * chunk), g1 = min(total, g0 + chunk);
    u64 sum = 0;
    for (u64 g = g0; g < g1; ++g)
        sum += v.keep[g];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        u64 run = 0;
        for (u32 t = 0; t < 1024; ++t) {
            const u64 x = part[t];
            part[t] = run;
            run += x;
        }
        v.scan[total] = run;
    }
    __syncthreads();
    u64 run = part[tid];
    for (u64 g = g0; g < g1; ++g) {
        v.scan[g] = run;
        run += v.keep[g];
    }
}

// compacted CSR offsets (separate launch: reads scan[] written by the whole scan workgroup)
__global__ void __launch_bounds__(256) k_compact_offsets(const u64 *__restrict__ off, u32 batch,
                                                         CompactView v, u64 *__restrict__ off_out)
{
    const u32 b = blockIdx.x * 256u + threadIdx.x;
    if (b <= batch)
        off_out[b] = v.scan[off[b]];                 // off[batch] == total
}

__global__ void __launch_bounds__(256) k_compact_scatter(const u64 *__restrict__ terms, u64 total_words,
                                                         u32 dL, CompactView v, u64 *__restrict__ out)
{
    const u64 w = (u64)blockIdx.x * 256u + threadIdx.x;
    if (w >= total_words)
        return;
    const u64 g = w / dL;
    if (v.keep[g])
        out[v.scan[g] * dL + (w - g * dL)] = terms[w];
}

size_t compact_scratch_bytes(u64 total_terms)
{
    // keys 16 B + rep 8 B + parity 8 B (2 slots per term) + slot_of 4 + keep 4 + scan 8 (+1)
    return (size_t)total_terms * (16 + 8 + 8 + 4 + 4 + 8) + 8 + 6 * 256;
}

hipError_t compact(u64 n_bits, u64 batch, u64 total_terms, const u64 *terms, const u64 *off, u64 *out,
                   u64 *off_out, void *scratch, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0)
        return hipSuccess;
    // slot indices (2*off[b]+slot, up to 2*total_terms) are kept in 32 bits
    if (batch >= (1ull << 31) || total_terms >= (1ull << 31))
        return hipErrorInvalidValue;
    auto up = [](uintptr_t x) { return (x + 255) & ~(uintptr_t)255; };
    unsigned char *p = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(scratch)));
    CompactView v;
    v.keys = reinterpret_cast<u64 *>(p);
    p = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(p + total_terms * 16)));
    v.rep = reinterpret_cast<u32 *>(p);
    p = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(p + total_terms * 8)));
    v.parity = reinterpret_cast<u32 *>(p);
    p = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(p + total_terms * 8)));
    v.slot_of = reinterpret_cast<u32 *>(p);
    p = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(p + total_terms * 4)));
    v.keep = reinterpret_cast<u32 *>(p);
    p = reinterpret_cast<unsigned char *>(up(reinterpret_cast<uintptr_t>(p + total_terms * 4)));
    v.scan = reinterpret_cast<u64 *>(p);
    hipError_t e;
    if (total_terms) {
        if ((e = hipMemsetAsync(v.keys, 0, total_terms * 16, s)) != hipSuccess)
            return e;
        if ((e = hipMemsetAsync(v.rep, 0xFF, total_terms * 8, s)) != hipSuccess)
            return e;
        if ((e = hipMemsetAsync(v.parity, 0, total_terms * 8, s)) != hipSuccess)
            return e;
        const u32 blocks = ceil_div_u64(total_terms, 256);
        k_compact_insert<<<blocks, 256, 0, s>>>(terms, off, (u32)batch, total_terms, (u32)dL, v);
        k_compact_match<<<blocks, 256, 0, s>>>(terms, off, (u32)batch, total_terms, (u32)dL, v);
        k_compact_decide<<<blocks, 256, 0, s>>>(off, (u32)batch, total_terms, v);
    }
    k_compact_scan<<<1, 1024, 0, s>>>(total_terms, v);
    k_compact_offsets<<<ceil_div_u64(batch + 1, 256), 256, 0, s>>>(off, (u32)batch, v, off_out);
    if (total_terms) {
        const u64 words = total_terms * dL;
        const u64 blocks64 = (words + 255) / 256;
        if (blocks64 > kMaxBlocks256)
            return hipErrorInvalidValue;
        k_compact_scatter<<<(u32)blocks64, 256, 0, s>>>(terms, words, (u32)dL, v, out);
    }
    return hipGetLastError();
}

} // namespace csgn
