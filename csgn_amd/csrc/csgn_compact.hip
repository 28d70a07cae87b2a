// csgn_compact.hip -- EXTENSION: mod-2 compaction of term lists (not reference behaviour).
// Hand-written CDNA4 (gfx950) HIP; shared helpers in csgn_device.h, design notes in DESIGN.md 4.8.
//
// Decryption XORs over terms (/root/reference/src/SecretKey.cpp:139), so a term occurring an even number
// of times contributes nothing and one occurring an odd number of times contributes once: each
// ciphertext is rewritten as its distinct odd-multiplicity terms, in order of first occurrence.  The
// reference never does this (add is pure concatenation, src/Ciphertext.cpp:107-122), so it is opt-in
// and never runs on a parity path.
//
// Shape of the work: 8*dL*T_in bytes read, 8*dL*T_out written; everything else must stay small beside
// that.  The batch is cut into GROUPS of consecutive ciphertexts of at most capT terms (one ciphertext
// of up to capT terms, or a run of smaller ones), and one 512-thread workgroup (two share a CU) takes one
// group:
//   1. its units go from HBM into REGISTERS, wave w owning a contiguous span of 64-unit rows (coalesced
//      16-byte loads, R = 20 of them in flight per lane) -- they are read ONCE;
//   2. every unit is hashed where it sits (UMAC's NH step with a position tweak: two v_mad_u64_u32) and
//      the U unit hashes of a term are summed inside the wave through an LDS strip;
//   3. one lane per term chains {48-bit tag, term index} into a bucket of its ciphertext (ONE ds_wrxchg
//      on the bucket's head, as many buckets as terms) and walks its bucket: the smallest index with
//      the same tag is the class's representative, the class's size mod 2 decides;
//   4. every unit whose term joined a class is compared with the same unit of the representative (one
//      16-byte load from L2, U lanes per term, coalesced).  A difference means a tag collision between
//      unequal terms: the walk is redone with full word compares inside the bucket (exact, slow,
//      practically never taken: 2^-48 per pair);
//   5. survivors = representatives of odd classes; a ballot scan ranks them; the group's output
//      offset comes from a decoupled look-back over one 8-byte {flag, count} granule per group
//      (tickets are handed out in group order, so a group only ever waits for groups that are
//      already running);
//   6. the registers are stored to their final place (16-byte non-temporal stores, U lanes per term).
// Ciphertexts of more than capT terms ("large") cannot be deduplicated inside one workgroup: their
// terms are hashed first (k_cl_hash: the first read) and every {hash, index} pair is dealt by the top
// bits of the hash to one of T/1024 PARTITIONS of its ciphertext (sorted by partition in LDS a stripe of
// 8192 terms at a time, one atomic on a partition's cursor per stripe -- no read-modify-write of a table in
// HBM); a partition fits LDS and is
// deduplicated there like a group (k_cl_scatter, k_cl_dedup), leaving one keep byte per term, the terms that
// joined a class compared with their representatives.  The main kernel then moves the chunks of
// a large ciphertext exactly like a group, reading the keep bytes instead of making the decision:
// their terms are read twice.  A partition that overflows (one term repeated thousands of times) or a
// hash collision between unequal terms sends the call to the exact path: an open-addressing table in
// HBM with full compares inside the probe loop (k_cl_clear, k_cl_insert).
#include "csgn_device.h"

namespace csgn {

namespace {

constexpr u32 kCT = 512;                  // threads of a group's workgroup: two workgroups share a CU
// Two builds of the main kernel: 20 units per lane in registers (10 240 units = 1024 terms at N=1247; 122 VGPRs, two
// workgroups per CU) and 48 (24 576 units = 1792 terms at N=1247, 768 at N=4096 -- BASELINE config 5's end size; one
// workgroup per CU), the second taken when the caller's bound says the batch has ciphertexts between the two sizes.
constexpr int kCR = 20, kCRWide = 48;
constexpr u32 kCapUnits = kCT * kCR, kCapUnitsWide = kCT * kCRWide;
constexpr u32 kMaxGroupTerms = 1024, kMaxGroupTermsWide = 1792;     // LDS: 24 B per term + the hash strips, under 64 KiB

// control words (u64 each) at the head of the scratch block; zeroed, with the status granules, per call
// (kCtrlCollision: a partition overflowed or unequal terms shared a hash -- the exact path decides the large ciphertexts)
enum { kCtrlTicket = 0, kCtrlGroups = 1, kCtrlChunks = 2, kCtrlCollision = 3, kCtrlParts = 4, kCtrlStripes = 5, kCtrlGeomB = 6, kCtrlWords = 32 };

// Partitions of a large ciphertext of T terms: P = 2^lp >= T / 1024 of them, picked by the top lp bits of a term's hash,
// each with room for cap = 2T / P <= 2048 {hash, index} pairs (twice the mean; P <= 2 cannot overflow at all).
constexpr u32 kPartTerms = 1024, kPartCap = 2048;
constexpr u32 kStripeTerms = 8192, kStripeMaxLp = 11;       // k_cl_scatter
struct PartGeom {
    u32 lp, cap;
};
__host__ __device__ inline PartGeom part_geom(u64 T)
{
    PartGeom g;
    g.lp = 0;
    while (((u64)kPartTerms << g.lp) < T)
        ++g.lp;
    const u64 cap = (2 * T) >> g.lp;
    g.cap = (u32)(cap < kPartCap ? cap : kPartCap);
    return g;
}
// Cursor of partition q of a ciphertext, in the head of the ciphertext's share of `slot_of`: a 128-byte line each
// where the share has the room (atomics on one line queue behind one another)
__host__ __device__ inline u32 cursor_stride(u64 c_terms, u32 lp) { return ((u64)32 << lp) <= c_terms ? 32u : 1u; }

struct Geom {
    u32 U;          // units per term
    u32 capT;       // terms per group at most
    u32 wshift;     // log2 of the window: ciphertexts whose first terms share a window may share a group
    u32 bigT;       // a ciphertext of more terms than this is a group by itself: capT - window
    u32 wany;       // the window is NOT a power of two: capT - (the caller's bound on a ciphertext), divided by dW
    FastDiv dW;
};

// max_terms: the caller's bound on one ciphertext's terms (0 = unknown).  A run of ciphertexts that begin in one window of
// W terms and are at most bigT terms each spans fewer than W + bigT <= capT terms.  Without a bound W = bigT = capT / 2:
// runs of tiny ciphertexts make HALF-full groups (2^22 single terms: 8192 groups of 512, each paying a whole group's
// look-back and latencies).  With a small bound m the window is capT - m: 2^20 ciphertexts of four terms are 4112 groups.
Geom make_geom(u32 U, bool wide = false, u64 max_terms = 0)
{
    Geom g;
    g.U = U;
    g.capT = wide ? min(kMaxGroupTermsWide, kCapUnitsWide / U) : min(kMaxGroupTerms, kCapUnits / U);
    g.wshift = 0;
    while ((2u << g.wshift) <= g.capT / 2)
        ++g.wshift;
    g.bigT = g.capT - (1u << g.wshift);
    g.wany = 0;
    g.dW = csgn_fastdiv_make(1u);
    if (max_terms != 0 && max_terms < g.bigT) {
        g.bigT = (u32)max_terms;
        g.wany = 1;
        g.dW = csgn_fastdiv_make(g.capT - g.bigT);
    }
    return g;
}
// the window a ciphertext's first term lies in (term indices stay under 2^31: csgn::compact)
__device__ inline u32 window_of(u64 o, const Geom &g) { return g.wany ? csgn_fastdiv((u32)o, g.dW) : (u32)(o >> g.wshift); }

// groups a call can produce at most: one per window, two per big ciphertext, the chunks of large ones
u64 group_bound(u64 total_terms, const Geom &g) { return 4 + 10 * (total_terms / g.capT + 1); }

struct GroupDesc;
struct Layout {
    GroupDesc *groups;
    u64 *ctrl, *status, *chunks, *partial, *hash, *tab, *plist, *slist;
    u32 *gpos, *gposB, *par, *slot_of;
    u64 *partialB;
    unsigned char *keepb;
    size_t bytes, head_bytes;
};

Layout make_layout(void *scratch, u64 batch, u64 total_terms, const Geom &g)
{
    auto up = [](uintptr_t x) { return (x + 255) & ~(uintptr_t)255; };
    const u64 ng = group_bound(total_terms, g);
    uintptr_t p = up(reinterpret_cast<uintptr_t>(scratch));
    const uintptr_t p0 = p;
    auto take = [&](size_t bytes) {
        const uintptr_t at = p;
        p = up(p + bytes);
        return at;
    };
    Layout l;
    l.ctrl = reinterpret_cast<u64 *>(take((kCtrlWords + ng) * 8));
    l.status = l.ctrl + kCtrlWords;
    l.head_bytes = (kCtrlWords + ng) * 8;
    l.groups = reinterpret_cast<GroupDesc *>(take(ng * 32));
    l.chunks = reinterpret_cast<u64 *>(take(ng * 8));
    l.gpos = reinterpret_cast<u32 *>(take(batch * 4));
    l.gposB = reinterpret_cast<u32 *>(take(batch * 4));
    l.partial = reinterpret_cast<u64 *>(take((batch / 256 + 2) * 8));
    l.partialB = reinterpret_cast<u64 *>(take((batch / 256 + 2) * 8));
    l.hash = reinterpret_cast<u64 *>(take(total_terms * 8));
    l.tab = reinterpret_cast<u64 *>(take(total_terms * 16));
    l.par = reinterpret_cast<u32 *>(take(total_terms * 8));
    l.slot_of = reinterpret_cast<u32 *>(take(total_terms * 4));
    l.plist = reinterpret_cast<u64 *>(take((ng + total_terms / (kPartTerms / 2) + 1) * 8));
    l.slist = reinterpret_cast<u64 *>(take((ng + total_terms / kStripeTerms + 1) * 8));
    l.keepb = reinterpret_cast<unsigned char *>(take(total_terms));
    l.bytes = (p - p0) + 256;
    return l;
}

// ------------------------------------------------------------------------------- hashing
// Contribution of unit k of a term to the term's hash; the contributions are summed mod 2^64 (and the
// sum is finished with splitmix64).  UMAC's NH step: 32-bit words plus 32-bit keys, multiplied in pairs
// to 64 bits -- two v_mad_u64_u32 per 16-byte unit.  The keys move with the unit's position in the term
// (two full-rate 24-bit multiplies), so equal units in different places contribute differently.  A
// contribution cannot be the same for two units that differ in ONE word (a product of two non-zero 32-bit
// numbers is not 0 mod 2^64); anything else is a ~2^-64 accident -- and an accident only costs the
// exact redo, never a wrong merge.
constexpr u32 kNhA = 0x9E3779B9u, kNhB = 0x85EBCA6Bu, kNhC = 0xC2B2AE35u, kNhD = 0x27D4EB2Fu;
constexpr u32 kNhP = 0x9E3779u, kNhQ = 0xC2B2AFu;            // 24-bit odd position multipliers

__device__ inline u64 unit_hash(unit16 v, u32 k)
{
    const u32 p = __umul24(k + 1u, kNhP), q = __umul24(k + 1u, kNhQ);
    return (u64)(v.x + kNhA + p) * (u64)(v.y + kNhB + q) + (u64)(v.z + kNhC + q) * (u64)(v.w + kNhD + p);
}
__device__ inline u64 unit_hash(unit8 v, u32 k)
{
    const u32 p = __umul24(k + 1u, kNhP), q = __umul24(k + 1u, kNhQ);
    return (u64)((u32)v + kNhA + p) * (u64)((u32)(v >> 32) + kNhB + q);
}
// term and position of a lane's units: unit j = i * kCT + lane for i = 0, 1, ... walks on by a fixed stride
struct UnitPos {
    u32 t, k;
};
struct UnitWalk {
    u32 U, step_t, step_k;       // kCT = step_t * U + step_k
    FastDiv dU;
};
__device__ inline UnitPos walk_first(const UnitWalk &w, u32 lane_unit)
{
    UnitPos p;
    p.t = csgn_fastdiv(lane_unit, w.dU);
    p.k = lane_unit - p.t * w.U;
    return p;
}
__device__ inline void walk_next(const UnitWalk &w, UnitPos &p)
{
    p.k += w.step_k;
    p.t += w.step_t;
    if (p.k >= w.U) {
        p.k -= w.U;
        p.t += 1u;
    }
}
__device__ inline bool unit_same(unit16 a, unit16 b)
{
    const unit16 d = a ^ b;
    return (d.x | d.y | d.z | d.w) == 0u;
}
__device__ inline bool unit_same(unit8 a, unit8 b) { return a == b; }
template <typename Unit>
__device__ inline Unit unit_zero();
template <>
__device__ inline unit16 unit_zero<unit16>()
{
    unit16 z = {0u, 0u, 0u, 0u};
    return z;
}
template <>
__device__ inline unit8 unit_zero<unit8>()
{
    return 0ull;
}

__device__ inline bool words_equal(const u64 *x, const u64 *y, u32 dL)
{
    bool same = true;
    for (u32 k = 0; k < dL; ++k)
        same = same && (x[k] == y[k]);
    return same;
}

__device__ inline unsigned long long *ull(u64 *p) { return reinterpret_cast<unsigned long long *>(p); }

// ----------------------------------------------------------------- the groups of a batch
// How many groups START at ciphertext c (0: it continues the run before it).  A run = consecutive
// ciphertexts of at most bigT terms whose first terms lie in one window of 2^wshift terms: together
// they span fewer than capT terms.  A ciphertext of more than bigT terms is alone; one of more than
// capT terms ("large") is cut into chunks of capT terms, one group each.
__device__ inline u32 group_count(const u64 *__restrict__ off, u32 c, Geom g, bool &large)
{
    const u64 o0 = off[c], t = off[c + 1] - o0;
    large = t > g.capT;
    bool start = c == 0;
    if (!start) {
        const u64 om = off[c - 1];
        start = t > g.bigT || o0 - om > g.bigT || window_of(o0, g) != window_of(om, g);
    }
    if (!start)
        return 0u;
    return large ? (u32)((t + g.capT - 1) / g.capT) : 1u;
}

__device__ inline u32 wave_incl_scan(u32 v)
{
    const u32 lane = threadIdx.x & (kWave - 1);
    for (u32 d = 1; d < kWave; d <<= 1) {
        const u32 n = __shfl_up(v, d, kWave);
        if (lane >= d)
            v += n;
    }
    return v;
}

// (also clears the call's control words and status granules: `head`, nothing of which is touched before
// the next kernel of the call)
// DUAL: the caller gave a bound on one ciphertext's terms and geometry A was cut to it (make_geom).  A caller whose
// ciphertexts break the bound must still get a legal result from a scratch block sized without it: every ciphertext over
// the bound is a group of its own, and thousands of them would overrun the group arrays.  So the groups are counted under
// the unbounded geometry B as well, every block leaves word of any ciphertext between the two limits (bit 63 of its B
// count), and k_cg_scan / k_cg_fill take geometry B for the whole call if any block did.
constexpr u64 kViolation = 1ull << 63;
__global__ void __launch_bounds__(256) k_cg_count(u32 batch, const u64 *__restrict__ off, Geom gA, Geom gB, u32 dual,
                                                  u32 *__restrict__ gposA, u32 *__restrict__ gposB,
                                                  u64 *__restrict__ partialA, u64 *__restrict__ partialB,
                                                  u64 *__restrict__ head, u64 head_words)
{
    __shared__ u32 wsum[2][4];
    __shared__ u32 s_viol;
    for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < head_words; i += (u64)gridDim.x * 256u)
        head[i] = 0ull;
    if (threadIdx.x == 0)
        s_viol = 0u;
    const u32 c = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    bool large;
    const u32 nA = c < batch ? group_count(off, c, gA, large) : 0u;
    const u32 nB = (dual && c < batch) ? group_count(off, c, gB, large) : 0u;
    const u32 inclA = wave_incl_scan(nA), inclB = dual ? wave_incl_scan(nB) : 0u;
    if (lane == kWave - 1) {
        wsum[0][wave] = inclA;
        wsum[1][wave] = inclB;
    }
    __syncthreads();
    if (dual && c < batch) {
        const u64 t = off[c + 1] - off[c];
        if (t > gA.bigT && t <= gB.bigT)
            s_viol = 1u;
    }
    u32 baseA = 0, totA = 0, baseB = 0, totB = 0;
    for (u32 w = 0; w < 4; ++w) {
        baseA += w < wave ? wsum[0][w] : 0u;
        totA += wsum[0][w];
        baseB += w < wave ? wsum[1][w] : 0u;
        totB += wsum[1][w];
    }
    if (c < batch) {
        gposA[c] = baseA + inclA - nA;                // block-local; k_cg_fill adds the block's base
        if (dual)
            gposB[c] = baseB + inclB - nB;
    }
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < (batch + 255u) / 256u) { // (blocks past the batch only clear)
        partialA[blockIdx.x] = totA;
        if (dual)
            partialB[blockIdx.x] = (u64)totB | (s_viol ? kViolation : 0ull);
    }
}

// (beyond 262 144 ciphertexts) exclusive scan of the blocks' counts, left in partialA whichever geometry it is for.
// One workgroup: a thread's consecutive counts are loaded together (sixteen for four million ciphertexts), the 1024
// thread sums are scanned by waves.  (The first form let thread 0 walk the 1024 sums in LDS: 42 of the kernel's 48 us.)
__global__ void __launch_bounds__(1024) k_cg_scan(u64 nblocks, u64 *__restrict__ partialA, const u64 *__restrict__ partialB,
                                                  u32 dual, u64 *__restrict__ ctrl)
{
    constexpr int kPer = 16;
    __shared__ u64 s_wave[16];
    __shared__ u32 s_flag;
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const u64 chunk = (nblocks + 1023) / 1024;
    const u64 c0 = min(nblocks, (u64)tid * chunk), c1 = min(nblocks, c0 + chunk);
    const bool in_regs = chunk <= (u64)kPer;
    if (tid == 0)
        s_flag = 0u;
    __syncthreads();
    u64 va[kPer], vb[kPer];
    bool viol = false;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            va[j] = c0 + j < c1 ? partialA[c0 + j] : 0ull;
            vb[j] = (dual && c0 + j < c1) ? partialB[c0 + j] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < kPer; ++j)
            viol = viol || (vb[j] & kViolation) != 0ull;
    } else if (dual) {
        for (u64 c = c0; c < c1; ++c)
            viol = viol || (partialB[c] & kViolation) != 0ull;
    }
    if (viol)
        s_flag = 1u;
    __syncthreads();
    const bool useB = s_flag != 0u;
    u64 sum = 0;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            va[j] = useB ? (vb[j] & ~kViolation) : va[j];
            sum += va[j];
        }
    } else {
        for (u64 c = c0; c < c1; ++c)
            sum += useB ? (partialB[c] & ~kViolation) : partialA[c];
    }
    u64 incl = sum;                                               // inclusive scan inside the wave ...
    for (u32 d = 1; d < kWave; d <<= 1) {
        const u64 nb = (u64)__shfl_up((unsigned long long)incl, d, kWave);
        if (lane >= d)
            incl += nb;
    }
    if (lane == kWave - 1)
        s_wave[wave] = incl;
    __syncthreads();
    u64 run = incl - sum, total = 0;                              // ... plus the waves before it
    for (u32 w = 0; w < 16; ++w) {
        run += w < wave ? s_wave[w] : 0ull;
        total += s_wave[w];
    }
    if (tid == 0) {
        ctrl[kCtrlGroups] = total;
        ctrl[kCtrlGeomB] = useB ? 1ull : 0ull;
    }
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < kPer; ++j)
            if (c0 + j < c1) {
                partialA[c0 + j] = run;
                run += va[j];
            }
    } else {
        for (u64 c = c0; c < c1; ++c) {
            const u64 v = useB ? (partialB[c] & ~kViolation) : partialA[c];
            partialA[c] = run;
            run += v;
        }
    }
}

// What a workgroup needs to know about a group, complete, so that one 32-byte (scalar) load after the
// ticket is all that stands between two groups.
struct GroupDesc {
    u64 tb, te;              // the group's terms
    u32 c0, c1;              // its ciphertexts [c0, c1)
    u32 chunk, large;        // a chunk of a large ciphertext: which one
};

// The thread of a group's FIRST ciphertext writes {tb, c0, chunk, large}, the thread of its LAST one
// {te, c1} (the same thread for a group of one).  The chunks of a LARGE ciphertext are written by the whole
// block together (a ciphertext of 2^20 terms has a thousand of them), listed a second time (any order) for
// k_cl_hash, and its partitions are listed for k_cl_dedup with their cursors zeroed.
__global__ void __launch_bounds__(256) k_cg_fill(u32 batch, const u64 *__restrict__ off, Geom gA, Geom gB, u32 dual,
                                                 const u32 *__restrict__ gposA, const u32 *__restrict__ gposB,
                                                 const u64 *__restrict__ partial, const u64 *__restrict__ partialB,
                                                 GroupDesc *__restrict__ groups, u64 *__restrict__ chunks,
                                                 u64 *__restrict__ plist, u64 *__restrict__ slist, u32 *__restrict__ cursor,
                                                 u64 *__restrict__ ctrl, u32 scanned)
{
    // the block's base: `partial` already scanned by k_cg_scan (scanned != 0), or -- up to 1024 blocks --
    // summed here, which saves the scan kernel's launch; block 0 then also leaves the total for the main kernel
    __shared__ u64 s_base, s_part[4];
    __shared__ u32 s_nlarge, s_lc[256];
    __shared__ u64 s_lbefore[256], s_at[256][3];
    u64 block_base = 0;
    bool useB = false;                                            // (see k_cg_count: the caller's bound did not hold)
    if (threadIdx.x == 0)
        s_nlarge = 0u;
    if (scanned) {
        block_base = partial[blockIdx.x];
        useB = dual && ctrl[kCtrlGeomB] != 0ull;
    } else {
        u64 mineA = 0, allA = 0, mineB = 0, allB = 0;
        for (u32 b = threadIdx.x; b < gridDim.x; b += 256u) {
            const u64 vA = partial[b], vB = dual ? partialB[b] : 0ull;
            allA += vA;
            mineA += b < blockIdx.x ? vA : 0ull;
            allB += vB & ~kViolation;
            mineB += b < blockIdx.x ? (vB & ~kViolation) : 0ull;
            useB = useB || (vB & kViolation) != 0ull;
        }
        useB = __syncthreads_or(useB ? 1 : 0) != 0;
        const bool want_total = blockIdx.x == 0;
        u64 red = useB ? (want_total ? allB : mineB) : (want_total ? allA : mineA);
        for (int o = 32; o > 0; o >>= 1)
            red += (u64)__shfl_down(red, o, 64);
        if ((threadIdx.x & (kWave - 1)) == 0)
            s_part[threadIdx.x >> 6] = red;
        __syncthreads();
        if (threadIdx.x == 0) {
            const u64 sum = s_part[0] + s_part[1] + s_part[2] + s_part[3];
            if (want_total)
                ctrl[kCtrlGroups] = sum;
            s_base = want_total ? 0ull : sum;
        }
        __syncthreads();
        block_base = s_base;
    }
    const Geom g = useB ? gB : gA;
    const u32 *__restrict__ gpos = useB ? gposB : gposA;
    __syncthreads();                                              // (s_nlarge is zero for everybody)
    const u32 c = blockIdx.x * 256u + threadIdx.x;
    bool large = false, next_large;
    const u32 n = c < batch ? group_count(off, c, g, large) : 0u;
    const u64 before = c < batch ? block_base + gpos[c] : 0ull;   // groups started before c
    if (large) {
        const u64 o0 = off[c], o1 = off[c + 1];
        if (n <= 16u) {
            // a few chunks (hence one or two partitions, one stripe): the ciphertext's own thread writes them -- a batch of
            // thousands of such ciphertexts is 256 threads a block at work, not a wave walking a list
            const PartGeom pg = part_geom(o1 - o0);
            const u32 nst = (u32)((o1 - o0 + kStripeTerms - 1) / kStripeTerms);
            const u64 at = atomicAdd(ull(ctrl + kCtrlChunks), (unsigned long long)n);
            const u64 pat = atomicAdd(ull(ctrl + kCtrlParts), (unsigned long long)(1u << pg.lp));
            const u64 sat = atomicAdd(ull(ctrl + kCtrlStripes), (unsigned long long)nst);
            for (u32 i = 0; i < n; ++i) {
                GroupDesc d;
                d.tb = o0 + (u64)i * g.capT;
                d.te = min(d.tb + g.capT, o1);
                d.c0 = c;
                d.c1 = c + 1u;
                d.chunk = i;
                d.large = 1u;
                groups[before + i] = d;
                chunks[at + i] = (u64)c | ((u64)i << 32);
            }
            for (u32 j = 0; j < nst; ++j)
                slist[sat + j] = (u64)c | ((u64)j << 32);
            for (u32 q = 0; q < (1u << pg.lp); ++q) {
                plist[pat + q] = (u64)c | ((u64)q << 32);
                cursor[o0 + (u64)q * cursor_stride(o1 - o0, pg.lp)] = 0u;
            }
        } else {
            const u32 slot = atomicAdd(&s_nlarge, 1u);
            s_lc[slot] = c;
            s_lbefore[slot] = before;
        }
    }
    __syncthreads();
    const u32 nlarge = s_nlarge;
    if (threadIdx.x < nlarge) {                                   // every large ciphertext's three reservations at once
        const u32 lc = s_lc[threadIdx.x];
        const u64 t = off[lc + 1] - off[lc];
        s_at[threadIdx.x][0] = atomicAdd(ull(ctrl + kCtrlChunks), (unsigned long long)((t + g.capT - 1) / g.capT));
        s_at[threadIdx.x][1] = atomicAdd(ull(ctrl + kCtrlParts), (unsigned long long)(1u << part_geom(t).lp));
        s_at[threadIdx.x][2] = atomicAdd(ull(ctrl + kCtrlStripes), (unsigned long long)((t + kStripeTerms - 1) / kStripeTerms));
    }
    __syncthreads();
    for (u32 li = threadIdx.x >> 6; li < nlarge; li += 4u) {       // a wave per large ciphertext: their latencies overlap
        const u32 ln = threadIdx.x & (kWave - 1);
        const u32 lc = s_lc[li];
        const u64 o0 = off[lc], o1 = off[lc + 1], first = s_lbefore[li];
        const u32 nch = (u32)((o1 - o0 + g.capT - 1) / g.capT);
        const PartGeom pg = part_geom(o1 - o0);
        const u32 nst = (u32)((o1 - o0 + kStripeTerms - 1) / kStripeTerms);
        const u64 at = s_at[li][0], pat = s_at[li][1], sat = s_at[li][2];
        for (u32 i = ln; i < nch; i += kWave) {
            GroupDesc d;
            d.tb = o0 + (u64)i * g.capT;
            d.te = min(d.tb + g.capT, o1);
            d.c0 = lc;
            d.c1 = lc + 1u;
            d.chunk = i;
            d.large = 1u;
            groups[first + i] = d;
            chunks[at + i] = (u64)lc | ((u64)i << 32);
        }
        for (u32 j = ln; j < nst; j += kWave)
            slist[sat + j] = (u64)lc | ((u64)j << 32);
        for (u32 q = ln; q < (1u << pg.lp); q += kWave) {
            plist[pat + q] = (u64)lc | ((u64)q << 32);
            cursor[o0 + (u64)q * cursor_stride(o1 - o0, pg.lp)] = 0u;
        }
    }
    if (c >= batch || large)
        return;
    const u64 o0 = off[c], o1 = off[c + 1];
    const u64 gid = n ? before : before - 1u;                     // the run c belongs to
    if (n) {
        groups[gid].tb = o0;
        groups[gid].c0 = c;
        groups[gid].chunk = 0u;
        groups[gid].large = 0u;
    }
    if (c + 1u == batch || group_count(off, c + 1u, g, next_large) > 0u) {
        groups[gid].te = o1;
        groups[gid].c1 = c + 1u;
    }
}

// ---------------------------------------------------------------------- the main kernel
struct CompactArgs {
    const void *terms;
    void *out;
    const u64 *off;
    u64 *off_out;
    u64 *ctrl;
    u64 *status;
    const GroupDesc *groups;
    const unsigned char *keepb;   // large ciphertexts: one keep byte per term (k_cl_dedup) ...
    const u64 *tab;          // ... or, on the exact path (kCtrlCollision), the open-addressing table, two slots per term
    const u32 *par;
    const u32 *slot_of;
    u64 tag_mask;            // all ones; a test narrows it to force tag collisions
    Geom g;
    FastDiv dU;
    u32 batch;
    u32 dL;
    u32 large_ready;         // the k_cl_* kernels ran: large ciphertexts have a keep decision
    u32 nt;                  // non-temporal stores of the survivors
    u32 stagger;             // 100 MHz ticks by which every second workgroup starts late (0 = together)
    u64 *stamps;             // dev only (CSGN_COMPACT_STAMPS)
};

// dev-only phase stamps (tools/prof_compact_phases.py builds with -DCSGN_COMPACT_STAMPS): 100 MHz ticks of
// thread 0 per group and phase, written just past the scratch block (the tool allocates the extra room)
#ifdef CSGN_COMPACT_STAMPS
#define CSGN_STAMP(i) do { if (tid == 0) a.stamps[(u64)gid * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CSGN_STAMP(i) do { } while (0)
#endif


constexpr int sub_rows(int R) { return R % 5 == 0 ? 5 : 4; }      // register rows hashed per staging round
constexpr int term_passes(int R) { return (int)(((R == kCR ? kMaxGroupTerms : kMaxGroupTermsWide) + kCT - 1) / kCT); }

// dynamic LDS of the main kernel for groups of capT terms
inline size_t main_lds_bytes(u32 capT, int R)
{
    return (size_t)capT * (8 + 4 + 4 + 4 + 4) + 64 + (size_t)(kCT / kWave) * sub_rows(R) * kWave * 8;
}

// Ranks of a group's survivors (per pass the waves' ballots, then one look at all the wave counts: s_rk[t] = rank << 1
// | keep) and the group's place in the output: wave 0's look-back, left in *s_prefix for the caller's next barrier to
// publish.  Returns the number of survivors.
// (Round 5: the look-back by the whole workgroup -- thread t reads the status of group gid - 1 - t, the kCT nearest
// predecessors in one round trip instead of up to eight windows of 64 -- changed nothing, 0.322 against 0.30-0.31
// ms: the 8.7 us a group spends here are not the walk, they are the wait for the SLOWEST of the up to 511 groups
// in flight before it to publish its count.)
template <int kPasses>
__device__ inline u32 rank_and_place(const u32 (&keep)[kPasses], u32 nt, u32 tid, u32 lane, u32 wave, u32 *s_wsum, u32 *s_rk,
                                     u64 *status, u32 gid, u64 *s_prefix, bool lds_only)
{
    u32 before[kPasses];
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
        const u64 m = __ballot(keep[p] != 0u);
        before[p] = (u32)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0)
            s_wsum[(u32)p * (kCT / kWave) + wave] = (u32)__popcll(m);
    }
    if (lds_only)
        lds_barrier();                                            // (global requests stay in flight)
    else
        __syncthreads();
    u32 count = 0u, mybase[kPasses];
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
        mybase[p] = count;
        for (u32 w = 0; w < kCT / kWave; ++w) {
            const u32 v = s_wsum[(u32)p * (kCT / kWave) + w];
            mybase[p] += w < wave ? v : 0u;
            count += v;
        }
    }
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
        const u32 t = (u32)p * kCT + tid;
        if (t < nt)
            s_rk[t] = ((mybase[p] + before[p]) << 1) | keep[p];
    }
    if (tid == 0)
        s_rk[nt] = count << 1;
    if (lds_only)
        lds_barrier();                                            // (a chunk's loads read the ranks: before wave 0 goes looking back)
    if (wave == 0) {
        const u64 excl = lookback(status, gid, count);
        if (lane == 0)
            *s_prefix = excl;
    }
    return count;
}

// Unit j of a group sits in wave w = j / (64 rows), register row i = (j / 64) % rows, lane j % 64, with
// rows = the group's 64-unit rows dealt evenly over the eight waves (20 for a full group): every wave owns
// a contiguous span of 64 rows units (coalesced 1 KiB wave loads all the same), so the hashes of a term's
// units meet inside ONE wave.
template <typename Unit, int R>
__global__ void __launch_bounds__(kCT, (R <= kCR ? 2 : 1) * kCT / 256) k_compact_main(CompactArgs a)
{
    constexpr int kSub = sub_rows(R), kPasses = term_passes(R);
    static_assert(R % kSub == 0, "hash staging rounds must tile the register rows");
    extern __shared__ u64 s_dyn[];
    const u32 capT = a.g.capT, U = a.g.U;
    u64 *s_node = s_dyn;                                          // [capT] hash sum, then {tag, next in bucket}
    u64 *s_part = s_node + capT;                                  // [waves][kSub * 64] unit hashes in flight
    u32 *s_head = reinterpret_cast<u32 *>(s_part + (kCT / kWave) * kSub * kWave);   // [capT] bucket heads
    u32 *s_rep = s_head + capT;                                   // [capT] smallest equal term
    u32 *s_rk = s_rep + capT;                                     // [capT + 1] rank << 1 | keep
    u32 *s_coff = s_rk + capT + 2;                                // [capT + 2] ciphertext starts inside the group
    __shared__ u64 s_desc[4];                                     // {tb, te, c0 | c1 << 32, chunk | large << 32 | ticket << 33}
    __shared__ u32 s_join, s_bad, s_wsum[kPasses * (kCT / kWave)];
    __shared__ u64 s_prefix;
    const Unit *__restrict__ terms = static_cast<const Unit *>(a.terms);
    Unit *__restrict__ out = static_cast<Unit *>(a.out);
    u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const u32 ngroups = (u32)a.ctrl[kCtrlGroups];
    const bool exact_large = a.ctrl[kCtrlCollision] != 0ull;      // who decided the large ciphertexts
    UnitWalk walk;                                                // from one register row to the next: 64 units on
    walk.U = U;
    walk.step_t = kWave / U;
    walk.step_k = kWave - walk.step_t * U;
    walk.dU = a.dU;

    const u64 *gwords = reinterpret_cast<const u64 *>(a.groups);  // a descriptor = four 8-byte words
    // Workgroups that start together and take equally long stay in step for the whole launch: all of them load, then
    // all of them hash and wait on their look-backs (HBM idle), then all of them store.  Every second workgroup -- the
    // second of the two that share a CU -- starts half a group late, so that one of a CU's two is in a memory phase
    // while the other computes.  (Tickets are taken AFTER the wait: group order is still start order.)
    if (a.stagger && (blockIdx.x & 1u)) {
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < a.stagger)
            __builtin_amdgcn_s_sleep(32);
    }
    if (wave == 0) {
        u32 gid = 0u;
        if (lane == 0)
            gid = atomicAdd(reinterpret_cast<u32 *>(a.ctrl + kCtrlTicket), 1u);
        gid = __builtin_amdgcn_readfirstlane(gid);
        if (lane < 4u) {
            u64 w = gid < ngroups ? gwords[(u64)gid * 4u + lane] : 0ull;
            if (lane == 3u)
                w = (w & 0x1FFFFFFFFull) | ((u64)gid << 33);      // the ticket rides in the descriptor
            s_desc[lane] = w;
        }
        if (lane == 0) {
            s_join = 0u;
            s_bad = 0u;
        }
    }
    __syncthreads();
    for (;;) {
        const u64 d0 = s_desc[0], d1 = s_desc[1], d2 = s_desc[2], d3 = s_desc[3];
        const u32 gid = (u32)(d3 >> 33);
        if (gid >= ngroups)
            break;
        CSGN_STAMP(0);
        const bool large = ((d3 >> 32) & 1ull) != 0ull;
        const u32 c0 = (u32)d2, c1 = (u32)(d2 >> 32), chunk = (u32)d3;
        const u64 tb = d0;
        const u32 nt = (u32)(d1 - tb), nunits = nt * U, ncts = c1 - c0;
        const u64 ub = tb * U;
        const bool multi = !large && ncts > 1u, staged = multi && ncts <= capT + 1u;
        // The thread index is made opaque once per group: otherwise everything derived from it (LDS
        // addresses, unit -> term maps, lane masks, twenty rows' worth of each) is hoisted out of the ticket
        // loop as loop-invariant and spilled -- and a spill reload queues behind the stores in flight.
        asm volatile("" : "+v"(tid));
        lane = tid & (kWave - 1);
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        // rows of 64 units per wave: the group's rows dealt evenly over the waves (a half-full group keeps every
        // wave busy for half the rows instead of half the waves for all of them)
        const u32 rows = min((u32)R, ((nunits + kWave - 1u) / kWave + kCT / kWave - 1u) / (kCT / kWave));
        const u32 span = rows * kWave;                            // units per wave
        const u32 j0 = wave * span + lane;
        CSGN_STAMP(1);

        // A chunk of a large ciphertext knows its survivors before it has read a single unit: its keep bytes are
        // loaded FIRST, and count, look-back and offsets run under the unit loads (LDS-only barriers: nothing waits
        // for the units until the stores) -- and the groups behind it learn its count at once.
        // (Not in the wide build: one workgroup per CU has nobody to overlap with, and the extra live range spills.)
        constexpr bool kEarly = R <= kCR;
        u32 keep[kPasses], count = 0u;
        auto large_keep = [&]() {
#pragma unroll
            for (int p = 0; p < kPasses; ++p) {
                const u32 t = (u32)p * kCT + tid;
                keep[p] = 0u;
                if (t < nt) {
                    keep[p] = 1u;                                 // no decision made: the chunk is copied as it is
                    if (a.large_ready) {
                        const u64 g = tb + t;
                        if (exact_large) {
                            const u32 s = a.slot_of[g];
                            keep[p] = ((u32)a.tab[s] - 1u == (u32)g && (a.par[s] & 1u)) ? 1u : 0u;
                        } else {
                            keep[p] = a.keepb[g];
                        }
                    }
                }
            }
        };
        if (kEarly && large) {
            large_keep();
            count = rank_and_place<kPasses>(keep, nt, tid, lane, wave, s_wsum, s_rk, a.status, gid, &s_prefix, true);
        }
        // 1. the group's units -> registers.  A chunk of a large ciphertext already knows which of its terms survive (the
        // ranks are in LDS): the units of the others are not read a second time -- at 50 % duplicates two thirds of this
        // kernel's bytes were reads of terms it then dropped.
        Unit reg[R];
        if (kEarly && large) {
            u32 jl = j0;
            asm volatile("" : "+v"(jl));
            UnitPos pos = walk_first(walk, jl);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const u32 j = jl + (u32)i * kWave;
                const bool want = (u32)i < rows && j < nunits && (s_rk[pos.t] & 1u) != 0u;
                reg[i] = want ? terms[ub + j] : unit_zero<Unit>();
                walk_next(walk, pos);
            }
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const u32 j = j0 + (u32)i * kWave;
                reg[i] = ((u32)i < rows && j < nunits) ? terms[ub + j] : unit_zero<Unit>();   // (non-temporal loads: no gain, measured)
            }
        }
        if (!large) {
            for (u32 x = tid; x < nt; x += kCT) {
                s_node[x] = 0ull;
                s_head[x] = 0u;
            }
            if (staged)
                for (u32 x = tid; x <= ncts; x += kCT)
                    s_coff[x] = (u32)(a.off[c0 + x] - tb);
            __syncthreads();
        }
#ifdef CSGN_COMPACT_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // so that stamp 2 - stamp 1 is the load time
        __syncthreads();
#endif
        CSGN_STAMP(2);

        if (!large) {
            // 2. term hashes: kSub rows of unit hashes at a time through the wave's own LDS strip, then
            //    one lane per term of the strip adds its units up (DS operations of one wave run in order:
            //    no barrier); only a term cut by the strip's edge needs an atomic
            {
                u64 *part = s_part + wave * (kSub * kWave);
                u32 jh = j0;                                      // (a copy of its own per phase: the twenty
                asm volatile("" : "+v"(jh));                      //  {term, position} pairs must not stay live)
                UnitPos pos = walk_first(walk, jh);
#pragma unroll
                for (int r = 0; r < R / kSub; ++r) {
                    if ((u32)(r * kSub) >= rows)                  // wave-uniform: the wave's rows are done
                        break;
                    const u32 w0 = wave * span + (u32)r * (kSub * kWave);             // first unit of the strip
#pragma unroll
                    for (int q = 0; q < kSub; ++q) {
                        part[q * (int)kWave + (int)lane] = unit_hash(reg[r * kSub + q], pos.k);
                        walk_next(walk, pos);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (w0 < nunits) {                            // wave-uniform
                        const u32 w1 = min(min(w0 + kSub * kWave, (wave + 1u) * span), nunits);
                        const u32 tf = csgn_fastdiv(w0, a.dU), tl = csgn_fastdiv(w1 - 1u, a.dU);
                        for (u32 t = tf + lane; t <= tl; t += kWave) {
                            const u32 u0 = max(t * U, w0), u1 = min(t * U + U, w1);
                            u64 s0 = 0ull, s1 = 0ull;
                            u32 u = u0;
                            for (; u + 1u < u1; u += 2u) {
                                s0 += part[u - w0];
                                s1 += part[u + 1u - w0];
                            }
                            if (u < u1)
                                s0 += part[u - w0];
                            if (u1 - u0 == U)
                                s_node[t] = s0 + s1;
                            else
                                atomicAdd(ull(s_node + t), s0 + s1);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __syncthreads();
            CSGN_STAMP(3);
            // 3. one lane per term: chain the term into a bucket of its ciphertext (as many buckets as the
            //    ciphertext has terms): ONE exchange on the bucket's head, no probing
            u32 bucket[kPasses];
            u64 tag[kPasses];
#pragma unroll
            for (int p = 0; p < kPasses; ++p) {
                const u32 t = (u32)p * kCT + tid;
                bucket[p] = 0u;
                tag[p] = 0ull;
                if (t < nt) {
                    u32 a0 = 0u, b0 = nt;
                    if (staged) {                                 // largest i with s_coff[i] <= t
                        u32 lo = 0u, hi = ncts;
                        while (hi - lo > 1u) {
                            const u32 mid = (lo + hi) >> 1;
                            if (s_coff[mid] <= t)
                                lo = mid;
                            else
                                hi = mid;
                        }
                        a0 = s_coff[lo];
                        b0 = s_coff[lo + 1u];
                    } else if (multi) {
                        const u32 c = csr_find(a.off, c0, c1, tb + t);
                        a0 = (u32)(a.off[c] - tb);
                        b0 = (u32)(a.off[c + 1u] - tb);
                    }
                    const u64 h = csgn_splitmix64(s_node[t]);
                    tag[p] = (h >> 16) & a.tag_mask;
                    bucket[p] = a0 + (u32)(((u64)(u32)h * (b0 - a0)) >> 32);
                    const u32 next = atomicExch(s_head + bucket[p], t + 1u);
                    s_node[t] = (tag[p] << 16) | (u64)next;
                }
            }
            __syncthreads();
            // ... and walk the bucket: the smallest term with the same tag is the class's representative
            u32 cnt[kPasses];
#pragma unroll
            for (int p = 0; p < kPasses; ++p) {
                const u32 t = (u32)p * kCT + tid;
                u32 rep = t;
                cnt[p] = 0u;
                if (t < nt) {
                    for (u32 e = s_head[bucket[p]]; e != 0u;) {
                        const u64 node = s_node[e - 1u];
                        if ((node >> 16) == tag[p]) {
                            ++cnt[p];
                            rep = min(rep, e - 1u);
                        }
                        e = (u32)(node & 0xFFFFu);
                    }
                    s_rep[t] = rep;
                    if (rep != t)
                        s_join = 1u;
                }
                keep[p] = (t < nt && rep == t && (cnt[p] & 1u)) ? 1u : 0u;
            }
            __syncthreads();
            CSGN_STAMP(4);
            // 4. every unit of a term that joined a class is compared with its representative's
            if (s_join) {
                u32 jv = j0;
                asm volatile("" : "+v"(jv));
                UnitPos pos = walk_first(walk, jv);
                bool bad = false;
                constexpr int VB = 2;                              // rows whose loads travel together (4 spill a data register)
                static_assert(R % VB == 0, "verify batches must tile the register rows");
#pragma unroll
                for (int r = 0; r < R / VB; ++r) {
                    // every lane loads SOMETHING valid (its representative's unit, or its own unit again) so that the
                    // VB loads of a batch are issued back to back instead of one per branch; a batch in which no lane
                    // of the wave joined anybody is skipped (wave-uniform)
                    u32 at[VB];                                    // unit index inside the group
                    bool need[VB];
                    bool any = false;
#pragma unroll
                    for (int q = 0; q < VB; ++q) {
                        const u32 j = jv + (u32)(r * VB + q) * kWave;
                        const bool in = (u32)(r * VB + q) < rows && j < nunits;
                        const u32 rep = in ? s_rep[pos.t] : 0u;
                        need[q] = in && rep != pos.t;
                        at[q] = need[q] ? rep * U + pos.k : (in ? j : 0u);
                        any = any || need[q];
                        walk_next(walk, pos);
                    }
                    if (__ballot(any) != 0ull) {
                        Unit other[VB];
#pragma unroll
                        for (int q = 0; q < VB; ++q)
                            other[q] = terms[ub + at[q]];
#pragma unroll
                        for (int q = 0; q < VB; ++q)
                            bad |= need[q] && !unit_same(other[q], reg[r * VB + q]);
                    }
                }
                if (bad)
                    s_bad = 1u;
                __syncthreads();
                if (s_bad) {                                      // a tag collision: walk again with full compares
                    const u64 *words = reinterpret_cast<const u64 *>(a.terms) + tb * a.dL;
#pragma unroll
                    for (int p = 0; p < kPasses; ++p) {
                        const u32 t = (u32)p * kCT + tid;
                        u32 rep = t;
                        cnt[p] = 0u;
                        if (t < nt) {
                            for (u32 e = s_head[bucket[p]]; e != 0u;) {
                                const u64 node = s_node[e - 1u];
                                if ((node >> 16) == tag[p] &&
                                    (e - 1u == t || words_equal(words + (u64)t * a.dL, words + (u64)(e - 1u) * a.dL, a.dL))) {
                                    ++cnt[p];
                                    rep = min(rep, e - 1u);
                                }
                                e = (u32)(node & 0xFFFFu);
                            }
                        }
                        keep[p] = (t < nt && rep == t && (cnt[p] & 1u)) ? 1u : 0u;
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    s_join = 0u;
                    s_bad = 0u;
                }
            }
            CSGN_STAMP(5);
        }
        CSGN_STAMP(6);
        if (!kEarly && large)
            large_keep();
        if (!(kEarly && large))
            count = rank_and_place<kPasses>(keep, nt, tid, lane, wave, s_wsum, s_rk, a.status, gid, &s_prefix, false);
        CSGN_STAMP(7);
        if (kEarly && large)
            lds_barrier();
        else
            __syncthreads();
        const u64 prefix = s_prefix;
        CSGN_STAMP(8);
        // The next ticket, taken by the last wave while the others write; its descriptor is loaded before
        // that wave's own stores are issued, so it does not queue behind them.
        u64 nd = 0ull;
        if (wave == kCT / kWave - 1u) {
            u32 ngid = 0u;
            if (lane == 0)
                ngid = atomicAdd(reinterpret_cast<u32 *>(a.ctrl + kCtrlTicket), 1u);
            ngid = __builtin_amdgcn_readfirstlane(ngid);
            if (lane < 4u && ngid < ngroups)
                nd = gwords[(u64)ngid * 4u + lane];
            if (lane == 3u)
                nd = (nd & 0x1FFFFFFFFull) | ((u64)ngid << 33);
        }
        if (!large) {
            for (u32 x = tid; x < ncts; x += kCT) {
                // (a group of ONE ciphertext starts at its first term: no dependent load in front of the stores)
                const u32 rel = staged ? s_coff[x] : multi ? (u32)(a.off[c0 + x] - tb) : 0u;
                a.off_out[c0 + x] = prefix + (s_rk[rel] >> 1);
            }
        } else if (chunk == 0u && tid == 0) {
            a.off_out[c0] = prefix;
        }
        if (gid == ngroups - 1u && tid == 0)
            a.off_out[a.batch] = prefix + count;
        CSGN_STAMP(9);
        // 6. registers -> their final place
        {
            u32 js = j0;
            asm volatile("" : "+v"(js));
            UnitPos pos = walk_first(walk, js);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const u32 j = js + (u32)i * kWave;
                if ((u32)i < rows && j < nunits) {
                    const u32 rk = s_rk[pos.t];
                    if (rk & 1u) {
                        if (a.nt)
                            __builtin_nontemporal_store(reg[i], out + (prefix + (rk >> 1)) * U + pos.k);
                        else
                            out[(prefix + (rk >> 1)) * U + pos.k] = reg[i];
                    }
                }
                walk_next(walk, pos);
            }
        }
        CSGN_STAMP(10);
        __syncthreads();                                          // everybody is done with s_desc and the tables
        if (wave == kCT / kWave - 1u && lane < 4u)
            s_desc[lane] = nd;
        __syncthreads();
    }
}

// ------------------------------------------------- large ciphertexts: partitions by hash
struct LargeArgs {
    const void *terms;
    const u64 *off;
    u64 *ctrl;
    const u64 *chunks;
    const u64 *plist, *slist;
    u64 *hash;
    u64 *tab;                // {tag}[2 per term]: the partitions' tags; exact path: the open-addressing table
    u32 *par;                // {index in the ciphertext}[2 per term] beside the tags; exact path: class parities
    u32 *slot_of;            // the partitions' cursors at the head of each ciphertext's share; exact path: a term's slot
    unsigned char *keepb;    // one byte per term: 1 = it survives
    u64 tag_mask;
    Geom g;
    u32 dL;
    FastDiv dGU, dTpi;       // by min(U, 64) lanes per term, by 64 / that terms per wave instruction
};

struct Chunk {
    u32 c;
    u64 c_begin, c_terms;     // the ciphertext
    u64 tb;                   // the chunk's first term
    u32 nt;
};
__device__ inline Chunk chunk_of(const LargeArgs &a, u64 ci)
{
    const u64 desc = a.chunks[ci];
    Chunk k;
    k.c = (u32)desc;
    k.c_begin = a.off[k.c];
    k.c_terms = a.off[k.c + 1] - k.c_begin;
    k.tb = k.c_begin + (desc >> 32) * a.g.capT;
    k.nt = (u32)min((u64)a.g.capT, k.c_begin + k.c_terms - k.tb);
    return k;
}

// The first read of a large ciphertext.  min(U, 64) consecutive lanes take a term (lane l of them its units l,
// l + 64, ...), 64 / U terms per wave instruction -- whole consecutive terms, 60 of 64 lanes busy at N=1247 -- and
// eight such instructions in flight.  The unit hashes meet in the wave's own LDS strip (DS operations of one
// wave run in order: no barrier), where one lane per term adds them up, writes the term's hash and sets its
// keep byte (both coalesced).  The next piece's descriptor is fetched while this one streams.
template <typename Unit>
__global__ void __launch_bounds__(256) k_cl_hash(LargeArgs a)
{
    constexpr int kFly = 8;
    __shared__ u64 s_strip[4][kFly][kWave];
    const Unit *__restrict__ terms = static_cast<const Unit *>(a.terms);
    const u32 lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, U = a.g.U;
    const u32 GU = min(U, (u32)kWave), tpi = kWave / GU, tpw = tpi * kFly;
    const u32 grp = csgn_fastdiv(lane, a.dGU), gl = lane - grp * GU;
    const bool live = grp < tpi;
    // a workgroup's piece of work: a quarter of a stripe (2048 terms: eleven turns of 4 x 48 terms at N=1247, the last
    // one two thirds full; by chunks of 1024 terms every sixth turn ran a third full)
    constexpr u32 kPiece = kStripeTerms / 4;
    // (piece pi = quarter pi / stripes of stripe pi % stripes: the first quarters -- all there is of a ciphertext of under
    // 2048 terms -- are dealt over the whole grid)
    const u64 nstripes = a.ctrl[kCtrlStripes], npieces = nstripes * 4u;
    auto piece_of = [&](u64 pi) {
        const u64 quarter = pi / nstripes;
        const u64 desc = a.slist[pi - quarter * nstripes];
        Chunk k;
        k.c = (u32)desc;
        k.c_begin = a.off[k.c];
        k.c_terms = a.off[k.c + 1] - k.c_begin;
        const u64 t_lo = min(k.c_terms, (desc >> 32) * kStripeTerms + quarter * kPiece);
        const u64 t_hi = min(k.c_terms, min(((desc >> 32) + 1u) * kStripeTerms, t_lo + kPiece));
        k.tb = k.c_begin + t_lo;
        k.nt = (u32)(t_hi - t_lo);
        return k;
    };
    if (blockIdx.x >= npieces)
        return;
    Chunk k = piece_of(blockIdx.x);
    for (u64 ci = blockIdx.x; ci < npieces; ci += gridDim.x) {
        const u64 cn = ci + gridDim.x;
        const Chunk next = piece_of(cn < npieces ? cn : ci);
        for (u32 t0 = wave * tpw; t0 < k.nt; t0 += 4u * tpw) {
            u64 acc[kFly];
#pragma unroll
            for (int i = 0; i < kFly; ++i)
                acc[i] = 0ull;
            for (u32 kk = gl; kk < U; kk += GU) {
                Unit v[kFly];
#pragma unroll
                for (int i = 0; i < kFly; ++i) {
                    const u32 t = t0 + (u32)i * tpi + grp;
                    v[i] = (live && t < k.nt) ? terms[(k.tb + t) * U + kk] : unit_zero<Unit>();
                }
#pragma unroll
                for (int i = 0; i < kFly; ++i)
                    acc[i] += unit_hash(v[i], kk);                // (dead lanes: never read back)
            }
#pragma unroll
            for (int i = 0; i < kFly; ++i)
                s_strip[wave][i][lane] = acc[i];
            __builtin_amdgcn_wave_barrier();
            for (u32 x = lane; x < tpw && t0 + x < k.nt; x += kWave) {
                // term t0 + x = instruction x / tpi, group x % tpi
                const u32 i = csgn_fastdiv(x, a.dTpi), gq = x - i * tpi;
                const u64 *row = &s_strip[wave][i][gq * GU];
                u64 s0 = 0ull, s1 = 0ull;
                u32 u = 0u;
                for (; u + 1u < GU; u += 2u) {
                    s0 += row[u];
                    s1 += row[u + 1u];
                }
                if (u < GU)
                    s0 += row[u];
                const u64 gt = k.tb + t0 + x;
                a.hash[gt] = csgn_splitmix64(s0 + s1);
                a.keepb[gt] = 1;
            }
            __builtin_amdgcn_wave_barrier();
        }
        k = next;
    }
}

// The {hash, index} pair of every term goes to the partition the top bits of the hash pick.  Scattered one by one
// the pairs cost more than the terms' first read (8 M partial-line writes for 4 M terms: 80 us of a 117 us kernel;
// and a cursor taken per term serves its thousand atomics one after the other).  So a workgroup takes a STRIPE of
// 8192 consecutive terms of one ciphertext and SORTS its pairs by partition in LDS first (count, scan, rank: LDS
// atomics), reserves its share of every partition with ONE atomic on the partition's cursor, and writes every
// partition's run of pairs side by side: whole 32- and 64-byte pieces instead of single words.  (Beyond 2048
// partitions -- ciphertexts of more than two million terms -- every term takes the cursor itself.)
constexpr u32 kScatterThreads = 1024;
inline size_t scatter_lds_bytes() { return (size_t)kStripeTerms * 12 + (size_t)(3u << kStripeMaxLp) * 4 + 64 * 4; }
__global__ void __launch_bounds__(kScatterThreads) k_cl_scatter(LargeArgs a)
{
    extern __shared__ u64 s_dyn[];
    u64 *s_h = s_dyn;                                             // [stripe] hashes, sorted by partition
    u32 *s_i = reinterpret_cast<u32 *>(s_h + kStripeTerms);       // [stripe] ... and their terms
    u32 *s_cnt = s_i + kStripeTerms;                              // [P] terms per partition, then the local cursors
    u32 *s_lbase = s_cnt + (1u << kStripeMaxLp);                  // [P] where a partition's run starts in s_h
    u32 *s_gbase = s_lbase + (1u << kStripeMaxLp);                // [P] ... and in the partition itself
    u32 *s_w = s_gbase + (1u << kStripeMaxLp);                    // [waves] scan
    constexpr int kPer = kStripeTerms / kScatterThreads;          // terms per thread: their hashes stay in registers
    const u64 nstripes = a.ctrl[kCtrlStripes];
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    bool overflow = false;
    for (u64 si = blockIdx.x; si < nstripes; si += gridDim.x) {
        const u64 desc = a.slist[si];
        const u32 c = (u32)desc;
        const u64 c_begin = a.off[c], T = a.off[c + 1] - c_begin;
        const u64 t_lo = (desc >> 32) * kStripeTerms;
        const u32 ns = (u32)(min(T, t_lo + kStripeTerms) - t_lo);
        const PartGeom pg = part_geom(T);
        const u32 cs = cursor_stride(T, pg.lp), P = 1u << pg.lp;
        u32 *cursor = a.slot_of + c_begin;
        u64 *tags = a.tab + 2 * c_begin;
        u32 *idx = a.par + 2 * c_begin;
        const bool full_tag = ~a.tag_mask == 0ull;
        if (pg.lp > kStripeMaxLp) {
            for (u32 x = tid; x < ns; x += kScatterThreads) {
                const u64 h = a.hash[c_begin + t_lo + x];
                const u32 q = (u32)(h >> (64u - pg.lp));
                const u32 pos = atomicAdd(cursor + (u64)q * cs, 1u);
                if (pos < pg.cap) {
                    tags[(u64)q * pg.cap + pos] = full_tag ? h : ((h >> 16) & a.tag_mask);
                    idx[(u64)q * pg.cap + pos] = (u32)(t_lo + x);
                } else {
                    overflow = true;
                }
            }
            continue;
        }
        for (u32 x = tid; x < P; x += kScatterThreads)
            s_cnt[x] = 0u;
        __syncthreads();
        u64 h[kPer];
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const u32 x = tid + (u32)i * kScatterThreads;
            h[i] = x < ns ? a.hash[c_begin + t_lo + x] : 0ull;
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i)
            if (tid + (u32)i * kScatterThreads < ns)
                atomicAdd(s_cnt + (pg.lp ? (u32)(h[i] >> (64u - pg.lp)) : 0u), 1u);
        __syncthreads();
        // exclusive scan of the counts (a thread takes P / 1024 consecutive partitions), and the reservations
        {
            const u32 per = (P + kScatterThreads - 1u) / kScatterThreads, b0 = tid * per;
            u32 mine = 0u;
            for (u32 j = 0; j < per; ++j)
                mine += b0 + j < P ? s_cnt[b0 + j] : 0u;
            const u32 incl = wave_incl_scan(mine);
            if (lane == kWave - 1)
                s_w[wave] = incl;
            __syncthreads();
            u32 run = incl - mine;
            for (u32 w = 0; w < wave; ++w)
                run += s_w[w];
            for (u32 j = 0; j < per; ++j)
                if (b0 + j < P) {
                    const u32 n = s_cnt[b0 + j];
                    s_lbase[b0 + j] = run;
                    s_cnt[b0 + j] = run;
                    s_gbase[b0 + j] = n ? atomicAdd(cursor + (u64)(b0 + j) * cs, n) : 0u;
                    run += n;
                }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const u32 x = tid + (u32)i * kScatterThreads;
            if (x < ns) {
                const u32 at = atomicAdd(s_cnt + (pg.lp ? (u32)(h[i] >> (64u - pg.lp)) : 0u), 1u);
                s_h[at] = h[i];
                s_i[at] = (u32)(t_lo + x);
            }
        }
        __syncthreads();
        for (u32 x = tid; x < ns; x += kScatterThreads) {
            const u64 hh = s_h[x];
            const u32 q = pg.lp ? (u32)(hh >> (64u - pg.lp)) : 0u;
            const u32 pos = s_gbase[q] + (x - s_lbase[q]);
            if (pos < pg.cap) {
                tags[(u64)q * pg.cap + pos] = full_tag ? hh : ((hh >> 16) & a.tag_mask);
                idx[(u64)q * pg.cap + pos] = s_i[x];
            } else {
                overflow = true;
            }
        }
        __syncthreads();
    }
    if (overflow)
        a.ctrl[kCtrlCollision] = 1ull;
}

// One workgroup per partition: its pairs go to LDS, are chained into buckets by tag (ONE exchange on the
// bucket's head, as in the main kernel) and every pair walks its bucket: the smallest index with the same
// tag is the class's representative, the class's size mod 2 decides.  Only terms that do NOT survive
// write their keep byte (k_cl_hash set it).  Every unit of a term that joined a class is then compared with
// the same unit of its representative (min(U, 64) lanes a pair, four pairs in flight per lane group): a
// difference is a hash collision between unequal terms and sends the call's large ciphertexts to the exact path.
template <typename Unit>
__global__ void __launch_bounds__(256) k_cl_dedup(LargeArgs a)
{
    constexpr int kFly = 4;
    __shared__ u64 s_tag[kPartCap];
    __shared__ u32 s_idx[kPartCap], s_next[kPartCap], s_head[kPartCap];
    __shared__ u32 s_njoin;
    if (a.ctrl[kCtrlCollision] != 0ull)
        return;
    const u64 nparts = a.ctrl[kCtrlParts];
    const Unit *__restrict__ terms = static_cast<const Unit *>(a.terms);
    const u32 tid = threadIdx.x, lane = tid & (kWave - 1), U = a.g.U;
    const u32 GU = min(U, (u32)kWave), ppi = kWave / GU;          // lanes per pair, pairs per wave instruction
    const u32 grp = csgn_fastdiv(lane, a.dGU), gl = lane - grp * GU;
    bool bad = false;
    for (u64 pi = blockIdx.x; pi < nparts; pi += gridDim.x) {
        const u64 desc = a.plist[pi];
        const u32 c = (u32)desc, q = (u32)(desc >> 32);
        const u64 c_begin = a.off[c], T = a.off[c + 1] - c_begin;
        const PartGeom pg = part_geom(T);
        const u32 n = min(a.slot_of[c_begin + (u64)q * cursor_stride(T, pg.lp)], pg.cap);
        const u64 base = 2 * c_begin + (u64)q * pg.cap;
        u32 nb = 1u;
        while (nb < n)
            nb <<= 1;
        if (tid == 0)
            s_njoin = 0u;
        for (u32 x = tid; x < nb; x += 256u)
            s_head[x] = 0u;
        {
            u64 tg[kPartCap / 256];
            u32 ix[kPartCap / 256];
#pragma unroll
            for (u32 r = 0; r < kPartCap / 256; ++r) {            // (all of a lane's loads in flight together)
                const u32 x = tid + r * 256u;
                tg[r] = x < n ? a.tab[base + x] : 0ull;
                ix[r] = x < n ? a.par[base + x] : 0u;
            }
#pragma unroll
            for (u32 r = 0; r < kPartCap / 256; ++r) {
                const u32 x = tid + r * 256u;
                if (x < n) {
                    s_tag[x] = tg[r];
                    s_idx[x] = ix[r];
                }
            }
        }
        __syncthreads();
        for (u32 x = tid; x < n; x += 256u)
            s_next[x] = atomicExch(s_head + ((u32)s_tag[x] & (nb - 1u)), x + 1u);
        __syncthreads();
        u32 jm[kPartCap / 256], jr[kPartCap / 256];               // a lane's terms that joined, and whom
        u32 njoin = 0u;
#pragma unroll
        for (u32 r = 0; r < kPartCap / 256; ++r) {
            const u32 x = tid + r * 256u;
            jm[r] = jr[r] = 0u;
            if (x < n) {
                const u64 tag = s_tag[x];
                const u32 mine = s_idx[x];
                u32 rep = mine, cnt = 0u;
                for (u32 e = s_head[(u32)tag & (nb - 1u)]; e != 0u; e = s_next[e - 1u])
                    if (s_tag[e - 1u] == tag) {
                        ++cnt;
                        rep = min(rep, s_idx[e - 1u]);
                    }
                if (rep != mine || !(cnt & 1u))
                    a.keepb[c_begin + mine] = 0;
                if (rep != mine) {
                    jm[r] = mine;
                    jr[r] = rep;
                    njoin |= 1u << r;
                }
            }
        }
        __syncthreads();                                          // the chains are read: heads and links become the list
        if (__ballot(njoin != 0u) != 0ull) {                      // wave-uniform
#pragma unroll
            for (u32 r = 0; r < kPartCap / 256; ++r) {
                const bool j = (njoin >> r) & 1u;
                const u64 m = __ballot(j);
                if (m != 0ull) {
                    u32 at = 0u;
                    if (lane == 0u)
                        at = atomicAdd(&s_njoin, (u32)__popcll(m));
                    at = __shfl(at, 0, kWave) + (u32)__popcll(m & ((1ull << lane) - 1ull));
                    if (j) {
                        s_head[at] = jm[r];
                        s_next[at] = jr[r];
                    }
                }
            }
        }
        __syncthreads();
        const u32 nj = s_njoin;
        if (grp < ppi)
            for (u32 j0 = (tid >> 6) * ppi * kFly; j0 < nj; j0 += 4u * ppi * kFly) {
                u64 tm[kFly], tr[kFly];
#pragma unroll
                for (int i = 0; i < kFly; ++i) {
                    const u32 j = j0 + (u32)i * ppi + grp;
                    tm[i] = c_begin + (j < nj ? s_head[j] : 0u);  // (past the list: the ciphertext's first term twice)
                    tr[i] = c_begin + (j < nj ? s_next[j] : 0u);
                }
                for (u32 kk = gl; kk < U; kk += GU) {
                    Unit x[kFly], y[kFly];
#pragma unroll
                    for (int i = 0; i < kFly; ++i) {
                        x[i] = terms[tm[i] * U + kk];
                        y[i] = terms[tr[i] * U + kk];
                    }
#pragma unroll
                    for (int i = 0; i < kFly; ++i)
                        bad |= !unit_same(x[i], y[i]);
                }
            }
        __syncthreads();
    }
    if (bad)
        a.ctrl[kCtrlCollision] = 1ull;
}

// ---- the exact path (kCtrlCollision set): an open-addressing table in HBM per large ciphertext, two slots per
// term, entries {32-bit tag, GLOBAL term index + 1}; a tag match is confirmed by comparing the terms' words
__global__ void __launch_bounds__(256) k_cl_clear(LargeArgs a)
{
    const u64 nchunks = a.ctrl[kCtrlChunks];
    if (a.ctrl[kCtrlCollision] == 0ull)
        return;
    for (u64 ci = blockIdx.x; ci < nchunks; ci += gridDim.x) {
        const Chunk k = chunk_of(a, ci);
        for (u32 x = threadIdx.x; x < 2u * k.nt; x += 256u) {
            a.tab[2 * k.tb + x] = 0ull;
            a.par[2 * k.tb + x] = 0u;
        }
    }
}

// one lane per term: insert into the ciphertext's region [2*begin, 2*(begin+terms)) of the table
__global__ void __launch_bounds__(256) k_cl_insert(LargeArgs a)
{
    const u64 nchunks = a.ctrl[kCtrlChunks];
    if (a.ctrl[kCtrlCollision] == 0ull)
        return;
    const u64 *words = static_cast<const u64 *>(a.terms);
    for (u64 ci = blockIdx.x; ci < nchunks; ci += gridDim.x) {
        const Chunk k = chunk_of(a, ci);
        const u64 base = 2 * k.c_begin, ns = 2 * k.c_terms;
        for (u32 t = threadIdx.x; t < k.nt; t += 256u) {
            const u64 g = k.tb + t, h = a.hash[g];
            u64 slot = ((u64)(u32)h * ns) >> 32;
            const u64 tag = (h >> 32) & a.tag_mask;
            const u64 entry = (tag << 32) | (g + 1);
            for (;;) {
                u64 cur = __hip_atomic_load(a.tab + base + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == 0ull) {
                    cur = atomicCAS(ull(a.tab + base + slot), 0ull, entry);
                    if (cur == 0ull)
                        break;
                }
                if ((cur >> 32) == tag &&
                    words_equal(words + g * a.dL, words + (u64)((u32)cur - 1u) * a.dL, a.dL)) {
                    atomicMin(ull(a.tab + base + slot), entry);
                    break;
                }
                slot = slot + 1 == ns ? 0 : slot + 1;
            }
            atomicXor(a.par + base + slot, 1u);
            a.slot_of[g] = (u32)(base + slot);
        }
    }
}

template <typename Unit>
hipError_t compact_launch(u32 U, u64 dL, u64 batch, u64 total_terms, u64 max_terms, const u64 *terms,
                          const u64 *off, u64 *out, u64 *off_out, void *scratch, hipStream_t s)
{
    // the wide build only on the caller's word that some ciphertext needs it and none is larger still
    const Geom narrow = make_geom(U), wide_g = make_geom(U, true);
    const bool wide_groups = max_terms > narrow.capT && max_terms <= wide_g.capT;
    const Geom g = wide_groups ? make_geom(U, true, max_terms) : make_geom(U, false, max_terms);
    const Layout l = make_layout(scratch, batch, total_terms, g);
    hipError_t e;
    // the call's head (control words + status granules) is cleared by the first kernel itself: no memset
    // node (see zero_words), no launch of its own
    const u32 cblocks = ceil_div_u64(batch, 256);
    // (gB: the geometry without the caller's bound, taken by the whole call if a ciphertext breaks the bound: k_cg_count)
    const Geom gB = make_geom(U, wide_groups);
    const u32 dual = g.wany;
    k_cg_count<<<max(cblocks, (u32)min((u64)256, l.head_bytes / 8 / 2048)), 256, 0, s>>>((u32)batch, off, g, gB, dual, l.gpos, l.gposB, l.partial,
                                                                                        l.partialB, l.ctrl, l.head_bytes / 8);
    const bool scan = cblocks > 1024u;
    if (scan)
        k_cg_scan<<<1, 1024, 0, s>>>(cblocks, l.partial, l.partialB, dual, l.ctrl);
    k_cg_fill<<<cblocks, 256, 0, s>>>((u32)batch, off, g, gB, dual, l.gpos, l.gposB, l.partial, l.partialB, l.groups, l.chunks, l.plist, l.slist,
                                       l.slot_of, l.ctrl, scan ? 1u : 0u);
    if ((e = hipGetLastError()) != hipSuccess)
        return e;

    // tag bits kept by the tables: all of them, unless a test asks for collisions
    const int bits = tune(TUNE_COMPACT_TAG_BITS);
    const u64 tag_mask = (bits > 0 && bits < 48) ? (1ull << bits) - 1ull : ~0ull;
    const u64 ng = group_bound(total_terms, g);
    const bool maybe_large = (max_terms ? max_terms : total_terms) > g.capT;
    if (maybe_large) {
        LargeArgs la;
        la.terms = terms;
        la.off = off;
        la.ctrl = l.ctrl;
        la.chunks = l.chunks;
        la.plist = l.plist;
        la.slist = l.slist;
        la.hash = l.hash;
        la.tab = l.tab;
        la.par = l.par;
        la.slot_of = l.slot_of;
        la.keepb = l.keepb;
        la.tag_mask = tag_mask;
        la.g = g;
        la.dL = (u32)dL;
        la.dGU = csgn_fastdiv_make(min(U, (u32)kWave));
        la.dTpi = csgn_fastdiv_make((u32)kWave / min(U, (u32)kWave));
        const u32 grid = (u32)min(ng, (u64)2048);
        k_cl_hash<Unit><<<grid, 256, 0, s>>>(la);
        // (per call: the attribute belongs to the current device's copy of the kernel)
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cl_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)scatter_lds_bytes())) != hipSuccess)
            return e;
        k_cl_scatter<<<(u32)min(ng + total_terms / kStripeTerms, (u64)512), kScatterThreads, scatter_lds_bytes(), s>>>(la);
        k_cl_dedup<Unit><<<(u32)min(ng + total_terms / (kPartTerms / 2), (u64)1024), 256, 0, s>>>(la);
        // (no-ops unless a partition overflowed or unequal terms shared a hash)
        k_cl_clear<<<min(grid, 256u), 256, 0, s>>>(la);
        k_cl_insert<<<min(grid, 256u), 256, 0, s>>>(la);
    }
    CompactArgs a;
    a.terms = terms;
    a.out = out;
    a.off = off;
    a.off_out = off_out;
    a.ctrl = l.ctrl;
    a.status = l.status;
    a.groups = l.groups;
    a.keepb = l.keepb;
    a.tab = l.tab;
    a.par = l.par;
    a.slot_of = l.slot_of;
    a.tag_mask = tag_mask;
    a.g = g;
    a.dU = csgn_fastdiv_make(U);
    a.batch = (u32)batch;
    a.dL = (u32)dL;
    a.large_ready = maybe_large ? 1u : 0u;
    a.nt = tune(TUNE_COMPACT_NT) ? 1u : 0u;
    a.stagger = (u32)std::max(0, tune(TUNE_COMPACT_STAGGER_US)) * 100u;
    // dev only: just past the scratch block as csgn_compact_scratch_bytes sizes it
    a.stamps = reinterpret_cast<u64 *>(static_cast<char *>(scratch) +
                                       make_layout(nullptr, batch, total_terms, make_geom((u32)min(dL, (u64)kCapUnits / 2))).bytes);
    const size_t lds = main_lds_bytes(g.capT, wide_groups ? kCRWide : kCR);
    const int grid = tune(TUNE_COMPACT_GRID) > 0 ? tune(TUNE_COMPACT_GRID) : 512;      // two workgroups per CU
    if (wide_groups)
        k_compact_main<Unit, kCRWide><<<(u32)min(ng, (u64)grid / 2), kCT, lds, s>>>(a);       // one workgroup per CU
    else
        k_compact_main<Unit, kCR><<<(u32)min(ng, (u64)grid), kCT, lds, s>>>(a);
    return hipGetLastError();
}

} // namespace

// ------------------------------------------------------------------------------ public
bool compact_supported(u64 n_bits)
{
    const u64 dL = (n_bits + 63) / 64;
    return dL <= kCapUnits / 2;                    // a group must hold two terms at least (8-byte units)
}

size_t compact_scratch_bytes(u64 n_bits, u64 batch, u64 total_terms)
{
    const u64 dL = (n_bits + 63) / 64;
    // the 8-byte-unit geometry has the smaller groups, hence the larger bound
    const Geom g = make_geom((u32)min(dL, (u64)kCapUnits / 2));
    return make_layout(nullptr, batch, total_terms, g).bytes;
}

hipError_t compact(u64 n_bits, u64 batch, u64 total_terms, u64 max_terms, const u64 *terms, const u64 *off,
                   u64 *out, u64 *off_out, void *scratch, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    if (batch == 0)
        return hipSuccess;
    // term and slot indices (up to 2 * total_terms) are kept in 32 bits
    if (batch >= (1ull << 31) || total_terms >= (1ull << 31) || !compact_supported(n_bits))
        return hipErrorInvalidValue;
    if (dL % 2 == 0 && aligned16(terms) && aligned16(out))
        return compact_launch<unit16>((u32)(dL / 2), dL, batch, total_terms, max_terms, terms, off, out, off_out,
                                      scratch, s);
    return compact_launch<unit8>((u32)dL, dL, batch, total_terms, max_terms, terms, off, out, off_out, scratch, s);
}

} // namespace csgn
