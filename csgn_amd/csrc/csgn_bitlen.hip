// csgn_bitlen.hip -- decrypt and permutation of ONE ciphertext that carries an explicit `bitlen` side
// array (built through the reference's 4-argument constructor / setBitlen with a pattern other than
// the canonical 64,...,64,N%64).  The reference treats (v, bitlen) as a bit STREAM: word i contributes
// its top bitlen[i] bits (src/SecretKey.cpp:110-124, src/Ciphertext.cpp:16-31), and addresses the
// stream at flat positions n*k + s[i].  Here the stream is never unpacked: an exclusive prefix sum of
// bitlen gives every word's first stream position, and a position is found by a galloping search
// that starts at word q/64 (bitlen <= 64, so the word of position q is never before it).
// Hand-written CDNA4 (gfx950) HIP; a correctness path (any bit pattern the class API can build
// decrypts on the device), not a bandwidth one.
#include "csgn_device.h"

namespace csgn {

namespace {

__device__ inline u64 clamp_bitlen(u64 b) { return b > 64u ? 64u : b; }   // > 64 is undefined in the reference

// exclusive scan of min(bitlen, 64) over `len` words: per-1024-word chunk scans, a scan of the chunk
// totals, and the fix-up (the shape of the ragged multiply's planner)
__global__ void __launch_bounds__(256) k_bitlen_chunks(u64 len, const u64 *__restrict__ bitlen,
                                                       u64 *__restrict__ pos, u64 *__restrict__ partial)
{
    __shared__ u64 sums[256];
    const u32 tid = threadIdx.x;
    const u64 b0 = (u64)blockIdx.x * 1024u + (u64)tid * 4u;
    u64 c[4], mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = (b0 + j < len) ? clamp_bitlen(bitlen[b0 + j]) : 0;
        mine += c[j];
    }
    sums[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        u64 run = 0;
        for (u32 t = 0; t < 256; ++t) {
            const u64 v = sums[t];
            sums[t] = run;
            run += v;
        }
        partial[blockIdx.x] = run;
    }
    __syncthreads();
    u64 run = sums[tid];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (b0 + j < len)
            pos[b0 + j] = run;
        run += c[j];
    }
}

__global__ void __launch_bounds__(1024) k_bitlen_partials(u64 nchunks, u64 len, u64 *__restrict__ partial,
                                                          u64 *__restrict__ pos)
{
    __shared__ u64 part[1024];
    const u32 tid = threadIdx.x;
    const u64 chunk = (nchunks + 1023) / 1024;
    const u64 c0 = min(nchunks, (u64)tid * chunk), c1 = min(nchunks, c0 + chunk);
    u64 sum = 0;
    for (u64 c = c0; c < c1; ++c)
        sum += partial[c];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        u64 run = 0;
        for (u32 t = 0; t < 1024; ++t) {
            const u64 v = part[t];
            part[t] = run;
            run += v;
        }
        pos[len] = run;                                    // total stream length
    }
    __syncthreads();
    u64 run = part[tid];
    for (u64 c = c0; c < c1; ++c) {
        const u64 v = partial[c];
        partial[c] = run;
        run += v;
    }
}

__global__ void __launch_bounds__(256) k_bitlen_fix(u64 len, const u64 *__restrict__ partial, u64 *__restrict__ pos)
{
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < len)
        pos[i] += partial[i >> 10];
}

// bit at stream position q (0 when q is past the end, as the oracle defines the reference's
// out-of-bounds read)
__device__ inline u32 stream_bit(const u64 *__restrict__ v, const u64 *__restrict__ pos, u64 len, u64 q)
{
    if (q >= pos[len])
        return 0u;
    // largest w in [q/64, len) with pos[w] <= q; words of zero bitlen share their successor's position,
    // the LAST one with pos[w] <= q is the one that holds the bit
    u64 lo = q >> 6, step = 1;
    if (lo >= len)
        lo = len - 1;
    u64 hi = lo + 1;
    while (hi < len && pos[hi] <= q) {
        lo = hi;
        step <<= 1;
        hi = (len - lo > step) ? lo + step : len;
    }
    while (hi - lo > 1) {
        const u64 mid = lo + ((hi - lo) >> 1);
        if (pos[mid] <= q)
            lo = mid;
        else
            hi = mid;
    }
    return (u32)(v[lo] >> (63u - (u32)(q - pos[lo]))) & 1u;
}

// XOR over terms k of (AND over key indices of the bit at n*k + s[i]); src/SecretKey.cpp:126-140
__global__ void __launch_bounds__(256) k_decrypt_stream(u64 n_bits, u64 terms, u64 len, u64 d,
                                                        const u64 *__restrict__ v, const u64 *__restrict__ pos,
                                                        const u64 *__restrict__ key, u32 *__restrict__ parity)
{
    const u64 k = (u64)blockIdx.x * 256u + threadIdx.x;
    u32 dec = 0;
    if (k < terms) {
        dec = 1;
        for (u64 i = 0; i < d && dec; ++i)
            dec &= stream_bit(v, pos, len, n_bits * k + key[i]);
    }
    const u64 odd = __ballot(dec);
    if ((threadIdx.x & (kWave - 1)) == 0 && (__popcll(odd) & 1))
        atomicXor(parity, 1u);
}

__global__ void k_parity_to_byte(const u32 *__restrict__ parity, uint8_t *__restrict__ bit)
{
    *bit = (uint8_t)(*parity & 1u);
}

// new bit j = stream bit perm[j] for j < min(N, stream length); one lane per output bit, one wave
// per output word (src/Ciphertext.cpp:33-69: the result is ONE term whatever the input holds)
__global__ void __launch_bounds__(64) k_permute_stream(u64 n_bits, u64 len, const u64 *__restrict__ v,
                                                       const u64 *__restrict__ pos, const u32 *__restrict__ perm,
                                                       u64 *__restrict__ out)
{
    const u32 lane = threadIdx.x;
    const u64 j = (u64)blockIdx.x * 64u + (63u - lane);    // ballot bit l <-> word bit l
    u32 b = 0;
    if (j < n_bits && j < pos[len])
        b = stream_bit(v, pos, len, perm[j]);
    const u64 word = __ballot(b);
    if (lane == 0)
        out[blockIdx.x] = word;
}

} // namespace

// scratch: [pos: len+1 words][partial: ceil(len/1024)+1 words][parity: 1 word]
size_t bitlen_scratch_bytes(u64 len) { return ((size_t)len + 1 + (len + 1023) / 1024 + 1 + 2) * 8; }

static hipError_t stream_positions(u64 len, const u64 *bitlen, u64 *pos, u64 *partial, hipStream_t s)
{
    const u64 nchunks = (len + 1023) / 1024;
    if (nchunks > kMaxBlocks256)
        return hipErrorInvalidValue;
    if (len)
        k_bitlen_chunks<<<(u32)nchunks, 256, 0, s>>>(len, bitlen, pos, partial);
    k_bitlen_partials<<<1, 1024, 0, s>>>(nchunks, len, partial, pos);
    if (len)
        k_bitlen_fix<<<ceil_div_u64(len, 256), 256, 0, s>>>(len, partial, pos);
    return hipGetLastError();
}

hipError_t decrypt_bitlen(u64 n_bits, u64 d, u64 len, const u64 *v, const u64 *bitlen, const u64 *key,
                          uint8_t *bit, void *scratch, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    u64 *pos = reinterpret_cast<u64 *>(scratch);
    u64 *partial = pos + len + 1;
    u32 *parity = reinterpret_cast<u32 *>(partial + (len + 1023) / 1024 + 1);
    hipError_t e = zero_words(reinterpret_cast<u64 *>(parity), 1, s);      // (a kernel, as every zero fill of a compute path: csgn_device.h)
    if (e != hipSuccess)
        return e;
    const u64 terms = len / dL;                            // src/SecretKey.cpp:126 (times = len/defLen)
    if (terms) {
        if ((e = stream_positions(len, bitlen, pos, partial, s)) != hipSuccess)
            return e;
        if ((terms + 255) / 256 > kMaxBlocks256)
            return hipErrorInvalidValue;
        k_decrypt_stream<<<ceil_div_u64(terms, 256), 256, 0, s>>>(n_bits, terms, len, d, v, pos, key, parity);
    }
    k_parity_to_byte<<<1, 1, 0, s>>>(parity, bit);
    return hipGetLastError();
}

hipError_t permute_bitlen(u64 n_bits, u64 len, const u64 *v, const u64 *bitlen, const u32 *perm, u64 *out,
                          void *scratch, hipStream_t s)
{
    const u64 dL = (n_bits + 63) / 64;
    u64 *pos = reinterpret_cast<u64 *>(scratch);
    u64 *partial = pos + len + 1;
    hipError_t e = stream_positions(len, bitlen, pos, partial, s);
    if (e != hipSuccess)
        return e;
    if (len == 0)
        return zero_words(out, dL, s);
    k_permute_stream<<<(u32)dL, 64, 0, s>>>(n_bits, len, v, pos, perm, out);
    return hipGetLastError();
}

} // namespace csgn
