// csgn_harness.hip -- synthetic operand words and the order-sensitive digest shared with oracle/csgn_oracle.c.
// Hand-written CDNA4 (gfx950) HIP; shared helpers in csgn_device.h, design notes in DESIGN.md.
#include "csgn_device.h"

namespace csgn {

namespace {

// ---------------------------------------------------------------------------------------
// harness kernels (definitions shared with oracle/csgn_oracle.c)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_synth_fill(u64 seed, u32 dL, u64 tail, u64 first_word,
                                                    u64 n_words, u64 *__restrict__ out)
{
    u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 stride = (u64)gridDim.x * 256u;
    for (; i < n_words; i += stride) {
        const u64 idx = first_word + i;
        u64 w = csgn_splitmix64(seed + CSGN_GOLDEN * (idx + 1));
        if (idx % dL == dL - 1)
            w &= tail;
        out[i] = w;
    }
}

__global__ void __launch_bounds__(256) k_digest(const u64 *__restrict__ w, u64 n_words,
                                                u64 first_index, u64 *__restrict__ d_digest)
{
    u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    const u64 stride = (u64)gridDim.x * 256u;
    u64 acc = 0;
    for (; i < n_words; i += stride)
        acc += csgn_splitmix64(w[i] + CSGN_GOLDEN * (first_index + i + 1));
    // wave-level fold, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & (kWave - 1)) == 0 && acc != 0)
        atomicAdd(reinterpret_cast<unsigned long long *>(d_digest), acc);
}

} // namespace

// ------------------------------------------------------------------------------ public

hipError_t synth_fill(u64 seed, u64 n_bits, u64 first_word, u64 n_words, u64 *out, hipStream_t s)
{
    if (n_words == 0)
        return hipSuccess;
    const u64 dL = (n_bits + 63) / 64;
    const u32 rem = (u32)(n_bits & 63);
    const u64 tail = rem ? ~0ull << (64 - rem) : ~0ull;
    const u64 want = (n_words + 255) / 256;
    const u32 blocks = (u32)(want < 16384 ? want : 16384);
    k_synth_fill<<<blocks, 256, 0, s>>>(seed, (u32)dL, tail, first_word, n_words, out);
    return hipGetLastError();
}

hipError_t digest(const u64 *w, u64 n_words, u64 first_index, u64 *d_digest, hipStream_t s)
{
    if (n_words == 0)
        return hipSuccess;
    const u64 want = (n_words + 255) / 256;
    const u32 blocks = (u32)(want < 8192 ? want : 8192);
    k_digest<<<blocks, 256, 0, s>>>(w, n_words, first_index, d_digest);
    return hipGetLastError();
}

hipError_t circuit_zero_words(u64 *p, u64 n, hipStream_t s) { return zero_words(p, n, s); }

} // namespace csgn
