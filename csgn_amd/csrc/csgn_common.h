// csgn_common.h -- shared host/device helpers for the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

// A 16-byte "unit" (two 64-bit term words): one global_load/store_dwordx4 per lane,
// 1 KiB per wave instruction.  Used whenever dL is even; odd dL falls back to 8-byte units.
typedef __attribute__((ext_vector_type(4))) unsigned int unit16;
typedef u64 unit8;

#define CSGN_GOLDEN 0x9E3779B97F4A7C15ull

__host__ __device__ inline u64 csgn_splitmix64(u64 z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Division of a 32-bit numerator by a launch-invariant divisor without the ~30-instruction
// emulated udiv: q = (t + ((n - t) >> 1)) >> shift with t = mulhi(n, magic)
// (round-up method of Granlund & Montgomery in its branch-free 32-bit form).
struct FastDiv {
    u32 d;
    u32 magic;
    u32 shift;
};

__host__ __device__ inline FastDiv csgn_fastdiv_make(u32 d)
{
    FastDiv f;
    f.d = d;
    f.magic = 0;
    f.shift = 0;
    if (d <= 1)
        return f;                       // handled by the d==1 test in csgn_fastdiv
#if defined(__HIP_DEVICE_COMPILE__)
    u32 l = 31u - (u32)__clz((int)d);      // floor(log2 d)
#else
    u32 l = 31u - (u32)__builtin_clz(d);   // floor(log2 d)
#endif
    if ((d & (d - 1)) == 0) {           // power of two: t = 0, q = (n >> 1) >> (l-1)
        f.shift = l - 1;
        return f;
    }
    u64 num = 1ull << (32 + l);
    u64 m = num / d;
    u64 rem = num - m * d;
    m += m;
    u64 twice = rem + rem;
    if (twice >= d)
        m += 1;
    f.magic = (u32)(m + 1);
    f.shift = l;
    return f;
}

__host__ __device__ inline u32 csgn_fastdiv(u32 n, const FastDiv &f)
{
    if (f.d == 1)
        return n;
    u32 t = (u32)(((u64)n * f.magic) >> 32);   // v_mul_hi_u32 on the device
    return (t + ((n - t) >> 1)) >> f.shift;
}
