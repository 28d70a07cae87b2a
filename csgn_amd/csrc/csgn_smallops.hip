// csgn_smallops.hip -- a list of small, independent adds and multiplies evaluated by ONE launch.
// Hand-written CDNA4 (gfx950) HIP; shared helpers in csgn_device.h.
//
// Why: BASELINE config 1 is single operations on ciphertexts of one or two terms through a value-semantic class API
// (tests/basic_operations.cpp:26-40; Ciphertext::operator* / operator+, src/Ciphertext.cpp:204-247).  The bytes are
// nothing -- 480 for a 1 x 1 product -- and a kernel launch is 2-3 us of host time, so one launch per operation is 10-20 x
// the reference's 0.12-0.28 us.  The class layer therefore QUEUES small operations and hands the queue over as a list of
// {left, right, out, t1, t2, kind} records (csgn_small_ops): one workgroup per record, the records read straight from the
// caller's (device-addressable) memory.  Results are the words the one-at-a-time kernels write.
#include "csgn_device.h"

#include "csgn_hip.h"

namespace csgn {

namespace {

template <typename Unit>
__device__ inline void small_op_body(const csgn_small_op &op, u32 U, FastDiv dU)
{
    const Unit *L = reinterpret_cast<const Unit *>(op.left), *R = reinterpret_cast<const Unit *>(op.right);
    Unit *out = reinterpret_cast<Unit *>(op.out);
    if (op.kind == 1u) {                                        // all-pairs AND (src/Ciphertext.cpp:153-163)
        const u32 total = op.t1 * op.t2 * U;
        for (u32 u = threadIdx.x; u < total; u += blockDim.x) {
            const u32 term = csgn_fastdiv(u, dU), k = u - term * U;
            const u32 i = term / op.t2, j = term - i * op.t2;
            out[u] = L[i * U + k] & R[j * U + k];
        }
    } else {                                                    // concatenation (src/Ciphertext.cpp:107-122)
        const u32 lu = op.t1 * U, total = lu + op.t2 * U;
        for (u32 u = threadIdx.x; u < total; u += blockDim.x)
            out[u] = u < lu ? L[u] : R[u - lu];
    }
}

__global__ void __launch_bounds__(128) k_small_ops(const csgn_small_op *__restrict__ ops, u32 dL, FastDiv dDL, FastDiv dHalf)
{
    const csgn_small_op op = ops[blockIdx.x];                   // workgroup-uniform: scalar loads
    const uintptr_t all = reinterpret_cast<uintptr_t>(op.left) | reinterpret_cast<uintptr_t>(op.right) |
                          reinterpret_cast<uintptr_t>(op.out);
    if ((dL & 1u) == 0u && (all & 15u) == 0u)
        small_op_body<unit16>(op, dL / 2u, dHalf);
    else
        small_op_body<unit8>(op, dL, dDL);
}

} // namespace

hipError_t small_ops(u64 n_bits, u64 count, const csgn_small_op *ops, hipStream_t s)
{
    if (count == 0)
        return hipSuccess;
    if (count > kMaxBlocks256)
        return hipErrorInvalidValue;
    const u32 dL = (u32)((n_bits + 63) / 64);
    k_small_ops<<<(u32)count, 128, 0, s>>>(ops, dL, csgn_fastdiv_make(dL), csgn_fastdiv_make(dL >= 2 ? dL / 2 : 1));
    return hipGetLastError();
}

} // namespace csgn
