// csgn_capi_util.h -- what the translation units of the extern "C" surface share (csgn_capi.hip, csgn_circuit.hip):
// the thread-local error message, argument checks, the derivation of a circuit encrypt node's key.  Internal.
#pragma once

#include "csgn_hip.h"
#include "csgn_kernels.h"

namespace csgn {
namespace capi {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int hip_fail(hipError_t e, const char *what);
bool product_below(uint64_t a, uint64_t b, uint64_t c, uint64_t limit);   // a*b*c < limit without wrapping
int check_n(uint64_t n_bits);                                              // shape limit shared by every entry point
void node_key_from(const csgn_rng &rng, uint32_t node_key[8]);

inline hipStream_t S(void *stream) { return reinterpret_cast<hipStream_t>(stream); }

} // namespace capi
} // namespace csgn

#define HIP_TRY(expr)                              \
    do {                                           \
        hipError_t e_ = (expr);                    \
        if (e_ != hipSuccess)                      \
            return csgn::capi::hip_fail(e_, #expr); \
    } while (0)

#define REQUIRE(cond, ...)                                           \
    do {                                                             \
        if (!(cond))                                                 \
            return csgn::capi::fail(CSGN_ERR_INVALID, __VA_ARGS__);  \
    } while (0)
