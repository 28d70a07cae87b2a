// certfhe/Helpers.h -- Library / Helper statics of the drop-in certFHE API.
// Public surface of /root/reference/src/Helpers.h:21,38,43.
#ifndef CERTFHE_HELPERS_H
#define CERTFHE_HELPERS_H

#include "utils.h"

namespace certFHE {

class Library {
    Library() {}

  public:
    // Seeds libc's rand() from the wall clock, as the reference does
    // (src/Helpers.cpp:8-12).  The GPU runtime is brought up lazily on first device use.
    static void initializeLibrary();

    // --- extensions (not in the reference) ---
    // Select the GPU used by this thread's certFHE objects (default: $CSGN_DEVICE or 0).
    // Throws std::runtime_error when no gfx950 device is usable: there is no CPU fallback.
    static void useDevice(int device);
    static int currentDevice();
    // Freed ciphertext buffers are cached per size class for re-use (hipMalloc/hipFree are
    // slow and hipFree synchronises); this hands the cached HBM back to the driver.
    static void releaseDeviceCache();
    // operator* / operator+ on ciphertexts of at most 8 terms a side are QUEUED per thread and evaluated together -- one
    // launch per 256 operations instead of one each -- when a value is looked at (getValues, decrypt, serialize, a larger
    // operation) or the queue is full; results and their words are unchanged.  On by default (CSGN_NO_DEFER=1 in the
    // environment starts with it off); deferSmallOperations(false) evaluates what is queued and computes at once from
    // then on; flush() evaluates the calling thread's queue now (e.g. before timing something else).
    static void deferSmallOperations(bool on);
    static void flush();
};

class Helper {
    Helper() {}

  public:
    // Linear membership test (src/Helpers.cpp:18-26).
    static bool exists(const uint64_t *v, const uint64_t len, const uint64_t value);

    // Kept for source compatibility (src/Helpers.h:43).  The reference deletes a void*,
    // which is undefined; here the pointer must be a uint64_t buffer obtained with
    // new / new[].  No certFHE getter returns memory the caller owns.
    static void deletePointer(void *pointer, bool isArray);
};

} // namespace certFHE

#endif
