// Helpers.cpp -- Library / Helper statics (public surface of the reference's Helpers.h).
#include "Helpers.h"

#include <ctime>

#include "runtime.h"

namespace certFHE {

void Library::initializeLibrary()
{
    // device first (HIP start-up consumes rand() draws of its own), then the same observable
    // effect as src/Helpers.cpp:8-12: libc's generator seeded from the clock
    detail::ensureDevice();
    srand((unsigned)time(NULL));
}

void Library::useDevice(int device) { detail::selectDevice(device); }

int Library::currentDevice() { return detail::activeDevice(); }

void Library::releaseDeviceCache() { detail::releaseBlockCache(); }

void Library::deferSmallOperations(bool on) { detail::setDeferral(on); }

void Library::flush() { detail::flushDeferred(); }

bool Helper::exists(const uint64_t *v, const uint64_t len, const uint64_t value)
{
    for (uint64_t i = 0; i < len; ++i)
        if (v[i] == value)
            return true;
    return false;
}

void Helper::deletePointer(void *pointer, bool isArray)
{
    if (!pointer)
        return;
    uint64_t *p = static_cast<uint64_t *>(pointer);
    if (isArray)
        delete[] p;
    else
        delete p;
}

} // namespace certFHE
