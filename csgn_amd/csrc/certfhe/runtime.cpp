// runtime.cpp -- device bring-up, error translation and HBM buffers for the certFHE:: classes.
#include "runtime.h"

#include <cstdlib>

namespace certFHE {
namespace detail {

namespace {
thread_local int g_device = -1;       // -1: not initialised on this thread
thread_local int g_requested = -1;    // Library::useDevice
}

void check(int rc, const char *what)
{
    if (rc == CSGN_OK)
        return;
    std::string msg = std::string("certFHE (MI355X): ") + what + " failed [" + std::to_string(rc) +
                      "]: " + csgn_last_error();
    throw std::runtime_error(msg);
}

void selectDevice(int device)
{
    g_requested = device;
    g_device = -1;
    ensureDevice();
}

int activeDevice()
{
    ensureDevice();
    return g_device;
}

void ensureDevice()
{
    if (g_device >= 0)
        return;
    int dev = g_requested;
    if (dev < 0) {
        const char *env = getenv("CSGN_DEVICE");
        dev = (env && *env) ? atoi(env) : 0;
    }
    check(csgn_init(dev), "csgn_init");
    g_device = dev;
    // One round trip through every kind of call the classes make.  Measured on ROCm 7.2: the
    // first host-to-device copy of a process draws from libc's rand(); later calls do not.
    // Doing it here keeps a caller's `srand(seed); encrypt...` sequence bit-reproducible
    // (SecretKey's constructor and Library::initializeLibrary reach this point first).
    void *scratch = nullptr;
    uint64_t probe[4] = {1, 2, 3, 4};
    check(csgn_malloc(&scratch, sizeof(probe)), "csgn_malloc");
    check(csgn_memcpy_h2d(scratch, probe, sizeof(probe), stream()), "csgn_memcpy_h2d");
    check(csgn_synth_fill(0, 64, 0, 4, static_cast<uint64_t *>(scratch), stream()), "csgn_synth_fill");
    check(csgn_memcpy_d2h(probe, scratch, sizeof(probe), stream()), "csgn_memcpy_d2h");
    check(csgn_free(scratch), "csgn_free");
}

DevicePayload::~DevicePayload()
{
    if (ptr)
        csgn_free(ptr);   // nothing useful to do with a failure in a destructor
}

std::shared_ptr<DevicePayload> allocBytes(size_t bytes)
{
    ensureDevice();
    std::shared_ptr<DevicePayload> p = std::make_shared<DevicePayload>();
    if (bytes)
        check(csgn_malloc(&p->ptr, bytes), "csgn_malloc");
    p->words = bytes / 8;
    return p;
}

std::shared_ptr<DevicePayload> allocWords(uint64_t words) { return allocBytes((size_t)words * 8); }

std::shared_ptr<DevicePayload> uploadWords(const uint64_t *host, uint64_t words)
{
    std::shared_ptr<DevicePayload> p = allocWords(words);
    if (words) {
        check(csgn_memcpy_h2d(p->ptr, host, (size_t)words * 8, stream()), "csgn_memcpy_h2d");
        check(csgn_stream_sync(stream()), "csgn_stream_sync");   // host buffer may die right after
    }
    return p;
}

void downloadBytes(void *host, const void *dev, size_t bytes)
{
    if (bytes)
        check(csgn_memcpy_d2h(host, dev, bytes, stream()), "csgn_memcpy_d2h");
}

void syncDevice()
{
    if (g_device >= 0)
        check(csgn_stream_sync(stream()), "csgn_stream_sync");
}

} // namespace detail
} // namespace certFHE
