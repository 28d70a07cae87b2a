// runtime.h -- private glue between the certFHE:: classes and the C ABI (csgn_hip.h).
// Not installed; user code only sees include/certfhe/.
#pragma once

#include <stdint.h>

#include <memory>
#include <stdexcept>
#include <string>

#include "csgn_hip.h"

namespace certFHE {
namespace detail {

// A block of HBM owned through csgn_malloc/csgn_free.  Published payloads are immutable:
// operators allocate a new one, copies share it.
struct DevicePayload {
    void *ptr;
    uint64_t words;
    size_t capacity;      // bytes actually reserved (a size class of the block cache)
    int device;           // GPU the block lives on
    std::shared_ptr<DevicePayload> parent;   // set: this is a VIEW into parent's block (a result of a flushed queue); nothing of its own to free
    DevicePayload() : ptr(nullptr), words(0), capacity(0), device(-1) {}
    ~DevicePayload();
    DevicePayload(const DevicePayload &) = delete;
    DevicePayload &operator=(const DevicePayload &) = delete;
    uint64_t *data() const { return static_cast<uint64_t *>(ptr); }
};

// Throws std::runtime_error carrying csgn_last_error() when rc != CSGN_OK.
void check(int rc, const char *what);

// Brings up the GPU for the calling thread on first use (device: Library::useDevice,
// else $CSGN_DEVICE, else 0).  Throws when no gfx950 device is usable -- no CPU fallback.
void ensureDevice();
void selectDevice(int device);
int activeDevice();

std::shared_ptr<DevicePayload> allocBytes(size_t bytes);
std::shared_ptr<DevicePayload> allocWords(uint64_t words);
std::shared_ptr<DevicePayload> uploadWords(const uint64_t *host, uint64_t words);
void downloadBytes(void *host, const void *dev, size_t bytes);
void syncDevice();
// Two pinned host buffers per thread and device for staged transfers (kStageBytes each, made on first use):
// a copy between HBM and one of them is a DMA at the link's rate and asynchronous to the host, so it
// overlaps with whatever the host does with the other (stream I/O in Ciphertext::serialize/deserialize).
const size_t kStageBytes = (size_t)8 << 20;
void *stageBuffer(int which);
// dev -> sink / source -> dev in kStageBytes pieces through the two buffers, the transfer of one piece
// overlapping the host's handling of the previous one.  `consume(ptr, n)` / `produce(ptr, n)` see pinned memory.
void downloadStaged(const void *dev, size_t bytes, void (*consume)(void *ctx, const void *piece, size_t n), void *ctx);
void uploadStaged(void *dev, size_t bytes, void (*produce)(void *ctx, void *piece, size_t n), void *ctx);
// Pinned host byte(s) a kernel can write directly (valid after syncDevice()); *dev_alias is the
// address to hand to the kernel.  One slot per thread and device, reused by every call.
volatile unsigned char *resultSlot(void **dev_alias);
// Spins (bounded: ~200 us) until *slot differs from `armed`, then returns; past the bound, or if the byte never comes,
// synchronises the stream instead (which also surfaces a failed launch).
void awaitByte(volatile unsigned char *slot, unsigned char armed);
// Pinned (page-locked) host blocks for LARGE host mirrors (Ciphertext::getValues): a device-to-host copy into pinned
// memory is one DMA at the link's rate, into pageable memory it is staged through a bounce buffer and a host memcpy
// (8 GB/s for a 168 MB mirror, a sixth of the link).  Pinning itself is slow (tens of milliseconds for such a block),
// so freed blocks are kept per size class -- process-wide, a mirror may be dropped on any thread -- and handed out
// again; at most kPinnedPoolBytes stay cached.  *capacity receives the block's real size (pass it back to pinnedGive).
const size_t kPinnedPoolBytes = (size_t)2 << 30;
void *pinnedTake(size_t bytes, size_t *capacity);
void pinnedGive(void *host, size_t capacity);
void releasePinnedPool();
// Returns every cached HBM block of the calling thread to the driver.
void releaseBlockCache();

// ---- deferred small operations (round 5).  A kernel launch is 2-3 us of host time; a 1 x 1 product is
// 480 bytes.  operator* / operator+ on ciphertexts of at most kDeferMaxTerms terms a side therefore do not launch:
// they append a LazyNode to the calling thread's queue and return a ciphertext that points at it.  The queue is
// evaluated -- ONE csgn_small_ops launch per dependency level, results side by side in one HBM block -- when a value
// is needed (getValues, decrypt, serialize, an operation that is not deferred), when kDeferBatch operations have
// piled up, or when the thread ends.  A node keeps its operands alive (payloads are immutable), so the caller's
// objects may die or be reassigned in between; results are the words of the immediate path.
const uint32_t kDeferMaxTerms = 8;
const size_t kDeferBatch = 256;
struct DeferQueue;
struct LazyNode {
    std::shared_ptr<DevicePayload> value;          // set once evaluated (under the queue's lock)
    std::shared_ptr<DevicePayload> pa, pb;         // finished operands ...
    std::shared_ptr<LazyNode> la, lb;              // ... or operands still pending in the same queue
    uint64_t n_bits, dl;
    uint32_t t1, t2;
    bool product;
    DeferQueue *queue;                             // the queue that holds the node; nullptr once evaluated
    int index, level;                              // scratch of the flush
};
// An operand: a finished payload or a pending node (exactly one of the two).  Returns nullptr when the operation is
// not to be deferred (too large, deferral off, thread shutting down): the caller then computes at once.
std::shared_ptr<LazyNode> deferSmallOp(bool product, uint64_t n_bits, uint64_t dl, uint64_t t1, uint64_t t2,
                                       const std::shared_ptr<DevicePayload> &pa, const std::shared_ptr<LazyNode> &la,
                                       const std::shared_ptr<DevicePayload> &pb, const std::shared_ptr<LazyNode> &lb);
std::shared_ptr<DevicePayload> valueOf(const std::shared_ptr<LazyNode> &node);   // evaluates the node's queue if it has to
void flushDeferred();                  // the calling thread's queue
void setDeferral(bool on);             // process-wide switch (Library::deferSmallOperations; CSGN_NO_DEFER=1 starts off)
bool deferralOn();

inline void *stream() { return nullptr; }   // the classes run on the default stream

} // namespace detail
} // namespace certFHE
