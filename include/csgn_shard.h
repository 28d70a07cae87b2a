/*
 * csgn_shard.h -- C ABI of libcsgn_shard.so: batch sharding across the GPUs of one node and the
 * one exchange the hot path has, an RCCL all-gather of per-pair result TERM COUNTS over xGMI.
 *
 * The reference is single-threaded and has no multi-device notion at all; this is the native
 * driver SURVEY 7 step 6 / 8(e) plan around Ciphertext::operator* (src/Ciphertext.cpp:231-247):
 * every ciphertext pair of a batch is independent, so pair p of a global batch of B goes to rank
 * p*G/B (contiguous ranges, the same answer for any GPU count), each rank multiplies its own
 * shard with csgn_mul_* (include/csgn_hip.h) into its own HBM, and the only data that crosses
 * xGMI is one uint64 per pair -- the result's term count, newlen/dL of src/Ciphertext.cpp:146 --
 * and optionally one decrypted byte per pair.  No ciphertext word ever leaves its GPU.
 *
 * libcsgn_shard.so links librccl directly (no torch, no MPI).  Two ways to form the communicator:
 *   - one process, one host thread per GPU:   csgn_comm_init_all()  (ncclCommInitAll)
 *   - one process per GPU:                    csgn_comm_unique_id() on rank 0, ship the 128 bytes
 *                                             to the others out of band, csgn_comm_init_rank().
 * Conventions are those of csgn_hip.h: POD arguments, d_* = device pointers, int status
 * (csgn_status) + csgn_shard_last_error(), `stream` = hipStream_t as void*, NULL = the legacy
 * default stream exactly as in csgn_hip.h (so a NULL handed to csgn_mul_uniform, to
 * csgn_shard_product_counts and to the gather is ONE stream and the three are ordered);
 * CSGN_STREAM_OF_COMM names the communicator's own non-blocking stream.
 *
 * Failure rules (the reference has none; a multi-rank program needs them):
 *   - no entry point blocks for ever on a dead peer: csgn_comm_barrier waits at most
 *     csgn_comm_set_timeout_ms (default 120 s), then aborts the communicator and returns
 *     CSGN_ERR_TIMEOUT;
 *   - a rank that fails calls csgn_comm_abort on its communicator (thread-per-GPU programs: on every
 *     communicator of the process): ncclCommAbort makes the peers' pending collectives return instead
 *     of waiting for the missing rank;
 *   - the communicator is created against ONE RCCL: init compares ncclGetVersion() of the library the
 *     process really bound with the NCCL_VERSION_CODE this file was compiled against and refuses a
 *     major difference always and a minor one unless the caller passes CSGN_COMM_ALLOW_MINOR_SKEW
 *     (csgn_comm_init_*_ex; the plain forms are CSGN_COMM_STRICT).  csgn_comm_rccl_info reports both
 *     versions and the library's path.
 */
#ifndef CSGN_SHARD_H
#define CSGN_SHARD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *csgn_shard_last_error(void);

/* ---------------------------------------------------------------- partition (host only) ---- */

/* [*lo, *hi) = the pairs rank `rank` of `world` owns out of `total_pairs`:
 * lo = ceil(rank*B/G), hi = ceil((rank+1)*B/G).  Shards differ by at most one pair. */
int csgn_shard_range(uint64_t total_pairs, int rank, int world, uint64_t *lo, uint64_t *hi);
/* owner of pair p: the rank whose range contains it (-1 for bad arguments). */
int csgn_shard_owner(uint64_t pair, uint64_t total_pairs, int world);

/* ----------------------------------------------------------------------- communicator ---- */

typedef struct csgn_comm csgn_comm;
#define CSGN_COMM_ID_BYTES 128
/* `stream` value that selects the communicator's own stream (csgn_comm_stream). */
#define CSGN_STREAM_OF_COMM ((void *)(intptr_t)-1)
/* status added to csgn_status for this library: a bounded wait ran out (the communicator is aborted) */
#define CSGN_ERR_TIMEOUT (-5)

/* Which RCCL is this?  *h_runtime = ncclGetVersion() of the librccl the process bound (2.27.7 ->
 * 22707), *h_header = the NCCL_VERSION_CODE libcsgn_shard.so was compiled against, h_path (<= cap
 * bytes, may be NULL) = the file that library was mapped from.  No communicator, no GPU needed. */
int csgn_comm_rccl_info(int *h_runtime, int *h_header, char *h_path, size_t cap);
/* init flags */
#define CSGN_COMM_STRICT 0u
/* accept a runtime RCCL whose MINOR version differs from the header's (same major).  For processes in
 * which another component has already mapped its own librccl (a PyTorch process: torch/lib/librccl.so),
 * where binding to that one copy is the only way to have ONE RCCL in the process.  The entry points
 * used here (ncclGetUniqueId, ncclCommInitRank/All, ncclCommDestroy/Abort, ncclCommGetAsyncError,
 * ncclAllGather, ncclBroadcast, ncclAllReduce, ncclGroupStart/End) have had the same signatures in
 * every 2.x release.  A major difference is refused whatever the flags. */
#define CSGN_COMM_ALLOW_MINOR_SKEW 1u

/* Number of HIP devices visible to this process (no RCCL call). */
int csgn_comm_device_count(int *h_count);
/* One process driving `ndev` GPUs with one host thread each: fills comms[0..ndev) (rank i on
 * device devices[i], or device i when devices == NULL).  Each comm owns a non-blocking stream on
 * its device (csgn_comm_stream; pass it, or CSGN_STREAM_OF_COMM, to use it).  The calling thread's
 * current device is the same after the call as before.  On failure comms[] holds what was made
 * (NULL or a communicator each): pass every non-NULL one to csgn_comm_destroy. */
int csgn_comm_init_all(int ndev, const int *devices, csgn_comm **comms);               /* flags = CSGN_COMM_STRICT */
int csgn_comm_init_all_ex(int ndev, const int *devices, unsigned flags, csgn_comm **comms);
/* One process per GPU: rank 0 makes the id, every rank (0 included) joins with it.  csgn_comm_init_rank*
 * leaves the calling thread ON `device` (the thread that owns the rank).  Rank PROCESSES map one another's
 * buffers through HIP IPC: on hosts whose driver only supports dmabuf IPC (this pool), export
 * HSA_ENABLE_IPC_MODE_LEGACY=0 before the process makes its first HIP call -- the library cannot set it
 * late enough to matter. */
int csgn_comm_unique_id(unsigned char h_id[CSGN_COMM_ID_BYTES]);
int csgn_comm_init_rank(const unsigned char h_id[CSGN_COMM_ID_BYTES], int rank, int world, int device,
                        csgn_comm **comm);                                             /* flags = CSGN_COMM_STRICT */
int csgn_comm_init_rank_ex(const unsigned char h_id[CSGN_COMM_ID_BYTES], int rank, int world, int device,
                           unsigned flags, csgn_comm **comm);
/* Waits for the communicator's stream (unless aborted), frees it; the caller's current device is unchanged. */
int csgn_comm_destroy(csgn_comm *comm);
/* ncclCommAbort: releases every peer blocked in a collective with this communicator's rank missing.
 * Callable from any host thread, any number of times; afterwards only csgn_comm_destroy is valid.
 * ncclCommAbort frees the RCCL handle, so the call is serialised against the owner's use of it: a thread
 * that is inside csgn_comm_gather_*, csgn_comm_barrier or csgn_comm_check at that moment finishes its
 * (short, asynchronous) RCCL call first and the last one out performs the abort; later calls fail with
 * "communicator was aborted".  Only an owner that stays inside RCCL for more than two seconds (blocked
 * on a peer that never connects) is aborted under its feet -- the case ncclCommAbort exists for. */
int csgn_comm_abort(csgn_comm *comm);
/* CSGN_OK while the communicator is healthy; CSGN_ERR_HIP with the RCCL error text once an
 * asynchronous error was recorded (a peer died, a transport failed) or it was aborted. */
int csgn_comm_check(csgn_comm *comm);
/* Longest wait of csgn_comm_barrier in milliseconds (0 = wait for ever; default 120 000). */
int csgn_comm_set_timeout_ms(csgn_comm *comm, uint64_t timeout_ms);
/* Per-communicator options (never process-wide).  CSGN_COMM_OPT_FORCE_GROUPED_BROADCAST != 0 makes
 * the gathers take the uneven-shard form (one ncclBroadcast per rank in a group) even when the
 * shards are equal: the same result, used by the tests to run that branch at any world size. */
#define CSGN_COMM_OPT_FORCE_GROUPED_BROADCAST 1
int csgn_comm_set_option(csgn_comm *comm, int option, int value);
int csgn_comm_rank(const csgn_comm *comm);
int csgn_comm_world(const csgn_comm *comm);
int csgn_comm_device(const csgn_comm *comm);
void *csgn_comm_stream(const csgn_comm *comm);       /* the comm's own hipStream_t */

/* ---------------------------------------------------------------------- the exchange ---- */

/* The layout of a gather of `total_pairs` elements over `world` ranks, on the host: rank r
 * contributes h_len[r] elements that land at h_lo[r] of the gathered array ( = csgn_shard_range);
 * *h_equal = 1 when every h_len is the same, i.e. the exchange is a single ncclAllGather, 0 when it is
 * the grouped broadcast form.  h_lo, h_len: `world` entries each.  This is the arithmetic
 * csgn_comm_gather_* run on; exported so that it can be checked for every world size without a GPU. */
int csgn_shard_gather_plan(uint64_t total_pairs, int world, uint64_t *h_lo, uint64_t *h_len, int *h_equal);

/* All ranks call this with their shard's per-pair term counts d_local[hi-lo] (csgn_shard_range of
 * total_pairs); every rank receives all `total_pairs` counts in global pair order in d_all.
 * Equal shards: one ncclAllGather.  Uneven shards (B % G != 0): one grouped ncclBroadcast per
 * rank into its slice of d_all -- still a single fused RCCL operation.  Asynchronous on `stream`
 * (NULL = the legacy default stream, CSGN_STREAM_OF_COMM = the comm's own): enqueue the producer of
 * d_local on the same stream, or order the two yourself. */
int csgn_comm_gather_counts(csgn_comm *comm, const uint64_t *d_local, uint64_t total_pairs,
                            uint64_t *d_all, void *stream);
/* The same for one byte per pair (decrypted bits). */
int csgn_comm_gather_bytes(csgn_comm *comm, const uint8_t *d_local, uint64_t total_pairs,
                           uint8_t *d_all, void *stream);
/* Stream-ordered barrier across the ranks (a 1-element all-reduce) followed by a wait for the stream
 * that is bounded by csgn_comm_set_timeout_ms: on expiry the communicator is aborted and the call
 * returns CSGN_ERR_TIMEOUT. */
int csgn_comm_barrier(csgn_comm *comm, void *stream);

/* Per-pair result term counts of a multiply from the operands' CSR offsets, on the device:
 * d_counts[b] = (offL[b+1]-offL[b]) * (offR[b+1]-offR[b])   (src/Ciphertext.cpp:146, newlen/dL).
 * For a uniform batch pass d_off_left = d_off_right = NULL and t1, t2: every count is t1*t2. */
int csgn_shard_product_counts(uint64_t batch, const uint64_t *d_off_left, const uint64_t *d_off_right,
                              uint64_t t1, uint64_t t2, uint64_t *d_counts, void *stream);

#ifdef __cplusplus
}
#endif
#endif
