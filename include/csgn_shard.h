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
 * (csgn_status) + csgn_shard_last_error(), `stream` = hipStream_t as void*.
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

/* Number of HIP devices visible to this process (no RCCL call). */
int csgn_comm_device_count(int *h_count);
/* One process driving `ndev` GPUs with one host thread each: fills comms[0..ndev) (rank i on
 * device devices[i], or device i when devices == NULL).  Each comm owns a non-blocking stream on
 * its device for callers that pass stream == NULL. */
int csgn_comm_init_all(int ndev, const int *devices, csgn_comm **comms);
/* One process per GPU: rank 0 makes the id, every rank (0 included) joins with it. */
int csgn_comm_unique_id(unsigned char h_id[CSGN_COMM_ID_BYTES]);
int csgn_comm_init_rank(const unsigned char h_id[CSGN_COMM_ID_BYTES], int rank, int world, int device,
                        csgn_comm **comm);
int csgn_comm_destroy(csgn_comm *comm);
int csgn_comm_rank(const csgn_comm *comm);
int csgn_comm_world(const csgn_comm *comm);
int csgn_comm_device(const csgn_comm *comm);
void *csgn_comm_stream(const csgn_comm *comm);       /* the comm's own hipStream_t */

/* ---------------------------------------------------------------------- the exchange ---- */

/* All ranks call this with their shard's per-pair term counts d_local[hi-lo] (csgn_shard_range of
 * total_pairs); every rank receives all `total_pairs` counts in global pair order in d_all.
 * Equal shards: one ncclAllGather.  Uneven shards (B % G != 0): one grouped ncclBroadcast per
 * rank into its slice of d_all -- still a single fused RCCL operation.  Asynchronous on `stream`
 * (NULL = the comm's own stream). */
int csgn_comm_gather_counts(csgn_comm *comm, const uint64_t *d_local, uint64_t total_pairs,
                            uint64_t *d_all, void *stream);
/* The same for one byte per pair (decrypted bits). */
int csgn_comm_gather_bytes(csgn_comm *comm, const uint8_t *d_local, uint64_t total_pairs,
                           uint8_t *d_all, void *stream);
/* Stream-ordered barrier across the ranks (a 1-element all-reduce) followed by a stream sync. */
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
