// csgn_tuning.h -- the library's tuning knobs (kernel choice and sweep parameters).
//
// Every knob is one int PER HOST THREAD (thread_local): setting it changes the calling thread's later
// dispatches and nobody else's.  Its start value in every thread is the built-in default, or the
// value of the environment variable CSGN_<KEY IN CAPITALS> sampled ONCE when libcsgn_hip.so is
// loaded; after that only csgn_set_tuning()/csgn_reset_tuning() (include/csgn_hip.h) change it.
// No compute entry point reaches getenv.  Knobs select among kernels that compute the same
// words: results never depend on them (tests/test_gpu_parity.py drives every form through them).
#pragma once

namespace csgn {

enum TuneKey {
    TUNE_MUL_M,          // tiled multiply: column units per lane, 0 = auto
    TUNE_MUL_TI,         // tiled multiply: left terms per LDS tile
    TUNE_MUL_NT,         // tiled multiply: 1 = non-temporal stores
    TUNE_MUL_FLAT,       // 0 = choose per shape; k > 0 = flat kernel, k units per lane; -1 = tiled kernel
    TUNE_MUL_BS,         // tiled multiply: block size override (0 = auto)
    TUNE_MUL_XCD,        // XCD-contiguous block order: 0 off, 1 flat kernel, 2 both kernels
    TUNE_MUL_TOUCH,      // operand touch pass: -1 = per shape, 0..3 = bit 0 left, bit 1 right operand
    TUNE_MUL_PF_KB,      // flat multiply without touch: left-term prefetch distance in KiB, -1 = auto
    TUNE_STREAM_XCD,     // 1x1 multiply / uniform add block order: -1 = by size, 0 / 1 forced
    TUNE_RAGGED_C,       // flat ragged kernels: 4 KiB chunks per workgroup, 0 = auto
    TUNE_RAGGED_FLAT,    // 1 = never use the tiled kernel for ragged batches
    TUNE_RAGGED_PF,      // ragged multiply: operand prefetch distance in pairs, 0 = off
    TUNE_RAGGED_TOUCH,   // ragged multiply: 1 = touch pass per 1 GiB output slice
    TUNE_RAGGED_M,       // flat ragged multiply: 4 KiB chunks per turn that share one speculative pair lookup and whose operand loads travel together: 1, 2 or 4
    TUNE_PERM_BALLOT,    // 1 = ballot bit-gather permutation kernel instead of the bit-plane kernel
    TUNE_PERM_NARROW,    // 1 = 8-byte staging accesses in the bit-plane kernel
    TUNE_PERM_WAVES,     // bit-plane kernel: waves per 64-term group; 0 = auto
    TUNE_PERM_PERSIST,   // bit-plane kernel: 1 = persistent workgroups striding over the groups, 0 = one group per workgroup, k > 1 = k groups per CU resident
    TUNE_DEC_LOOP,       // 1 = looping 256-term decrypt pass 1 instead of the segment form
    TUNE_ENC_LDS,        // 1 = LDS-staged encrypt kernel instead of the segment form
    TUNE_ENC_WAVE,       // device-RNG encrypt: 1 = wave-local kernel (default), 0 = segment kernel
    TUNE_ENC_COMPACT,    // keyed encrypt: compact LDS tables: -1 = auto (groups of 3+ passes), 0 / 1 forced
    TUNE_SHARED_GPU,     // 1 = other work streams through this GPU's HBM beside the caller: all-pairs multiplies take the LDS-tiled kernel instead of the operand-touch + flat pair, which depends on the memory-side cache (include/csgn_hip.h, "Sharing the GPU")
    TUNE_COMPACT_TAG_BITS, // compaction: hash-tag bits the tables keep, 0 = all (a test narrows them to force the collision path)
    TUNE_COMPACT_NT,     // compaction: 1 = non-temporal stores of the surviving terms
    TUNE_COMPACT_GRID,   // compaction: workgroups of the main kernel, 0 = 512 (two per CU)
    TUNE_RAGGED_CLASSES, // planned ragged multiply: 1 = the plan lists small pairs by size class and the multiply gives every class its own tiled launch (measured slower than the CSR kernel: off by default)
    TUNE_RAGGED_COOP,    // ragged multiply of small pairs: 1 = the wave-cooperative kernel (a wave walks its pairs together: no per-lane search or division), 0 = the CSR kernel, -1 = auto (the default: ranges that average 16 product terms a pair and more)
    TUNE_RAGGED_COOP_SPAN, // ... 64-unit blocks of output per wave (0 = 64)
    TUNE_RAGGED_COOP_K,  // ... blocks per group of the software pipeline: 2 or 4 (0 = 4)
    TUNE_RAGGED_COOP_TOUCH, // ... a wave touches the operands of an offset window.s pairs when it loads the window (one dword per 128-byte line): KiB a side at most, 0 = off
    TUNE_RAGGED_SLICE_MB, // ragged multiply: output slice size in MiB, each slice behind a touch of its operands (0 = by the operand share: 1 GiB, 512 MiB or unsliced)
    TUNE_RAGGED_COOP_PIPE, // ... 1 = software-pipelined form (loads of the next group in front of the stores of this one; hand-counted waits), 0 = loads, wait, stores (the default: level with the pipelined form since the operand touch)
    TUNE_RAGGED_XCD_GROUP, // CSR multiply / add: logical blocks per XCD turn (xcd_grouped_block); 0 = one contiguous eighth of the launch per XCD
    TUNE_RAGGED_COOP_XCD_GROUP, // wave-cooperative multiply: workgroups per XCD turn (xcd_grouped_block); 0 = contiguous eighths
    TUNE_COMPACT_STAGGER_US, // compaction: microseconds by which every second workgroup of the main kernel starts late (phases of the two workgroups of a CU interleave), 0 = together
    TUNE_ZERO_MEMSET,    // dev: 1 = zero fills on capturable paths are hipMemsetAsync (memset NODES in a circuit's graph) instead of the k_zero_words kernel (csgn_device.h, zero_words; tools/graph_memset_probe.hip)
    TUNE_COUNT
};

int tune(TuneKey k);                      // the calling thread's value
const char *tune_name(int k);             // "mul_m", ... (nullptr past the end)
bool tune_set(const char *name, int value);
bool tune_get(const char *name, int *value);
void tune_reset();                        // defaults, then the environment snapshot taken at load

} // namespace csgn
