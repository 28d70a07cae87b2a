#!/usr/bin/env python3
"""The fused fresh chain (csgn_encrypt_mul_keyed: Enc, Enc, * and Dec in ONE kernel) beside the same
chain as separate calls (csgn_encrypt_keyed x2, csgn_mul_uniform, csgn_decrypt_uniform), BASELINE
configs 2 and 4's shapes: 65 536 and 1 M fresh pairs at N=1247 (and N=4096).  Dev tool.

    python tools/bench_fused.py

Per pair the fused kernel writes 8*dL bytes (+1 byte); the unfused chain moves 5x that.  Times are
HIP events around the whole call sequence on one stream (median of 9), i.e. launch gaps included.
"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath

hip = HipPath(0)


def timed(fn, rounds=9):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) / 1e3)
    return statistics.median(ts), min(ts)


for n, d in [(1247, 16), (4096, 32)]:
    dl = hip.default_len(n)
    key = np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
    dmask, dkey = hip.upload(hip.key_mask(n, key)), hip.upload(key)
    ra, rb = hip.rng_from_seed(3, 8), hip.rng_from_seed(4, 8)
    for batch in (1 << 16, 1 << 20, 1 << 22):
        pa = hip.upload(np.random.default_rng(2).integers(0, 2, batch).astype(np.uint8))
        pb = hip.upload(np.random.default_rng(3).integers(0, 2, batch).astype(np.uint8))
        ea, eb, prod = hip.empty_words(batch * dl), hip.empty_words(batch * dl), hip.empty_words(batch * dl)

        def unfused():
            hip.encrypt_keyed(n, d, pa, dkey, dmask, ra, out=ea)
            hip.encrypt_keyed(n, d, pb, dkey, dmask, rb, out=eb)
            hip.mul_uniform(n, batch, 1, 1, ea, eb, out=prod)
            return hip.decrypt_uniform(n, batch, 1, prod, dmask)

        tu, tu_min = timed(unfused)
        tf, tf_min = timed(lambda: hip.encrypt_mul_keyed(n, d, pa, pb, dkey, dmask, ra, rb))
        tn, tn_min = timed(lambda: hip.encrypt_mul_keyed(n, d, pa, pb, dkey, dmask, ra, rb, with_bits=False))
        print(f"N={n} pairs={batch}: unfused Enc,Enc,*,Dec {tu*1e6:8.1f} us (min {tu_min*1e6:7.1f}) | "
              f"fused incl. Dec {tf*1e6:8.1f} us (min {tf_min*1e6:7.1f}) = {batch/tf/1e9:5.2f} G fresh mult/s, "
              f"{batch*dl*8/tf/1e9:6.0f} GB/s of product | fused, no bits {tn*1e6:8.1f} us | speed-up {tu/tf:4.2f}x", flush=True)
        del ea, eb, prod
        torch.cuda.empty_cache()
