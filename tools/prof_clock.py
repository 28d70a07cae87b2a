#!/usr/bin/env python3
"""Workload for the shader-clock evidence: the issue-bound kernels (bit-plane
permutation, keyed encrypt) launched (a) after two seconds of idle and (b) right after a second of
7 TB/s all-pairs multiplies, which is where tools/bench_ops.py measures them.  Run under
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES --kernel-trace ...
(tools/prof_r03_clock.sh); tools/clock_summary.py turns the per-dispatch rows into the effective
clock GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, 'DVFS give-back') beside GB/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath

hip = HipPath(0)
T = 1024
for n, d, batch in [(1247, 16, 1 << 22), (1247, 16, 1 << 24), (4096, 32, 1 << 22)]:
    dl = hip.default_len(n)
    key = np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
    dmask, dkey = hip.upload(hip.key_mask(n, key)), hip.upload(key)
    plain = hip.upload(np.random.default_rng(2).integers(0, 2, batch).astype(np.uint8))
    fresh = hip.empty_words(batch * dl)
    perm = hip.upload(np.random.default_rng(3).permutation(n).astype(np.uint32))
    rng = hip.rng_from_seed(3, 8)
    hip.encrypt_keyed(n, d, plain, dkey, dmask, rng, out=fresh)
    pairs = 64
    L = hip.synth_fill(1, 1247, 0, pairs * T * 20)
    R = hip.synth_fill(2, 1247, 0, pairs * T * 20)
    arena = hip.empty_words(pairs * T * T * 20)
    torch.cuda.synchronize()
    for phase in ("idle", "after_multiplies"):
        if phase == "idle":
            time.sleep(2.0)
        else:
            for _ in range(60):                       # ~0.1 s per 6 launches of 10.7 GB
                hip.mul_uniform(1247, pairs, T, T, L, R, out=arena)
        # marker dispatch so the summary can tell the phases apart: a 1-word digest of phase-specific length
        hip.digest(fresh[: (1 if phase == "idle" else 2)])
        for _ in range(4):
            hip.permute_uniform(n, batch, 1, fresh, perm)
        if phase == "after_multiplies":
            for _ in range(60):
                hip.mul_uniform(1247, pairs, T, T, L, R, out=arena)
            hip.digest(fresh[:3])
        else:
            time.sleep(2.0)
            hip.digest(fresh[:4])
        for _ in range(4):
            hip.encrypt_keyed(n, d, plain, dkey, dmask, rng, out=fresh)
        torch.cuda.synchronize()
    del fresh, L, R, arena
    torch.cuda.empty_cache()
print("done")
