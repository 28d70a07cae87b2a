#!/usr/bin/env python3
"""A few launches of the ragged multiply on the log-normal batch with cold operands, and of the uniform
flat kernel at the batch's mean shape, for rocprofv3 --pmc (dev tool; tools/prof_r03_ragged.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath, check
from csgn_amd import capi

hip = HipPath(0)
n, dl = 1247, 20
rng = np.random.default_rng(0)
def csr(c):
    o = np.zeros(len(c) + 1, dtype=np.uint64); o[1:] = np.cumsum(np.asarray(c, dtype=np.uint64)); return o
t1s = np.clip(rng.lognormal(3, 1, 16384), 1, 2000).astype(int)
t2s = np.clip(rng.lognormal(3, 1, 16384), 1, 2000).astype(int)
offL, offR = csr(t1s), csr(t2s)
sets = [(hip.synth_fill(10 + k, n, 0, int(offL[-1]) * dl), hip.synth_fill(20 + k, n, 0, int(offR[-1]) * dl)) for k in range(3)]
dL_, dR_ = hip.upload(offL), hip.upload(offR)
out, off_out = hip.mul_ragged(n, sets[0][0], dL_, sets[0][1], dR_)
mt1, mt2, tot = int(t1s.max()), int(t2s.max()), int(np.sum(t1s.astype(np.int64) * t2s))
print("ragged output terms", tot, "bytes", tot * dl * 8)
for k in range(6):
    Lk, Rk = sets[k % 3]
    check(hip.lib.csgn_mul_ragged(n, len(t1s), Lk.data_ptr(), dL_.data_ptr(), Rk.data_ptr(), dR_.data_ptr(),
                                  out.data_ptr(), off_out.data_ptr(), mt1, mt2, tot, hip.stream))
torch.cuda.synchronize()
del out
# uniform 32x32 x 16384 (16.8 GB would be too much: 2048 pairs per launch through a small arena)
pairs = 16384
L = hip.synth_fill(1, n, 0, pairs * 32 * dl); R = hip.synth_fill(2, n, 0, pairs * 32 * dl)
arena = hip.empty_words(pairs * 32 * 32 * dl)
for _ in range(3):
    hip.mul_uniform(n, pairs, 32, 32, L, R, out=arena)
torch.cuda.synchronize()
print("uniform output bytes", pairs * 1024 * dl * 8)
