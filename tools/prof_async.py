#!/usr/bin/env python3
"""A few calls of csgn_mul_ragged_async and of plan + csgn_mul_planned on two CSR batches, for
rocprofv3 --kernel-trace --stats (dev tool: which kernel of the call sequence takes the time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath
hip = HipPath(0)
n, dl = 1247, 20
def csr(c):
    o = np.zeros(len(c) + 1, dtype=np.uint64); o[1:] = np.cumsum(np.asarray(c, dtype=np.uint64)); return o
for name, t1s, t2s in [("1M 1x1", [1] * (1 << 20), [1] * (1 << 20)), ("skewed", [1024] + [1] * 65535, [1024] + [1] * 65535)]:
    offL, offR = csr(t1s), csr(t2s)
    L = hip.synth_fill(1, n, 0, int(offL[-1]) * dl); R = hip.synth_fill(2, n, 0, int(offR[-1]) * dl)
    dL_, dR_ = hip.upload(offL), hip.upload(offR)
    tot = int(np.sum(np.asarray(t1s, dtype=np.int64) * np.asarray(t2s, dtype=np.int64)))
    out = hip.empty_words(tot * dl); off_out = hip.empty_words(len(t1s) + 1)
    plan = hip.empty_words(int(hip.lib.csgn_mul_ragged_async_plan_words(len(t1s))))
    for _ in range(5):
        hip.mul_ragged_async(n, L, dL_, R, dR_, tot, out=out, off_out=off_out, plan=plan)
    torch.cuda.synchronize()
    for _ in range(3):
        hip.mul_ragged(n, L, dL_, R, dR_)
    torch.cuda.synchronize()
    print(name, "done", flush=True)
