#!/usr/bin/env python3
"""csgn_mul_ragged_async on the skewed batch (one 1024x1024 pair + 65 535 singles), 60 calls on cold operand sets:
under `rocprofv3 --kernel-trace` the trace's start/end stamps give the kernels of one call and the idle time between
them (tools/prof_async_timeline.sh prints that).  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath
hip = HipPath(0)
n, dl = 1247, 20
def csr(c):
    o = np.zeros(len(c) + 1, dtype=np.uint64); o[1:] = np.cumsum(np.asarray(c, dtype=np.uint64)); return o
t1s = t2s = [1024] + [1] * 65535
offL, offR = csr(t1s), csr(t2s)
sets = [(hip.synth_fill(1 + 2 * k, n, 0, int(offL[-1]) * dl), hip.synth_fill(2 + 2 * k, n, 0, int(offR[-1]) * dl)) for k in range(3)]
dOL, dOR = hip.upload(offL), hip.upload(offR)
tot = int(np.sum(np.asarray(t1s, dtype=np.int64) * np.asarray(t2s, dtype=np.int64)))
out = hip.empty_words(tot * dl); off_out = hip.empty_words(len(t1s) + 1)
plan = hip.empty_words(int(hip.lib.csgn_mul_ragged_async_plan_words(len(t1s))))
torch.cuda.synchronize()
for i in range(60):
    l, r = sets[i % 3]
    hip.mul_ragged_async(n, l, dOL, r, dOR, tot, out=out, off_out=off_out, plan=plan)
torch.cuda.synchronize()
print("done")
