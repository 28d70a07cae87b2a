#!/usr/bin/env python3
"""A few planned multiplies of one long-tailed batch of small pairs, for rocprofv3 --pmc (dev tool: is the CSR kernel
bound by its per-lane index arithmetic?).  MEAN=8 (default) or 16; CSGN_* knobs as usual."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from csgn_amd.batch import HipPath, check
hip = HipPath(0)
n, dl = 1247, 20
mean = int(os.environ.get("MEAN", "8"))
cnt = (1 << 18) if mean == 8 else (1 << 16)
rng = np.random.default_rng(0)
t1s = np.clip(rng.lognormal(np.log(mean) - 0.5, 1, cnt), 1, 600).astype(np.int64)
t2s = np.clip(rng.lognormal(np.log(mean) - 0.5, 1, cnt), 1, 600).astype(np.int64)
def csr(c):
    o = np.zeros(len(c) + 1, dtype=np.uint64); o[1:] = np.cumsum(np.asarray(c, dtype=np.uint64)); return o
offL, offR = csr(t1s), csr(t2s)
L = hip.synth_fill(1, n, 0, int(offL[-1]) * dl); R = hip.synth_fill(2, n, 0, int(offR[-1]) * dl)
dL_, dR_ = hip.upload(offL), hip.upload(offR)
out, off_out = hip.mul_ragged(n, L, dL_, R, dR_)
handle = hip.mul_plan(); hplan = (C.c_uint64 * 4)()
check(hip.lib.csgn_mul_plan_ragged(handle, len(t1s), dL_.data_ptr(), dR_.data_ptr(), off_out.data_ptr(), C.byref(hplan), hip.stream))
check(hip.lib.csgn_mul_plan_trust(handle, 1))
for _ in range(int(os.environ.get("CALLS", "3"))):
    check(hip.lib.csgn_mul_planned(handle, n, L.data_ptr(), R.data_ptr(), out.data_ptr(), hip.stream))
torch.cuda.synchronize()
print("out terms", int(hplan[0]), "bytes", int(hplan[0]) * dl * 8, flush=True)
