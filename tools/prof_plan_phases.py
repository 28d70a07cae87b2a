#!/usr/bin/env python3
"""Where k_plan's time goes (dev tool).  Needs libcsgn_hip.so built with -DCSGN_PLAN_STAMPS: every 4th workgroup
leaves 100 MHz wall-clock stamps {start, ticket, loaded, prefix known, offsets written, checksum added, counted done}
behind the status granules of the plan buffer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath
hip = HipPath(0)
n, dl = 1247, 20
batch = 1 << 20
off = hip.upload(np.arange(batch + 1, dtype=np.uint64))
L = hip.synth_fill(1, n, 0, batch * dl); R = hip.synth_fill(2, n, 0, batch * dl)
out = hip.empty_words(batch * dl); off_out = hip.empty_words(batch + 1)
plan = hip.empty_words(int(hip.lib.csgn_mul_ragged_async_plan_words(batch)))
for _ in range(4):
    hip.mul_ragged_async(n, L, off, R, off, batch, out=out, off_out=off_out, plan=plan)
torch.cuda.synchronize()
w = hip.download(plan)
nchunks = (batch + 4095) // 4096
head_words = int(os.environ.get("PLAN_HEAD_WORDS", "0")) or None
# the status granules start after the head; find them: granule k holds flag 2 in the top bits
start = int(os.environ.get("SCAN_AT", "0"))
print("scan block at word", start, "chunks", nchunks)
st = w[start + nchunks + 3: start + nchunks + 3 + ((nchunks + 3) // 4) * 8].reshape(-1, 8).astype(np.int64)
t0 = st[:, 0].min()
rel = (st[:, :6] - t0) / 100.0      # us
names = ["start", "ticket", "loaded", "prefix", "written", "counted"]
print("chunk  " + "  ".join(f"{x:>8}" for x in names))
for i in list(range(0, len(rel), max(1, len(rel) // 16))) + [len(rel) - 1]:
    print(f"{4*i:5d}  " + "  ".join(f"{x:8.2f}" for x in rel[i]))
print("max    " + "  ".join(f"{x:8.2f}" for x in rel.max(axis=0)))
print("mean d " + "  ".join(f"{x:8.2f}" for x in np.diff(rel, axis=1, prepend=rel[:, :1]).mean(axis=0)))
