#!/usr/bin/env python3
"""Where a compaction group's time goes (dev tool): builds libcsgn_hip with -DCSGN_COMPACT_STAMPS into
/tmp, runs the 4096 x 1024-term case once warm and prints the mean time between the main kernel's phase stamps.

    python tools/prof_compact_phases.py [dup_fraction [batch terms [max_terms]]]

max_terms 0 = bound unknown: ciphertexts beyond one workgroup's group take the hash-partition path and the stamps are
those of their chunks.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = "/tmp/libcsgn_hip_stamps.so"
if "CSGN_HIP_LIB" not in os.environ:
    from csgn_amd import build
    srcs = [os.path.join(build.CSRC, s) for s in build.HIP_SOURCES]
    subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-Wno-pass-failed", "-DCSGN_COMPACT_STAMPS", "-I" + build.INCLUDE, "-I" + build.CSRC,
                           "-o", LIB] + srcs)
    sys.exit(subprocess.call([sys.executable] + sys.argv, env=dict(os.environ, CSGN_HIP_LIB=LIB)))

import numpy as np
import torch
from csgn_amd.batch import HipPath

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
hip = HipPath(0)
n, dl, batch, terms = 1247, 20, 4096, 1024
if len(sys.argv) > 3:
    batch, terms = int(sys.argv[2]), int(sys.argv[3])
bound = int(sys.argv[4]) if len(sys.argv) > 4 else terms
w = hip.synth_fill(11, n, 0, batch * terms * dl).view(batch, terms, dl)
distinct = max(1, int(round(terms * (1.0 - frac))))
if distinct < terms:
    src = torch.randint(0, distinct, (batch, terms - distinct), device=hip.device)
    w[:, distinct:, :] = torch.gather(w[:, :distinct, :], 1, src.unsqueeze(-1).expand(-1, -1, dl))
w = w.reshape(-1)
off = torch.arange(0, (batch + 1) * terms, terms, dtype=torch.int64, device=hip.device)
total = batch * terms
nbytes = int(hip.lib.csgn_compact_scratch_bytes(n, batch, total))
ngroups = max(1, (batch * terms + 511) // 512 + 8)                # more than any batch can have
extra = 16 * 8 * (ngroups * 2 + 64)
scratch = torch.zeros(nbytes + extra, dtype=torch.uint8, device=hip.device)
out, off_out = hip.empty_words(total * dl), hip.empty_words(batch + 1)
for _ in range(5):
    hip.compact_ragged(n, w, off, total_terms=total, max_terms=bound, out=out, off_out=off_out, scratch=scratch, sync=False)
torch.cuda.synchronize()
st = scratch[nbytes: nbytes + ngroups * 16 * 8].view(torch.int64).cpu().numpy().reshape(ngroups, 16)
st = st[st[:, 0] != 0]                                            # the groups that ran
batch = st.shape[0]
names = ["start", "setup", "loads", "hash", "insert", "verify", "keep", "scan", "look-back+next", "off_out", "stores issued"]
d = np.diff(st[:, :11], axis=1) / 100.0                      # 100 MHz ticks -> us
print(f"duplicates {frac:.2f}, {terms} terms per ciphertext: {batch} groups, kernel span {(st[:, 10].max() - st[:, 0].min()) / 100.0:.1f} us")
for i, nm in enumerate(names[1:]):
    print(f"  {nm:<14} mean {d[:, i].mean():7.2f} us   median {np.median(d[:, i]):7.2f}   max {d[:, i].max():7.2f}")
print(f"  group total    mean {(st[:, 10] - st[:, 0]).mean() / 100.0:7.2f} us")
