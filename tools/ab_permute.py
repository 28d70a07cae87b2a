#!/usr/bin/env python3
"""Time csgn_permute_uniform in its kernel forms (bit-plane with 16-/8-byte staging, ballot), GB/s
of input+output, in ONE process (dev tool)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath
from csgn_amd import capi

hip = HipPath(0)


def timed(fn, rounds=9):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) / 1e3)
    return statistics.median(ts)


for n, batch in [(1247, 1 << 20), (1247, 1 << 22), (1247, 1 << 16), (4096, 1 << 18), (4096, 1 << 20), (2048, 1 << 20), (130, 1 << 20), (1300, 1 << 20), (8192, 1 << 18), (10000, 1 << 16)]:
    dl = hip.default_len(n)
    W = hip.synth_fill(3, n, 0, batch * dl)
    perm = hip.upload(np.random.default_rng(3).permutation(n).astype(np.uint32))
    row = []
    for form in ("planes3", "planes3/p4", "planes3/p6", "planes3/p8", "planes3/narrow", "ballot") * 2:
        capi.reset_tuning()
        capi.set_tuning("perm_ballot", form == "ballot")
        capi.set_tuning("perm_narrow", form.endswith("narrow"))
        for part in form.split("/")[1:]:
            if part[0] == "w":
                capi.set_tuning("perm_waves", int(part[1:]))
            if part[0] == "p":
                capi.set_tuning("perm_persist", int(part[1:]))
        t = timed(lambda: hip.permute_uniform(n, batch, 1, W, perm))
        row.append("%s %.0f" % (form, batch * 2 * 8 * dl / t / 1e9))
    capi.reset_tuning()
    print(f"N={n} batch={batch}: " + " | ".join(row), flush=True)
    del W
