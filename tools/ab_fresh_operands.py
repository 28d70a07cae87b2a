#!/usr/bin/env python3
"""Are operands that the previous kernel just WROTE warm or cold for the multiply?  (dev probe)
L and R are produced by csgn_add_uniform right before every timed multiply, into buffers that rotate through
more memory than the 256 MB memory-side cache; the multiply is timed alone under several dispatch choices.
    python tools/ab_fresh_operands.py [thin]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from csgn_amd.batch import HipPath
from csgn_amd import capi
from csgn_amd.capi import check
hip = HipPath(0)
SQUARE = [(8, 8, 32768), (16, 16, 16384), (32, 32, 4096), (64, 64, 1024), (8, 8, 131072)]
THIN = [(2, 2, 262144), (4, 4, 131072), (4, 16, 32768), (16, 4, 32768), (2, 64, 16384), (64, 2, 16384), (4, 64, 8192),
        (64, 4, 8192), (2, 1024, 1024), (1024, 2, 1024), (8, 32, 8192), (32, 8, 8192)]
VARIANTS = [("default", {}), ("touch0", {"mul_touch": 0}), ("touch3", {"mul_touch": 3}), ("tiled", {"mul_flat": -1}),
            ("flat2+touch", {"mul_flat": 2, "mul_touch": 3})]
TALL = [(2, 2, 262144), (4, 4, 131072), (16, 4, 32768), (64, 4, 8192), (64, 2, 16384), (1024, 2, 1024), (256, 8, 1024), (16, 2, 65536)]
if "tall" in sys.argv[1:]:
    VARIANTS = [("default", {}), ("flat1", {"mul_flat": 1, "mul_touch": 0}), ("flat2", {"mul_flat": 2, "mul_touch": 0}),
                ("flat4", {"mul_flat": 4, "mul_touch": 0}), ("flat8", {"mul_flat": 8, "mul_touch": 0}),
                ("flat1+touchL", {"mul_flat": 1, "mul_touch": 1}), ("flat2+touchL", {"mul_flat": 2, "mul_touch": 1}), ("default", {})]
MID = [(8, 8, 32768), (8, 8, 131072), (6, 6, 65536), (8, 16, 16384), (16, 8, 16384), (12, 12, 16384), (16, 16, 16384), (24, 24, 4096), (32, 32, 4096)]
if "mid" in sys.argv[1:]:
    VARIANTS = [("(warm-up)", {}), ("default", {}), ("flat2+touch", {"mul_flat": 2, "mul_touch": 3}), ("default", {}),
                ("flat2+touch", {"mul_flat": 2, "mul_touch": 3}), ("flat2", {"mul_flat": 2, "mul_touch": 0}), ("tiled", {"mul_flat": -1}),
                ("default", {}), ("flat2+touch", {"mul_flat": 2, "mul_touch": 3})]
if "bs" in sys.argv[1:]:
    VARIANTS = [("(warm-up)", {}), ("default", {}), ("tiled bs=256", {"mul_flat": -1}), ("tiled bs=128", {"mul_flat": -1, "mul_bs": 128}),
                ("tiled bs=64", {"mul_flat": -1, "mul_bs": 64}), ("tiled bs=128 ti=8", {"mul_flat": -1, "mul_bs": 128, "mul_ti": 8}), ("default", {})]
SHORT = [(4, 4, 131072), (8, 4, 65536), (16, 4, 32768), (64, 4, 8192), (8, 8, 32768), (16, 8, 16384), (32, 8, 8192), (64, 8, 4096),
         (256, 8, 1024), (10, 10, 16384), (12, 12, 16384), (64, 2, 16384), (1024, 2, 1024)]
if "short" in sys.argv[1:]:
    VARIANTS = [("(warm-up)", {}), ("default", {})] + [("bs%d ti%d" % (b, t), {"mul_flat": -1, "mul_bs": b, "mul_ti": t})
                                                        for b, t in ((128, 8), (128, 16), (128, 4), (64, 8), (64, 16))] + [("default", {})]
shapes = SHORT if "short" in sys.argv[1:] else MID if "bs" in sys.argv[1:] else THIN if "thin" in sys.argv[1:] else TALL if "tall" in sys.argv[1:] else MID if "mid" in sys.argv[1:] else SQUARE
for n in (1247, 4096):
    dl = hip.default_len(n)
    for t1, t2, batch in shapes:
        h1, h2 = t1 // 2, t2 // 2
        A1 = hip.synth_fill(1, n, 0, batch * h1 * dl); A2 = hip.synth_fill(2, n, 0, batch * h1 * dl)
        B1 = hip.synth_fill(3, n, 0, batch * h2 * dl); B2 = hip.synth_fill(4, n, 0, batch * h2 * dl)
        op_bytes = batch * (t1 + t2) * dl * 8
        nsets = max(3, int(600e6 // op_bytes) + 1)
        Ls = [hip.empty_words(batch * t1 * dl) for _ in range(nsets)]
        Rs = [hip.empty_words(batch * t2 * dl) for _ in range(nsets)]
        out = hip.empty_words(batch * t1 * t2 * dl)
        row = []
        for name, kn in VARIANTS:
            ts = []
            for it in range(10):
                k = it % nsets
                capi.reset_tuning()
                check(hip.lib.csgn_add_uniform(n, batch, h1, h1, A1.data_ptr(), A2.data_ptr(), Ls[k].data_ptr(), hip.stream))
                check(hip.lib.csgn_add_uniform(n, batch, h2, h2, B1.data_ptr(), B2.data_ptr(), Rs[k].data_ptr(), hip.stream))
                for kk, v in kn.items():
                    capi.set_tuning(kk, v)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); hip.mul_uniform(n, batch, t1, t2, Ls[k], Rs[k], out=out); b.record(); b.synchronize()
                if it >= 2:
                    ts.append(a.elapsed_time(b) / 1e3)
            tm = statistics.median(ts)
            kname = hip.lib.csgn_mul_uniform_kernel(n, batch, t1, t2).decode()
            row.append("%s %.0f%s" % (name, batch * 8 * dl * (t1 + t2 + t1 * t2) / tm / 1e9, " [" + kname + "]" if name == "default" and not any("[" in r for r in row) else ""))
        capi.reset_tuning()
        print(f"N={n} {t1}x{t2} x{batch} (operands {op_bytes/1e6:.0f} MB, {nsets} sets): " + " | ".join(row), flush=True)
        del Ls, Rs, out, A1, A2, B1, B2
