#!/usr/bin/env python3
"""All-pairs multiply with COLD operands (every launch reads pairs it has not seen for > 1 GB of
operand traffic, outputs stream through an arena), per shape and per kernel setting (dev tool).

    python tools/bench_cold.py [--variants "FLAT=-1;FLAT=1,TOUCH=1,PF_KB=0"]
"""
import argparse, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csgn_amd.batch import HipPath
from csgn_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="FLAT=-1;FLAT=-1,TOUCH=1;FLAT=1,PF_KB=0;FLAT=1,TOUCH=1,PF_KB=0;FLAT=1,TOUCH=3,PF_KB=0")
ap.add_argument("--shapes", default="1247:1024x1024;1247:256x256;1247:64x64;1247:32x32;1247:16x16;1247:1024x8;1247:8x1024;1247:2x383;4096:256x256;4096:383x2;4096:64x64;1300:128x128")
ap.add_argument("--launch-mb", type=int, default=2048)
args = ap.parse_args()
hip = HipPath(0)
KNOBS = ("FLAT", "TOUCH", "PF_KB", "XCD", "M", "TI")


def setenv(variant):
    capi.reset_tuning()
    for kv in variant.split(","):
        if kv:
            k, v = kv.split("=")
            capi.set_tuning("mul_" + k, v)


for shape in args.shapes.split(";"):
    n, tt = shape.split(":")
    n = int(n); t1, t2 = (int(x) for x in tt.split("x"))
    dl = hip.default_len(n)
    out_b = 8 * dl * t1 * t2
    op_b = 8 * dl * (t1 + t2)
    per_launch = max(1, min(args.launch_mb * (1 << 20) // out_b, 1 << 20))      # pairs per launch
    launches = max(4, min(64, (3 << 30) // max(1, per_launch * op_b)))          # distinct operand sets
    while launches * per_launch * op_b > (24 << 30) and launches > 4:
        launches //= 2
    L = hip.synth_fill(1, n, 0, launches * per_launch * t1 * dl)
    R = hip.synth_fill(2, n, 0, launches * per_launch * t2 * dl)
    arena = hip.empty_words(per_launch * t1 * t2 * dl * 2)
    alg = per_launch * (out_b + op_b)
    row = []
    for variant in args.variants.split(";"):
        setenv(variant)
        ts = []
        for rep in range(2):
            for k in range(launches):
                Lk = L[k * per_launch * t1 * dl:(k + 1) * per_launch * t1 * dl]
                Rk = R[k * per_launch * t2 * dl:(k + 1) * per_launch * t2 * dl]
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                hip.mul_uniform(n, per_launch, t1, t2, Lk, Rk, out=arena[(k % 2) * per_launch * t1 * t2 * dl:], out_slots=per_launch)
                b.record(); b.synchronize()
                if rep:
                    ts.append(a.elapsed_time(b) / 1e3)
        row.append("%6.0f" % (alg / statistics.median(ts) / 1e9))
    print(f"N={n} {t1}x{t2} pairs/launch={per_launch} sets={launches} ops={launches*per_launch*op_b/2**30:.1f}GiB: " + " | ".join(row), flush=True)
    del L, R, arena
    torch.cuda.empty_cache()
print("variants:", args.variants)
