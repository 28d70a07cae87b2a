#!/usr/bin/env python3
"""Interleaved A/B sweep of the all-pairs kernel's tunables on one GPU (dev tool).

    python tools/sweep_mul.py [--pairs 256] [--slots 64] [--rounds 5]

Prints, per variant, median/min launch time and algorithmic GB/s; plus hipMemset / torch
fill_ over the same arena as write-only reference points.
"""
import argparse
import os
import sys
import statistics

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csgn_amd.batch import HipPath
from csgn_amd.capi import check, set_tuning

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=256)
ap.add_argument("--slots", type=int, default=64)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--terms", type=int, default=1024)
ap.add_argument("--nbits", type=int, default=1247)
ap.add_argument("--variants", type=str, default="")
args = ap.parse_args()

hip = HipPath(0)
n, T = args.nbits, args.terms
dl = hip.default_len(n)
L = hip.synth_fill(1, n, 0, args.pairs * T * dl)
R = hip.synth_fill(2, n, 0, args.pairs * T * dl)
arena = hip.empty_words(args.slots * T * T * dl)
bytes_per_mul = 8 * dl * (2 * T + T * T)
arena_bytes = arena.numel() * 8

if args.variants:
    variants = [tuple(int(x) for x in v.split(",")) for v in args.variants.split(";")]
else:
    variants = [(m, ti, nt) for nt in (0, 1) for m in (1, 2, 4, 8) for ti in (16, 64, 256)]


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / 1e3


def run_mul(m, ti, nt, flat=0, bs=0, xcd=1, pfkb=-1):
    for k, v in (("mul_pf_kb", pfkb), ("mul_xcd", xcd), ("mul_m", m), ("mul_ti", ti), ("mul_nt", nt),
                 ("mul_flat", flat), ("mul_bs", bs)):
        set_tuning(k, v)
    hip.mul_uniform(n, args.pairs, T, T, L, R, out=arena, out_slots=args.slots)


res = {v: [] for v in variants}
ref = {"hipMemset": [], "torch.fill_": []}
for v in variants:          # warm
    run_mul(*v)
torch.cuda.synchronize()
for r in range(args.rounds):
    ref["hipMemset"].append(timed(lambda: check(hip.lib.csgn_memset(arena.data_ptr(), 0x5A, arena_bytes, hip.stream))))
    ref["torch.fill_"].append(timed(lambda: arena.fill_(r)))
    for v in variants:
        res[v].append(timed(lambda: run_mul(*v)))

for k, ts in ref.items():
    print(f"{k:>14}: median {arena_bytes / statistics.median(ts) / 1e9:8.1f} GB/s  best {arena_bytes / min(ts) / 1e9:8.1f} GB/s")
print(f"{'M,TI,NT,FLAT,BS,XCD':>14}  median GB/s   best GB/s   mult/s(median)")
for v, ts in sorted(res.items(), key=lambda kv: statistics.median(kv[1])):
    med, best = statistics.median(ts), min(ts)
    print(f"{str(v):>14}  {args.pairs * bytes_per_mul / med / 1e9:10.1f}  {args.pairs * bytes_per_mul / best / 1e9:10.1f}   {args.pairs / med:10.0f}")
