#!/usr/bin/env python3
"""Secondary kernels of the hot path vs the HBM roofline (dev tool; bench.py is the contract).

    python tools/bench_ops.py [--rounds 7] [--json out.json]

Algorithmic bytes per unit of work are SURVEY 8d's: mul 8*dL*(T1+T2+T1*T2), add 2*8*dL*(T1+T2),
decrypt 8*dL*T, encrypt 8*dL per ciphertext (device RNG: write only).
"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from csgn_amd.batch import HipPath

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--json", type=str, default="")
args = ap.parse_args()

hip = HipPath(0)
PEAK = 8.0e12
rows = []


def timed(fn, rounds=args.rounds):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) / 1e3)
    return statistics.median(ts), min(ts)


def report(name, units, unit_name, alg_bytes, fn):
    med, best = timed(fn)
    row = dict(op=name, units=units, unit=unit_name, median_ms=med * 1e3, best_ms=best * 1e3,
               rate=units / med, gbps=alg_bytes / med / 1e9, frac=alg_bytes / med / PEAK)
    rows.append(row)
    print(f"{name:<44} {med*1e3:9.3f} ms  {units/med:14.4g} {unit_name}/s  {alg_bytes/med/1e9:8.1f} GB/s  {100*alg_bytes/med/PEAK:5.1f}% of peak", flush=True)


for n, d in [(1247, 16), (4096, 32)]:
    dl = hip.default_len(n)
    key = np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
    dmask = hip.upload(hip.key_mask(n, key))
    dkey = hip.upload(key)
    # fresh 1x1 products (BASELINE configs 2 and 4)
    for batch in (65536, 1 << 20):
        L = hip.synth_fill(1, n, 0, batch * dl)
        R = hip.synth_fill(2, n, 0, batch * dl)
        out = hip.empty_words(batch * dl)
        report(f"mul 1x1 N={n} batch={batch}", batch, "mult", batch * 3 * 8 * dl,
               lambda: hip.mul_uniform(n, batch, 1, 1, L, R, out=out))
    # small shapes through the flat kernel
    for (t1, t2, batch) in [(2, 2, 1 << 18), (8, 8, 1 << 15), (32, 32, 4096)]:
        L = hip.synth_fill(1, n, 0, batch * t1 * dl)
        R = hip.synth_fill(2, n, 0, batch * t2 * dl)
        out = hip.empty_words(batch * t1 * t2 * dl)
        report(f"mul {t1}x{t2} N={n} batch={batch}", batch, "mult", batch * 8 * dl * (t1 + t2 + t1 * t2),
               lambda: hip.mul_uniform(n, batch, t1, t2, L, R, out=out))
    # mid/large shapes through the tiled kernel
    for (t1, t2, batch) in [(256, 256, 256), (1024, 64, 256), (64, 1024, 256), (1000, 1000, 16)] + ([(1024, 1024, 64)] if n == 1247 else [(512, 512, 32)]):
        L = hip.synth_fill(1, n, 0, batch * t1 * dl)
        R = hip.synth_fill(2, n, 0, batch * t2 * dl)
        out = hip.empty_words(batch * t1 * t2 * dl)
        report(f"mul {t1}x{t2} N={n} batch={batch}", batch, "mult", batch * 8 * dl * (t1 + t2 + t1 * t2),
               lambda: hip.mul_uniform(n, batch, t1, t2, L, R, out=out))
        del L, R, out
    if n == 1247:
        # BASELINE config 3's kernel by name: the LDS-tiled all-pairs kernel at 1024x1024 (the default dispatch
        # for this shape is touch + flat, timed by bench.py; knob mul_flat = -1 selects the tiled one)
        from csgn_amd import capi
        t1 = t2 = 1024
        batch = 64
        L = hip.synth_fill(1, n, 0, batch * t1 * dl)
        R = hip.synth_fill(2, n, 0, batch * t2 * dl)
        out = hip.empty_words(batch * t1 * t2 * dl)
        capi.set_tuning("mul_flat", -1)
        assert hip.lib.csgn_mul_uniform_kernel(n, batch, t1, t2).decode() == "k_mul_tiled"
        report(f"mul {t1}x{t2} N={n} batch={batch} [k_mul_tiled, LDS]", batch, "mult", batch * 8 * dl * (t1 + t2 + t1 * t2),
               lambda: hip.mul_uniform(n, batch, t1, t2, L, R, out=out))
        capi.reset_tuning()
        del L, R, out
    # add (concatenation)
    for (t1, t2, batch) in [(1, 1, 1 << 20), (1024, 1024, 1024)]:
        L = hip.synth_fill(1, n, 0, batch * t1 * dl)
        R = hip.synth_fill(2, n, 0, batch * t2 * dl)
        report(f"add {t1}+{t2} N={n} batch={batch}", batch, "add", batch * 2 * 8 * dl * (t1 + t2),
               lambda: hip.add_uniform(n, batch, t1, t2, L, R))
        del L, R
    # decrypt
    for (terms, batch) in [(1, 1 << 20), (1024, 4096), (1 << 20, 8)]:
        W = hip.synth_fill(3, n, 0, batch * terms * dl)
        report(f"decrypt T={terms} N={n} batch={batch}", batch * terms, "term", batch * terms * 8 * dl,
               lambda: hip.decrypt_uniform(n, batch, terms, W, dmask))
        del W
    # encrypt (keyed device generator: ChaCha8; VALU-issue-bound) and permutation, at 1 M and 4 M
    # ciphertexts (a 1 M launch is 50-80 us: launch ramp and the persistent workgroups' set-up are 10 %+ of it)
    for batch in (1 << 20, 1 << 22):
        plain = hip.upload(np.random.default_rng(2).integers(0, 2, batch).astype(np.uint8))
        fresh = hip.empty_words(batch * dl)
        rng = hip.rng_from_seed(7, 8)
        report(f"encrypt(keyed ChaCha8) N={n} batch={batch}", batch, "ct", batch * 8 * dl,
               lambda: hip.encrypt_keyed(n, d, plain, dkey, dmask, rng, out=fresh))
        perm = hip.upload(np.random.default_rng(3).permutation(n).astype(np.uint32))
        report(f"permute N={n} batch={batch}", batch, "ct", batch * 2 * 8 * dl,
               lambda: hip.permute_uniform(n, batch, 1, fresh, perm))
        del fresh, plain
        torch.cuda.empty_cache()

if args.json:
    with open(args.json, "w") as f:
        json.dump(rows, f, indent=1)
