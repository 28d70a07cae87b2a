#!/usr/bin/env python3
"""Secondary kernels of the hot path vs the HBM roofline (dev tool; bench.py is the contract).

    python tools/bench_ops.py [--rounds 7] [--json out.json]

Times are STEADY-STATE (see timed(): >= 30 ms of warm-up, runs of back-to-back calls between one pair of events), with
the first call after set-up beside them; every call's INPUTS rotate through >= 600 MB of independent sets so that no
call finds its operands in the 256 MB memory-side cache from its own previous run.  Algorithmic bytes per unit of work are SURVEY 8d's: mul 8*dL*(T1+T2+T1*T2), add 2*8*dL*(T1+T2),
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
from csgn_amd.capi import check

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--json", type=str, default="")
args = ap.parse_args()

hip = HipPath(0)
PEAK = 8.0e12
rows = []


def timed(fn, rounds=args.rounds):
    """Steady-state time per call: the shader clock of this pool takes milliseconds to come back up after an
    idle gap (an allocation, an upload, a host synchronisation) -- the first launches after one read up to 25 %
    slow (profiles/r03/ab_enc_wg.log: 3.4 TB/s, then 4.6-4.7 for the same configuration) -- so every timing
    warms up for >= 30 ms of back-to-back calls and then brackets runs of K calls (>= 2 ms each) with one
    pair of events.  Returns (median, best, first): `first` is the very first call, straight after set-up."""
    def bracket(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(k):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / 1e3 / k
    first = bracket(1)
    spent, est = first, first
    while spent < 30e-3:
        k = max(1, min(64, int(5e-3 / max(est, 1e-6))))
        est = bracket(k)
        spent += est * k
    k = max(1, min(64, int(2e-3 / max(est, 1e-6)) + 1))
    ts = [bracket(k) for _ in range(rounds)]
    return statistics.median(ts), min(ts), first


def report(name, units, unit_name, alg_bytes, in_bytes, make_inputs, run):
    """make_inputs(k) -> one independent set of input tensors; run(inputs) -> one call.  The call is timed on
    input sets taken in turn, as many as it takes to rotate through >= 600 MB (the memory-side cache holds
    256 MB), so a repeated call never finds its operands cached from its own previous run."""
    nsets = max(1, min(64, -(-int(600e6) // max(1, in_bytes)))) if in_bytes < 600e6 else 1
    sets = [make_inputs(k) for k in range(nsets)]
    turn = [0]

    def fn():
        run(sets[turn[0] % nsets])
        turn[0] += 1
    med, best, first = timed(fn)
    row = dict(op=name, units=units, unit=unit_name, median_ms=med * 1e3, best_ms=best * 1e3, first_call_ms=first * 1e3,
               input_sets=nsets, rate=units / med, gbps=alg_bytes / med / 1e9, frac=alg_bytes / med / PEAK)
    rows.append(row)
    print(f"{name:<44} {med*1e3:9.3f} ms  {units/med:14.4g} {unit_name}/s  {alg_bytes/med/1e9:8.1f} GB/s  {100*alg_bytes/med/PEAK:5.1f}% of peak"
          f"   ({nsets} input set{'s' if nsets > 1 else ''}; first call after set-up {alg_bytes/first/1e9:7.1f} GB/s)", flush=True)
    del sets


def mul_case(n, dl, t1, t2, batch, tag=""):
    out = hip.empty_words(batch * t1 * t2 * dl)
    report(f"mul {t1}x{t2} N={n} batch={batch}{tag}", batch, "mult", batch * 8 * dl * (t1 + t2 + t1 * t2),
           batch * 8 * dl * (t1 + t2),
           lambda k: (hip.synth_fill(1 + 2 * k, n, 0, batch * t1 * dl), hip.synth_fill(2 + 2 * k, n, 0, batch * t2 * dl)),
           lambda lr: hip.mul_uniform(n, batch, t1, t2, lr[0], lr[1], out=out))
    del out


for n, d in [(1247, 16), (4096, 32)]:
    dl = hip.default_len(n)
    key = np.random.default_rng(1).permutation(n)[:d].astype(np.uint64)
    dmask = hip.upload(hip.key_mask(n, key))
    dkey = hip.upload(key)
    # fresh 1x1 products (BASELINE configs 2 and 4)
    for batch in (65536, 1 << 20):
        mul_case(n, dl, 1, 1, batch)
    # small shapes through the flat kernel
    for (t1, t2, batch) in [(2, 2, 1 << 18), (8, 8, 1 << 15), (32, 32, 4096)]:
        mul_case(n, dl, t1, t2, batch)
    # mid/large shapes
    for (t1, t2, batch) in [(256, 256, 256), (1024, 64, 256), (64, 1024, 256), (1000, 1000, 16)] + ([(1024, 1024, 64)] if n == 1247 else [(512, 512, 32)]):
        mul_case(n, dl, t1, t2, batch)
    if n == 1247:
        # BASELINE config 3's kernel by name: the LDS-tiled all-pairs kernel at 1024x1024 (the default dispatch
        # for this shape is touch + flat, timed by bench.py; knob mul_flat = -1 selects the tiled one)
        from csgn_amd import capi
        capi.set_tuning("mul_flat", -1)
        assert hip.lib.csgn_mul_uniform_kernel(n, 64, 1024, 1024).decode() == "k_mul_tiled"
        mul_case(n, dl, 1024, 1024, 64, " [k_mul_tiled, LDS]")
        capi.reset_tuning()
    # add (concatenation)
    for (t1, t2, batch) in [(1, 1, 1 << 20), (1024, 1024, 1024)]:
        out = hip.empty_words(batch * (t1 + t2) * dl)
        report(f"add {t1}+{t2} N={n} batch={batch}", batch, "add", batch * 2 * 8 * dl * (t1 + t2), batch * 8 * dl * (t1 + t2),
               lambda k: (hip.synth_fill(1 + 2 * k, n, 0, batch * t1 * dl), hip.synth_fill(2 + 2 * k, n, 0, batch * t2 * dl)),
               lambda lr: check(hip.lib.csgn_add_uniform(n, batch, t1, t2, lr[0].data_ptr(), lr[1].data_ptr(), out.data_ptr(), hip.stream)))
        del out
    # decrypt
    for (terms, batch) in [(1, 1 << 20), (1024, 4096), (1 << 20, 8)]:
        bits = torch.empty(batch, dtype=torch.uint8, device=hip.device)
        scratch = torch.empty(int(hip.lib.csgn_decrypt_scratch_bytes(batch, batch * terms)), dtype=torch.uint8, device=hip.device)
        report(f"decrypt T={terms} N={n} batch={batch}", batch * terms, "term", batch * terms * 8 * dl, batch * terms * 8 * dl,
               lambda k: hip.synth_fill(3 + k, n, 0, batch * terms * dl),
               lambda W: check(hip.lib.csgn_decrypt_uniform(n, batch, terms, W.data_ptr(), dmask.data_ptr(), bits.data_ptr(),
                                                            scratch.data_ptr(), hip.stream)))
        del bits, scratch
    # encrypt (keyed device generator: ChaCha8; VALU-issue-bound) and permutation, at 1 M and 4 M
    # ciphertexts (a 1 M launch is 50-80 us: launch ramp and the persistent workgroups' set-up are 10 %+ of it)
    for batch in (1 << 20, 1 << 22):
        plain = hip.upload(np.random.default_rng(2).integers(0, 2, batch).astype(np.uint8))
        fresh = hip.empty_words(batch * dl)
        rng = hip.rng_from_seed(7, 8)
        report(f"encrypt(keyed ChaCha8) N={n} batch={batch}", batch, "ct", batch * 8 * dl, int(600e6),
               lambda k: None, lambda _unused: hip.encrypt_keyed(n, d, plain, dkey, dmask, rng, out=fresh))
        perm = hip.upload(np.random.default_rng(3).permutation(n).astype(np.uint32))
        pout = hip.empty_words(batch * dl)
        report(f"permute N={n} batch={batch}", batch, "ct", batch * 2 * 8 * dl, batch * 8 * dl,
               lambda k: hip.synth_fill(30 + k, n, 0, batch * dl),
               lambda W: check(hip.lib.csgn_permute_uniform(n, batch, 1, 0, W.data_ptr(), perm.data_ptr(), pout.data_ptr(), hip.stream)))
        del fresh, plain, pout
        torch.cuda.empty_cache()

if args.json:
    with open(args.json, "w") as f:
        json.dump(rows, f, indent=1)
