#!/usr/bin/env python3
"""Mod-2 compaction (csgn_compact_ragged) vs the HBM roofline (dev tool; bench.py is the contract).

    python tools/bench_compact.py [--rounds 7] [--json out.json] [--only SUBSTR]

Algorithmic bytes = 8*dL*(T_in + T_out): every term read once, every surviving term written once.  Times are
steady state (>= 30 ms of warm-up, runs of back-to-back calls between one pair of events); every input is larger
than the 256 MB memory-side cache or rotates through sets that are, so no call finds its terms cached.
Shapes: 4096 ciphertexts of 1024 terms (BASELINE config 3's operands) with 0 / 50 / 95 % duplicate terms; the square
of a 32-term sum (1024 product terms of which 32 survive: a_i a_j = a_j a_i cancel, a_i a_i = a_i stay); the literal
(a+b)^2 (4 terms -> 2) and single-term ciphertexts by the million; 2^20-term ciphertexts (the hash-partition path);
N=4096.
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
ap.add_argument("--only", type=str, default="")
ap.add_argument("--once", action="store_true", help="one call per case, no timing loop (for rocprofv3 --pmc)")
args = ap.parse_args()

hip = HipPath(0)
PEAK = 8.0e12
rows = []


def timed(fn, rounds=args.rounds):
    def bracket(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(k):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / 1e3 / k
    first = bracket(1)
    if args.once:
        return first, first, first
    spent, est = first, first
    while spent < 30e-3:
        k = max(1, min(64, int(5e-3 / max(est, 1e-6))))
        est = bracket(k)
        spent += est * k
    k = max(1, min(64, int(2e-3 / max(est, 1e-6)) + 1))
    ts = [bracket(k) for _ in range(rounds)]
    return statistics.median(ts), min(ts), first


def with_duplicates(n, dl, batch, terms, frac, seed):
    """batch ciphertexts of `terms` terms: the first (1-frac)*terms are distinct random terms, the rest are
    copies of randomly chosen ones among them (built on the device)."""
    w = hip.synth_fill(seed, n, 0, batch * terms * dl).view(batch, terms, dl)
    distinct = max(1, int(round(terms * (1.0 - frac))))
    if distinct < terms:
        g = torch.Generator(device=hip.device)
        g.manual_seed(seed)
        src = torch.randint(0, distinct, (batch, terms - distinct), device=hip.device, generator=g)
        w[:, distinct:, :] = torch.gather(w[:, :distinct, :], 1, src.unsqueeze(-1).expand(-1, -1, dl))
    return w.reshape(-1)


def case(name, n, counts_or_uniform, make_sets, max_terms=0):
    """make_sets() -> list of flat term tensors (all with the same CSR offsets)."""
    if args.only and args.only not in name:
        return
    dl = hip.default_len(n)
    if isinstance(counts_or_uniform, tuple):
        batch, terms = counts_or_uniform
        off = torch.arange(0, (batch + 1) * terms, terms, dtype=torch.int64, device=hip.device)
        total = batch * terms
    else:
        counts = np.asarray(counts_or_uniform, dtype=np.uint64)
        host_off = np.zeros(counts.size + 1, dtype=np.uint64)
        host_off[1:] = np.cumsum(counts)
        off = hip.upload(host_off)
        batch, total = counts.size, int(host_off[-1])
    sets = make_sets()
    out = hip.empty_words(total * dl)
    off_out = hip.empty_words(batch + 1)
    scratch = torch.empty(int(hip.lib.csgn_compact_scratch_bytes(n, batch, total)), dtype=torch.uint8, device=hip.device)
    turn = [0]

    def fn():
        hip.compact_ragged(n, sets[turn[0] % len(sets)], off, total_terms=total, max_terms=max_terms, out=out,
                           off_out=off_out, scratch=scratch, sync=False)
        turn[0] += 1
    fn()
    kept = int(hip.download(off_out[-1:])[0])
    alg = 8 * dl * (total + kept)
    med, best, first = timed(fn)
    row = dict(op=name, n=n, batch=batch, terms_in=total, terms_out=kept, median_ms=med * 1e3, best_ms=best * 1e3,
               first_call_ms=first * 1e3, input_sets=len(sets), gbps=alg / med / 1e9, frac=alg / med / PEAK,
               alg_bytes=alg, scratch_bytes=scratch.numel())
    rows.append(row)
    print(f"{name:<52} {med*1e3:9.3f} ms  in {total:>9} out {kept:>9} terms  {alg/med/1e9:8.1f} GB/s  "
          f"{100*alg/med/PEAK:5.1f}% of peak   ({len(sets)} set{'s' if len(sets) > 1 else ''}; first call {alg/first/1e9:7.1f} GB/s)",
          flush=True)
    del sets, out, off_out, scratch
    torch.cuda.empty_cache()


n, dl = 1247, 20
for frac in (0.0, 0.5, 0.95):
    case(f"compact 4096 x 1024 terms, {int(frac*100)}% duplicates N={n}", n, (4096, 1024),
         lambda frac=frac: [with_duplicates(n, dl, 4096, 1024, frac, 11)], max_terms=1024)
case(f"compact 4096 x 1024 terms, 0% duplicates, bound unknown N={n}", n, (4096, 1024),
     lambda: [with_duplicates(n, dl, 4096, 1024, 0.0, 11)])


def squares():
    s = hip.synth_fill(21, n, 0, 4096 * 32 * dl)
    return [hip.mul_uniform(n, 4096, 32, 32, s, s)]


case(f"compact (a1+..+a32)^2: 4096 x 1024 -> 32 terms N={n}", n, (4096, 1024), squares, max_terms=1024)


def literal_squares():
    s = hip.synth_fill(22, n, 0, (1 << 20) * 2 * dl)
    return [hip.mul_uniform(n, 1 << 20, 2, 2, s, s)]


case(f"compact (a+b)^2: 2^20 x 4 -> 2 terms N={n}", n, (1 << 20, 4), literal_squares, max_terms=4)
case(f"compact 2^22 x 1 term (nothing to merge) N={n}", n, (1 << 22, 1),
     lambda: [hip.synth_fill(23, n, 0, (1 << 22) * dl)], max_terms=1)
rng = np.random.default_rng(4)
ragged = np.minimum(np.maximum(rng.lognormal(np.log(48.0), 1.0, size=1 << 16), 1), 1024).astype(np.uint64)
case(f"compact ragged log-normal (mean {ragged.mean():.0f} terms) x 65536, 0% duplicates N={n}", n, ragged,
     lambda: [hip.synth_fill(24, n, 0, int(ragged.sum()) * dl)], max_terms=1024)
for frac in (0.0, 0.5):
    case(f"compact 4 x 2^20 terms, {int(frac*100)}% duplicates (hash partitions) N={n}", n, (4, 1 << 20),
         lambda frac=frac: [with_duplicates(n, dl, 4, 1 << 20, frac, 12)])
case(f"compact 256 x 16384 terms, 50% duplicates (hash partitions) N={n}", n, (256, 16384),
     lambda: [with_duplicates(n, dl, 256, 16384, 0.5, 13)])
n, dl = 4096, 64
for frac in (0.0, 0.5):
    case(f"compact 8192 x 256 terms, {int(frac*100)}% duplicates N={n}", n, (8192, 256),
         lambda frac=frac: [with_duplicates(n, dl, 8192, 256, frac, 14)], max_terms=256)
case(f"compact 2048 x 766 terms (config 5's end size), 0% duplicates, wide groups N={n}", n, (2048, 766),
     lambda: [with_duplicates(n, dl, 2048, 766, 0.0, 15)], max_terms=766)
case(f"compact 2048 x 766 terms (config 5's end size), 0% duplicates, bound unknown (hash partitions) N={n}", n, (2048, 766),
     lambda: [with_duplicates(n, dl, 2048, 766, 0.0, 15)])
n, dl = 1247, 20
case(f"compact 2048 x 1792 terms, 0% duplicates, wide groups N={n}", n, (2048, 1792),
     lambda: [with_duplicates(n, dl, 2048, 1792, 0.0, 16)], max_terms=1792)

if args.json:
    with open(args.json, "w") as f:
        json.dump(rows, f, indent=1)
