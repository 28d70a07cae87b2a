#!/usr/bin/env python3
"""What does the headline dispatch do when the chip is NOT idle?  (dev tool)

The bench shape (N=1247, 1024x1024 terms, streamed through a 128-slot arena) is multiplied on one HIP
stream while a SECOND stream runs back-to-back 1 GiB device-to-device copies -- a co-tenant that
streams through the same HBM and the same 256 MiB memory-side cache the operand touch pass relies on
(csgn_mul.hip: k_touch leaves <= 64 MB of operands there until their pairs run).  Both dispatches,
interleaved in ONE process (rule 24 of the guide):
    auto   = k_touch + k_mul_flat  (library default for this shape)
    tiled  = k_mul_tiled           (knob shared_gpu = 1: LDS-staged left tile, no cache dependence)
with and without the co-tenant.  Prints mult/s and algorithmic TB/s per arm and round, the
co-tenant's own copy rate, and the medians.

    python tools/bench_cotenant.py [--batch 8192] [--rounds 5] [--json out.json]
"""
import argparse, json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from csgn_amd.batch import HipPath
from csgn_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--slots", type=int, default=128)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--json", type=str, default="")
args = ap.parse_args()

hip = HipPath(0)
n, T, dl = 1247, 1024, 20
B, slots = args.batch, args.slots
opw, per = T * dl, T * T * dl
L = hip.synth_fill(1, n, 0, B * opw)
R = hip.synth_fill(2, n, 0, B * opw)
arena = hip.empty_words(slots * per)
src = torch.empty(1 << 27, dtype=torch.int64, device=hip.device).random_()      # 1 GiB
dst = torch.empty_like(src)
bytes_per_mul = 8 * dl * (2 * T + T * T)
main, side = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()


def run(arm, cotenant):
    capi.reset_tuning()
    if arm == "tiled":
        capi.set_tuning("shared_gpu", 1)                 # the documented knob for callers that share the GPU
    name = hip.lib.csgn_mul_uniform_kernel(n, B, T, T).decode()
    ncopies = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if cotenant:
        with torch.cuda.stream(side):
            c0.record()
            # enough copies queued to outlast the multiply (each ~0.7 ms alone, longer when sharing)
            ncopies = int(B * bytes_per_mul / 6e12 / 0.5e-3) + 50
            for _ in range(ncopies):
                dst.copy_(src, non_blocking=True)
            c1.record()
    with torch.cuda.stream(main):
        e0.record()
        hip.mul_uniform(n, B, T, T, L, R, out=arena, out_slots=slots)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    out = {"arm": arm, "kernel": name, "cotenant": cotenant, "ms": ms, "mult_per_s": B / (ms / 1e3),
           "TBps": B * bytes_per_mul / (ms / 1e3) / 1e12}
    if cotenant:
        cms = c0.elapsed_time(c1)
        out["copy_TBps_rw"] = ncopies * 2 * src.numel() * 8 / (cms / 1e3) / 1e12
        out["copies_outlasted_multiply"] = cms >= ms
    capi.reset_tuning()
    return out


rows = []
run("auto", False)                      # warm-up
for r in range(args.rounds):
    for cot in (False, True):
        for arm in ("auto", "tiled"):
            row = run(arm, cot)
            row["round"] = r
            rows.append(row)
            print(json.dumps(row), flush=True)
summary = {}
for cot in (False, True):
    for arm in ("auto", "tiled"):
        sel = [x for x in rows if x["arm"] == arm and x["cotenant"] == cot]
        summary[f"{arm}/{'cotenant' if cot else 'alone'}"] = {
            "kernel": sel[0]["kernel"], "median_TBps": statistics.median(x["TBps"] for x in sel),
            "min_TBps": min(x["TBps"] for x in sel), "max_TBps": max(x["TBps"] for x in sel),
            "median_mult_per_s": statistics.median(x["mult_per_s"] for x in sel),
            **({"median_copy_TBps_rw": statistics.median(x["copy_TBps_rw"] for x in sel),
                "copies_outlasted_multiply": all(x["copies_outlasted_multiply"] for x in sel)} if cot else {})}
print("SUMMARY " + json.dumps(summary, indent=1), flush=True)
if args.json:
    with open(args.json, "w") as f:
        json.dump({"config": {"n": n, "terms": T, "batch": B, "slots": slots, "copy_bytes": src.numel() * 8},
                   "rows": rows, "summary": summary}, f, indent=1)
