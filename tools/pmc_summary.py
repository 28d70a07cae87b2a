#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into profiles/<tag>_summary.json.

  python tools/pmc_summary.py gpurun_out/prof_r01 profiles/r01 [kernel-substring]

Reads <dir>/trace/*kernel_stats.csv (per-kernel average duration) and
<dir>/pmc_{WRITE_SIZE,FETCH_SIZE}/*counter_collection.csv, and reports per-launch HBM
traffic of the dominant kernel with the gfx950 corrections of MI355X_MICROARCH.md "HBM":
counters are in KiB; FETCH_SIZE under-reports wide coalesced reads by exactly 2x; WRITE_SIZE
is exact for 16-B-per-lane streaming stores.
"""
import csv
import glob
import json
import os
import sys


def read_counter(dirname, counter, kernel_sub):
    vals = []
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter and kernel_sub in row.get("Kernel_Name", ""):
                    vals.append(float(row["Counter_Value"]))
    return vals


def main():
    src, dst = sys.argv[1], sys.argv[2]
    kernel_sub = sys.argv[3] if len(sys.argv) > 3 else "k_mul_tiled"
    out = {"kernel": kernel_sub}
    for path in glob.glob(os.path.join(src, "trace", "*kernel_stats.csv")):
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel_sub in row["Name"]:
                    out["calls"] = int(row["Calls"])
                    out["avg_duration_ms"] = float(row["AverageNs"]) / 1e6
                    out["min_duration_ms"] = float(row["MinNs"]) / 1e6
                    out["max_duration_ms"] = float(row["MaxNs"]) / 1e6
    w = read_counter(os.path.join(src, "pmc_WRITE_SIZE"), "WRITE_SIZE", kernel_sub)
    r = read_counter(os.path.join(src, "pmc_FETCH_SIZE"), "FETCH_SIZE", kernel_sub)
    if w:
        out["write_size_kib_per_launch"] = sum(w) / len(w)
        out["hbm_write_bytes_per_launch"] = sum(w) / len(w) * 1024
    if r:
        out["fetch_size_kib_per_launch_raw"] = sum(r) / len(r)
        out["hbm_read_bytes_per_launch"] = sum(r) / len(r) * 1024 * 2      # gfx950 x2 correction
    if w and r:
        out["hbm_bytes_per_launch"] = out["hbm_write_bytes_per_launch"] + out["hbm_read_bytes_per_launch"]
        out["pmc_launches_sampled"] = [len(w), len(r)]
    for name in ("bench_trace.json", "pmc_WRITE_SIZE.json"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
            if lines:
                out.setdefault("bench_lines", {})[name] = json.loads(lines[-1])
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    with open(dst + "_summary.json", "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps({k: v for k, v in out.items() if k != "bench_lines"}, indent=1))


if __name__ == "__main__":
    main()
