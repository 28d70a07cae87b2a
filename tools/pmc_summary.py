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


def read_counter(dirname, counter, kernel_subs):
    """Counter values of every dispatch whose kernel name contains one of kernel_subs, keyed by
    the substring that matched."""
    vals = {k: [] for k in kernel_subs}
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                for k in kernel_subs:
                    if k in row.get("Kernel_Name", ""):
                        vals[k].append(float(row["Counter_Value"]))
                        break
    return vals


def per_launch(vals, main):
    """Sum over all listed kernels divided by the number of dispatches of the main kernel: helper
    kernels (the operand touch in front of every multiply launch) are charged to the launch they serve."""
    n = len(vals.get(main, []))
    return (sum(sum(v) for v in vals.values()) / n, n) if n else (None, 0)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    kernel_arg = sys.argv[3] if len(sys.argv) > 3 else "k_mul_tiled"
    kernel_subs = kernel_arg.split("+")          # e.g. "k_touch+k_mul_flat": the last one is the main kernel
    main_k = kernel_subs[-1]
    out = {"kernel": kernel_arg}
    for path in glob.glob(os.path.join(src, "trace", "*kernel_stats.csv")):
        with open(path) as f:
            for row in csv.DictReader(f):
                for k in kernel_subs:
                    if k in row["Name"]:
                        st = out.setdefault("stats", {}).setdefault(k, {})
                        st["calls"] = int(row["Calls"])
                        st["avg_duration_ms"] = float(row["AverageNs"]) / 1e6
                        st["min_duration_ms"] = float(row["MinNs"]) / 1e6
                        st["max_duration_ms"] = float(row["MaxNs"]) / 1e6
                        st["total_ms"] = float(row["TotalDurationNs"]) / 1e6 if "TotalDurationNs" in row else st["calls"] * st["avg_duration_ms"]
                        break
    if "stats" in out and main_k in out["stats"]:
        m = out["stats"][main_k]
        out["calls"] = m["calls"]
        out["avg_duration_ms"] = m["avg_duration_ms"]                                    # main kernel alone
        out["avg_duration_ms_with_helpers"] = sum(v["total_ms"] for v in out["stats"].values()) / m["calls"]
    w = read_counter(os.path.join(src, "pmc_WRITE_SIZE"), "WRITE_SIZE", kernel_subs)
    r = read_counter(os.path.join(src, "pmc_FETCH_SIZE"), "FETCH_SIZE", kernel_subs)
    wv, wn = per_launch(w, main_k)
    rv, rn = per_launch(r, main_k)
    if wn:
        out["write_size_kib_per_launch"] = wv
        out["hbm_write_bytes_per_launch"] = wv * 1024
    if rn:
        out["fetch_size_kib_per_launch_raw"] = rv
        out["hbm_read_bytes_per_launch"] = rv * 1024 * 2      # gfx950 x2 correction
        out["fetch_size_kib_raw_by_kernel"] = {k: (sum(v) / rn) for k, v in r.items()}
    if wn and rn:
        out["hbm_bytes_per_launch"] = out["hbm_write_bytes_per_launch"] + out["hbm_read_bytes_per_launch"]
        out["pmc_launches_sampled"] = [wn, rn]
    for name in ("bench_trace.json", "pmc_WRITE_SIZE.json"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
            if lines:
                out.setdefault("bench_lines", {})[name] = json.loads(lines[-1])
    # profiles/traffic_current.json: what bench.py reports as roofline.traffic -- tied to the launch
    # shape and to the kernel source it was measured on (bench.py drops it when either differs)
    if "--traffic-json" in sys.argv and wn and rn:
        line = None
        for name in ("pmc_WRITE_SIZE.json", "bench_trace.json"):
            line = out.get("bench_lines", {}).get(name) or line
        if line:
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            import bench
            cfg = line["config"]
            traffic = {
                "kernel": kernel_arg, "n_bits": cfg["n_bits"], "terms": cfg["terms"],
                "pairs_per_launch": cfg["pairs_per_launch"],
                "hbm_bytes_per_launch": out["hbm_bytes_per_launch"],
                "hbm_write_bytes_per_launch": out["hbm_write_bytes_per_launch"],
                "hbm_read_bytes_per_launch": out["hbm_read_bytes_per_launch"],
                "read_bytes_by_kernel": {k: v * 1024 * 2 for k, v in out["fetch_size_kib_raw_by_kernel"].items()},
                "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"],
                "kernel_source_sha16": bench.kernel_source_hash(),
                "captured": sys.argv[sys.argv.index("--traffic-json") + 2] if len(sys.argv) > sys.argv.index("--traffic-json") + 2 else dst,
                "batch_per_gpu": cfg["batch_per_gpu"], "pmc_launches_sampled": [wn, rn],
                "method": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE in separate passes, program directly after -- "
                          "(python3 bench.py); KiB units; FETCH_SIZE x2 (gfx950 wide-read correction); WRITE_SIZE exact. "
                          "Reads = operands once by k_touch (from HBM) + once by k_mul_flat (L2 fills served by the "
                          "memory-side cache the touch filled; the L2-side counter cannot tell them from HBM reads, "
                          "so both are counted)",
            }
            with open(sys.argv[sys.argv.index("--traffic-json") + 1], "w") as f:
                json.dump(traffic, f, indent=1)
                f.write("\n")
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    with open(dst + "_summary.json", "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps({k: v for k, v in out.items() if k != "bench_lines"}, indent=1))


if __name__ == "__main__":
    main()
