#!/usr/bin/env python3
"""clock_summary.py <counter_collection.csv> <out.json>: per dispatch of the issue-bound kernels,
effective shader clock = GRBM_GUI_ACTIVE / 8 XCDs / duration, wave-instructions per second, and
which phase of tools/prof_clock.py it ran in (the k_digest marker dispatches separate the phases:
word counts 1 = permute after idle, 2 = permute after multiplies, 3 = encrypt after multiplies,
4 = encrypt after idle)."""
import csv, json, re, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
by = {}
for r in rows:
    k = int(r["Dispatch_Id"])
    e = by.setdefault(k, {"name": r["Kernel_Name"], "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"]),
                          "grid": int(r["Grid_Size"])})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
out, phase, group = [], None, 0
PH = {1: "permute/idle", 2: "permute/after_multiplies", 3: "encrypt/after_multiplies", 4: "encrypt/idle"}
seen_marks = 0
for k in sorted(by):
    e = by[k]
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", e["name"])
    short = (m.group(1) + (m.group(2) or "")) if m else e["name"]
    if "digest" in short:
        # grid of the marker = ceil(words/256)*256 for tiny inputs; the order of markers is fixed
        seen_marks += 1
        phase = PH[[1, 4, 2, 3][(seen_marks - 1) % 4]]        # order of the markers in tools/prof_clock.py
        continue
    if "k_mul_flat" in short:
        dur = (e["end"] - e["start"]) * 1e-9
        if dur > 1e-3:
            out.append({"kernel": short, "phase": "multiply", "us": dur * 1e6,
                        "clock_GHz": e.get("GRBM_GUI_ACTIVE", 0) / 8 / dur / 1e9})
        continue
    if phase is None:
        continue
    dur = (e["end"] - e["start"]) * 1e-9
    out.append({"kernel": short, "phase": phase, "grid": e["grid"], "us": dur * 1e6,
                "clock_GHz": e.get("GRBM_GUI_ACTIVE", 0) / 8 / dur / 1e9,
                "G_wave_instr_per_s": e.get("SQ_INSTS_VALU", 0) / dur / 1e9,
                "cycles_per_wave_instr_per_SIMD": (e.get("GRBM_GUI_ACTIVE", 0) / 8) * 1024 / max(e.get("SQ_INSTS_VALU", 1), 1)})
summary = {}
for r in out:
    key = f'{r["kernel"]} grid={r.get("grid", 0)} {r["phase"]}'
    summary.setdefault(key, []).append(r)
res = {}
for key, v in summary.items():
    res[key] = {"n": len(v), "median_us": statistics.median(x["us"] for x in v),
                "median_clock_GHz": statistics.median(x["clock_GHz"] for x in v)}
    if "G_wave_instr_per_s" in v[0]:
        res[key]["median_G_wave_instr_per_s"] = statistics.median(x["G_wave_instr_per_s"] for x in v)
        res[key]["median_cycles_per_wave_instr_per_SIMD"] = statistics.median(x["cycles_per_wave_instr_per_SIMD"] for x in v)
json.dump({"what": "effective shader clock = GRBM_GUI_ACTIVE / 8 / kernel duration per dispatch (tools/prof_r03_clock.sh)",
           "per_kernel_and_phase": res}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(res, indent=1))
