#!/bin/bash
# Where do the waves of the ragged multiply spend their cycles on the log-normal batch, beside the uniform
# flat kernel at the mean shape?  One --pmc pass (SQ block) over tools/prof_ragged_small.py.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03_ragged
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc -o rag -- python3 tools/prof_ragged_small.py > $OUT/pmc.log 2>&1
echo "pmc rc=$?"
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
(head -1 $f; grep -E "k_mul_ragged_flat|k_mul_flat|k_mul_tiled|k_touch" $f) > $OUT/ragged_pmc.csv
python3 - "$OUT/ragged_pmc.csv" "$OUT/ragged_pmc_summary.json" <<'PY'
import csv, json, re, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
by = {}
for r in rows:
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", r["Kernel_Name"])
    e = by.setdefault(int(r["Dispatch_Id"]), {"name": (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"], "grid": int(r["Grid_Size"]),
                                              "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                              "vgpr": int(r["VGPR_Count"]), "lds": int(r["LDS_Block_Size"])})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
groups = {}
for e in by.values():
    groups.setdefault(f'{e["name"]} grid={e["grid"]} vgpr={e["vgpr"]} lds={e["lds"]}', []).append(e)
out = {}
for k, v in groups.items():
    med = lambda key: statistics.median(x.get(key, 0.0) for x in v)
    wc = med("SQ_WAVE_CYCLES") or 1.0
    out[k] = {"n": len(v), "median_us": med("us"), "waves": med("SQ_WAVES"), "valu_wave_instr": med("SQ_INSTS_VALU"),
              "valu_per_wave": med("SQ_INSTS_VALU") / (med("SQ_WAVES") or 1.0),
              "G_wave_instr_per_s": med("SQ_INSTS_VALU") / (med("us") * 1e-6) / 1e9 if med("us") else 0,
              "frac_wait_any": med("SQ_WAIT_ANY") / wc, "frac_wait_inst_any": med("SQ_WAIT_INST_ANY") / wc,
              "frac_active_inst_any": med("SQ_ACTIVE_INST_ANY") / wc, "frac_active_inst_valu": med("SQ_ACTIVE_INST_VALU") / wc}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/pmc
