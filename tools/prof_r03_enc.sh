#!/bin/bash
# Where do the waves of the keyed encrypt / fused chain / permutation spend their cycles?
# One --pmc pass (SQ block, 7 of 8 slots) over tools/prof_ops_small.py, the program directly after `--`.
#   SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked on s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall)
#                    + SQ_ACTIVE_INST_ANY (issuing), all in quad-cycles (MI355X_MICROARCH.md, PMC slots)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03_enc
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc -o enc -- python3 tools/prof_ops_small.py > $OUT/pmc.log 2>&1
echo "pmc rc=$?"
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
(head -1 $f; grep -E "k_encrypt_wave|k_encrypt_mul_wave|k_permute_planes" $f) > $OUT/enc_pmc.csv
python3 - "$OUT/enc_pmc.csv" "$OUT/enc_pmc_summary.json" <<'PY'
import csv, json, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
by = {}
for r in rows:
    import re as _re
    _m = _re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", r["Kernel_Name"])
    e = by.setdefault(int(r["Dispatch_Id"]), {"name": (_m.group(1) + (_m.group(2) or "")) if _m else r["Kernel_Name"], "grid": int(r["Grid_Size"]),
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
              "G_wave_instr_per_s": med("SQ_INSTS_VALU") / (med("us") * 1e-6) / 1e9 if med("us") else 0,
              "frac_wait_any": med("SQ_WAIT_ANY") / wc, "frac_wait_inst_any": med("SQ_WAIT_INST_ANY") / wc,
              "frac_active_inst_any": med("SQ_ACTIVE_INST_ANY") / wc, "frac_active_inst_valu": med("SQ_ACTIVE_INST_VALU") / wc}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/pmc
ls -la $OUT
