#!/bin/bash
# HBM traffic of the secondary kernels against their algorithmic bytes: WRITE_SIZE and FETCH_SIZE in SEPARATE
# --pmc passes over tools/prof_traffic_ops.py (KiB units; FETCH_SIZE x2 on gfx950), program directly after `--`.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03_traffic
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o ops -- python3 tools/prof_traffic_ops.py > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
python3 - $OUT <<'PY'
import csv, glob, json, re, sys
out = sys.argv[1]
alg = json.loads([l for l in open(out + "/WRITE_SIZE.log") if l.startswith("ALG ")][-1][4:])
tot = {}
for C in ("WRITE_SIZE", "FETCH_SIZE"):
    f = glob.glob(out + "/" + C + "/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if not m: continue
        d = tot.setdefault(m.group(1), {"WRITE_SIZE": 0.0, "FETCH_SIZE": 0.0, "n": 0})
        d[C] += float(r["Counter_Value"])
        if C == "WRITE_SIZE": d["n"] += 1
lines = ["kernel                 per dispatch: written MB (algorithmic)   read MB x2-corrected (algorithmic)   case"]
for k, a in alg.items():
    d = tot.get(k)
    if not d or not d["n"]: continue
    w = d["WRITE_SIZE"] * 1024 / d["n"] / 1e6; r = d["FETCH_SIZE"] * 2 * 1024 / d["n"] / 1e6
    lines.append("%-22s %10.1f (%8.1f)   %10.1f (%8.1f)   %s   [%d dispatches]" % (k, w, a["written"] / 1e6, r, a["read"] / 1e6, a["what"], d["n"]))
for k in ("k_touch2", "k_hits_parity"):
    d = tot.get(k)
    if d and d["n"]:
        lines.append("%-22s %10.1f              %10.1f              (helper)   [%d dispatches]" % (k, d["WRITE_SIZE"] * 1024 / d["n"] / 1e6, d["FETCH_SIZE"] * 2 * 1024 / d["n"] / 1e6, d["n"]))
open(out + "/ops_traffic.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/WRITE_SIZE $OUT/FETCH_SIZE
