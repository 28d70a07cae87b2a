#!/bin/bash
# usage: bash tools/prof_async_timeline.sh OUTDIR   (on the GPU box)
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o r -- python3 tools/prof_async_timeline.py > $OUT/async_timeline_run.log 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $OUT/async_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda s: s.split("(")[0].replace("void ", "").replace("csgn::", "").replace("(anonymous namespace)::", "")[:40]
# the last 40 kernels: name, duration, gap to the previous kernel's end
prev = None
tail = rows[-40:]
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{short(r['Kernel_Name']):<42} dur {(e - s) / 1e3:8.2f} us   gap before {gap:7.2f} us   grid {r.get('Grid_Size_X', r.get('Grid_Size',''))}")
    prev = e
PY
rm -rf $OUT/t
cat $OUT/async_timeline.txt
