#!/bin/bash
# rocprofv3 --kernel-trace --stats of the compaction of 4 ciphertexts x 2^20 terms (the hash-partition path).  usage: bash tools/prof_compact_large.sh OUTDIR
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o r -- python3 tools/bench_compact.py --only "4 x 2^20 terms, 0%" --rounds 3 > $OUT/compact_large_run.log 2>&1
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/compact_large_kernel_stats.csv
rm -rf $OUT/t
python3 - $OUT/compact_large_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:90]:<90} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
grep compact $OUT/compact_large_run.log | cut -c1-160
