#!/bin/bash
# wall clock per call, then rocprofv3 --kernel-trace --stats of the same calls (on the GPU box).  usage: bash tools/prof_ragged_ops.sh OUTDIR
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 tools/prof_ragged_ops.py > $OUT/ragged_ops.log 2>&1
cat $OUT/ragged_ops.log | grep -v amdgpu.ids
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o r -- python3 tools/prof_ragged_ops.py > $OUT/ragged_ops_under_rocprof.log 2>&1
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/ragged_ops_kernel_stats.csv
rm -rf $OUT/t
cut -d, -f1-4 $OUT/ragged_ops_kernel_stats.csv | head -30
