#!/bin/bash
# rocprofv3 --kernel-trace --stats of planned multiplies of the two long-tailed batches of small pairs (on the GPU box):
# the wave-cooperative kernel's durations in the profiler's own words.  usage: bash tools/prof_ragged_trace.sh OUTDIR
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for M in 8 16; do
  MEAN=$M CALLS=40 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$M -o r -- python3 tools/prof_ragged_valu.py > $OUT/run_mean$M.log 2>&1
  f=$(find $OUT/t$M -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/ragged_mean${M}_kernel_stats.csv
  rm -rf $OUT/t$M
  grep "out terms" $OUT/run_mean$M.log
done
head -4 $OUT/ragged_mean8_kernel_stats.csv | cut -c1-200
