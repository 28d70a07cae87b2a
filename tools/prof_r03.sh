#!/bin/bash
# Round-3 headline evidence, on the driver's configuration (python3 bench.py: batch 65536, 128 slots),
# the program directly after `--` (no wrapper: the profiler's preload initialises the GPU first).
#   1. rocprofv3 --kernel-trace --stats        -> per-kernel durations
#   2. rocprofv3 --pmc WRITE_SIZE, then FETCH_SIZE (separate passes, one step each) -> HBM traffic
# Usage (on the GPU box): bash tools/prof_r03.sh
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace rc=$?"
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
  echo "$C rc=$?"
done
find $OUT -name "*.csv" | head
# keep the small per-kernel summary; the per-dispatch counter files are cut to the multiply's kernels
python3 tools/pmc_summary.py $OUT $OUT/r03 k_touch+k_mul_flat --traffic-json $OUT/traffic_current.json profiles/r03
for C in WRITE_SIZE FETCH_SIZE; do
  f=$(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1)
  (head -1 $f; grep -E "k_touch|k_mul_flat|k_synth_fill" $f | head -400) > $OUT/pmc_$C.csv
done
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace $OUT/pmc_WRITE_SIZE $OUT/pmc_FETCH_SIZE
ls -la $OUT
