#!/bin/bash
# The two PMC passes that tie profiles/traffic_current.json to the multiply's current source (on the GPU box; the first
# lines of tools/prof_r04.sh without the trace): rocprofv3 --pmc WRITE_SIZE, then FETCH_SIZE, one bench step each.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_traffic
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
  echo "$C rc=$?"
done
python3 tools/pmc_summary.py $OUT $OUT/r04 k_touch+k_mul_flat --traffic-json $OUT/traffic_current.json profiles/r04
for C in WRITE_SIZE FETCH_SIZE; do
  f=$(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1)
  (head -1 $f; grep -E "k_touch|k_mul_flat|k_synth_fill" $f | head -400) > $OUT/pmc_$C.csv
done
rm -rf $OUT/pmc_WRITE_SIZE $OUT/pmc_FETCH_SIZE
ls -la $OUT
