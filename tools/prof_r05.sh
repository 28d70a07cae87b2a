#!/bin/bash
# Round-5 evidence (on the GPU box: bash tools/prof_r05.sh); programs directly after `--`.
#   1. headline, the driver's configuration: rocprofv3 --kernel-trace --stats, then --pmc WRITE_SIZE and FETCH_SIZE in
#      separate passes (one step each)                                      -> kernel_stats.csv, pmc_*.csv, traffic
#   2. plain logs: bench.py default (with the secondary suite) and --native-ranks, config 5 as tape / compiled graphs,
#      the ragged entry points (bounded and not), compaction, wire format + host mirror, class-API latencies
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r05
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --no-cpu-baseline --no-secondary > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace rc=$?"
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --no-secondary > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
  echo "$C rc=$?"
done
python3 tools/pmc_summary.py $OUT $OUT/r05 k_touch+k_mul_flat --traffic-json $OUT/traffic_current.json profiles/r05
for C in WRITE_SIZE FETCH_SIZE; do
  f=$(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1)
  (head -1 $f; grep -E "k_touch|k_mul_flat|k_synth_fill" $f | head -400) > $OUT/pmc_$C.csv
done
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $OUT/bench_trace.json $OUT/bench_under_rocprof.json
rm -rf $OUT/trace $OUT/pmc_WRITE_SIZE $OUT/pmc_FETCH_SIZE
# ---- plain logs
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python3 bench.py --native-ranks --no-secondary > $OUT/bench_native_ranks.json 2> $OUT/bench_native_ranks.err; echo "native rc=$?"
python3 tools/bench_graph.py 2>&1 | grep -v amdgpu > $OUT/bench_graph.log; echo "graph rc=$?"
python3 tools/prof_ragged_ops.py 2>&1 | grep -v amdgpu > $OUT/ragged_ops.log; echo "ragged ops rc=$?"
SHORT=1 CHUNKS=0 python3 tools/bench_ragged.py 2>&1 | grep -v amdgpu | grep -E "^mul_ragged|kernel only|async|kernels only" > $OUT/bench_ragged_short.log; echo "ragged short rc=$?"
python3 tools/bench_compact.py --json $OUT/compact.json 2>&1 | grep -v amdgpu > $OUT/compact.log; echo "compact rc=$?"
tests/cpp/dropin_driver wirebench > $OUT/wirebench.log 2>&1; echo "wirebench rc=$?"
tests/cpp/dropin_driver latency 20000 > $OUT/latency.log 2>&1; echo "latency rc=$?"
ls -la $OUT
