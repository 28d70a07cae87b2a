#!/bin/bash
# Round-4 evidence (on the GPU box: bash tools/prof_r04.sh [quick]); programs directly after `--`.
#   1. headline, the driver's configuration: rocprofv3 --kernel-trace --stats, then --pmc WRITE_SIZE and
#      FETCH_SIZE in separate passes (one step each)                      -> kernel_stats.csv, pmc_*.csv, traffic
#   2. compaction (csgn_compact_ragged, row f4): kernel trace + stats of tools/bench_compact.py on the
#      4096 x 1024-term cases, and the same two PMC passes with one call per case  -> compact_*.{csv,txt}
#   3. plain logs: bench.py default and --native-ranks, bench_compact, bench_ragged, bench_ops, wirebench
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace rc=$?"
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
  echo "$C rc=$?"
done
python3 tools/pmc_summary.py $OUT $OUT/r04 k_touch+k_mul_flat --traffic-json $OUT/traffic_current.json profiles/r04
for C in WRITE_SIZE FETCH_SIZE; do
  f=$(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1)
  (head -1 $f; grep -E "k_touch|k_mul_flat|k_synth_fill" $f | head -400) > $OUT/pmc_$C.csv
done
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $OUT/bench_trace.json $OUT/bench_under_rocprof.json
rm -rf $OUT/trace $OUT/pmc_WRITE_SIZE $OUT/pmc_FETCH_SIZE
# ---- compaction
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ctrace -o compact -- python3 tools/bench_compact.py --only "4096 x 1024 terms" > $OUT/compact_under_rocprof.log 2>&1
echo "compact trace rc=$?"
cp $(find $OUT/ctrace -name "*kernel_stats.csv" | head -1) $OUT/compact_kernel_stats.csv
rm -rf $OUT/ctrace
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/c_$C -o compact -- python3 tools/bench_compact.py --once --only "4096 x 1024 terms" > $OUT/compact_pmc_$C.log 2>&1
  echo "compact $C rc=$?"
done
python3 - $OUT <<'PY'
import csv, glob, re, sys
out = sys.argv[1]
cases = []          # (name, in terms, out terms) in run order
for ln in open(out + "/compact_pmc_WRITE_SIZE.log"):
    m = re.match(r"(compact .*?)\s+[\d.]+ ms\s+in\s+(\d+) out\s+(\d+) terms", ln)
    if m:
        cases.append((m.group(1).strip(), int(m.group(2)), int(m.group(3))))
vals = {}
for C in ("WRITE_SIZE", "FETCH_SIZE"):
    f = glob.glob(out + "/c_" + C + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_compact_main" in r["Kernel_Name"] and r.get("Counter_Name") == C]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals[C] = [float(r["Counter_Value"]) for r in rows]
lines = ["k_compact_main, per dispatch (KiB counters; FETCH_SIZE x2 on gfx950): written MB (algorithmic)  read MB (algorithmic)  traffic / algorithmic   case"]
n = min(len(vals["WRITE_SIZE"]), len(vals["FETCH_SIZE"]))
per = max(1, n // max(1, len(cases)))                     # dispatches per case (--once: the set-up call + one timed call)
for i, (name, tin, tout) in enumerate(cases):
    w = vals["WRITE_SIZE"][i * per:(i + 1) * per]; r = vals["FETCH_SIZE"][i * per:(i + 1) * per]
    if not w or not r: continue
    wmb = sum(w) / len(w) * 1024 / 1e6; rmb = sum(r) / len(r) * 2 * 1024 / 1e6
    aw, ar = tout * 160 / 1e6, tin * 160 / 1e6
    lines.append("%10.1f (%8.1f)  %10.1f (%8.1f)  %6.3f   %s  [%d dispatches]" % (wmb, aw, rmb, ar, (wmb + rmb) / (aw + ar), name, len(w)))
open(out + "/compact_traffic.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/c_WRITE_SIZE $OUT/c_FETCH_SIZE
# ---- plain logs
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python3 bench.py --native-ranks > $OUT/bench_native_ranks.json 2> $OUT/bench_native_ranks.err; echo "native rc=$?"
python3 bench.py --native-ranks --force-collective --no-cpu-baseline > $OUT/bench_native_ranks_collective.json 2>> $OUT/bench_native_ranks.err; echo "native+collective rc=$?"
python3 tools/bench_compact.py --json $OUT/compact.json > $OUT/compact.log 2>&1; echo "compact rc=$?"
tests/cpp/dropin_driver wirebench > $OUT/wirebench.log 2>&1; echo "wirebench rc=$?"
(python3 tools/prof_compact_phases.py 0.0; python3 tools/prof_compact_phases.py 0.5; python3 tools/prof_compact_phases.py 0.95) 2>&1 | grep -v amdgpu > $OUT/compact_phases.log; echo "phases rc=$?"
SHORT=1 python3 tools/bench_ragged.py 2>&1 | grep -v amdgpu | grep -E "^mul_ragged|kernel only|async" > $OUT/bench_ragged_short.log; echo "ragged short rc=$?"
if [ "$1" != "quick" ]; then
  python3 tools/bench_ragged.py > $OUT/bench_ragged.log 2>&1; echo "ragged rc=$?"
  python3 tools/bench_ops.py --json $OUT/ops.json > $OUT/ops.log 2>&1; echo "ops rc=$?"
fi
ls -la $OUT
