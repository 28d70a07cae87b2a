#!/bin/bash
# Profile the bench command with rocprofv3 (kernel trace + stats), then PMC passes for HBM traffic.
# Usage (on the GPU box): bash tools/prof_bench.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py "$@" --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace rc=$?"
ls -R $OUT/trace | head -20
