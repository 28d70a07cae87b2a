#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, one counter per pass
# (MI355X_MICROARCH.md "HBM": FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 -- separate passes).
# Usage (on the GPU box): bash tools/prof_pmc.sh <tag> [bench args...]
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o bench -- python3 bench.py "$@" --no-cpu-baseline --no-verify > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
  echo "$C rc=$?"
done
ls -R $OUT | grep -c csv
