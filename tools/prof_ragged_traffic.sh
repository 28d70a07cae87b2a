#!/bin/bash
# HBM traffic of the wave-cooperative ragged multiply by PMC (WRITE_SIZE and FETCH_SIZE in separate passes; KiB units,
# FETCH_SIZE x2 on gfx950), per dispatch, against the algorithmic bytes.  usage: bash tools/prof_ragged_traffic.sh OUTDIR
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for M in 8 16; do
  for C in WRITE_SIZE FETCH_SIZE; do
    MEAN=$M CALLS=6 rocprofv3 --pmc $C --output-format csv -d $OUT/p -o r -- python3 tools/prof_ragged_valu.py > $OUT/run.log 2>&1
    f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
    echo "mean $M $C: $(python3 tools/pmc_kernels.py $f k_mul_ragged_coop)  [$(grep 'out terms' $OUT/run.log)]" >> $OUT/ragged_traffic.txt
    rm -rf $OUT/p
  done
done
cat $OUT/ragged_traffic.txt
