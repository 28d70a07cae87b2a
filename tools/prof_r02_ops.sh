#!/bin/bash
# VALU evidence for the two issue-bound secondary kernels (keyed encrypt, bit-plane permutation):
# kernel durations, then SQ_INSTS_VALU / SQ_WAVES / SQ_BUSY_CYCLES in a separate --pmc run.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02_ops
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ops -- python3 tools/prof_ops_small.py > $OUT/trace.log 2>&1
echo "trace rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -o ops -- python3 tools/prof_ops_small.py > $OUT/pmc.log 2>&1
echo "pmc rc=$?"
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/ops_kernel_stats.csv
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
(head -1 $f; grep -E "k_encrypt_wave|k_permute_planes" $f) > $OUT/ops_pmc_valu.csv
rm -rf $OUT/trace $OUT/pmc
ls -la $OUT
