#!/bin/bash
# Effective shader clock of the issue-bound kernels, idle vs right after memory-heavy work
# (VERDICT r2 #5).  One --pmc pass with kernel timestamps; the program directly after `--`.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03_clock
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc -o clock -- python3 tools/prof_clock.py > $OUT/pmc.log 2>&1
echo "pmc rc=$?"
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
(head -1 $f; grep -E "k_encrypt_wave|k_permute_planes|k_digest|k_mul_flat" $f) > $OUT/clock_pmc.csv
python3 tools/clock_summary.py $OUT/clock_pmc.csv $OUT/clock_summary.json
rm -rf $OUT/pmc
ls -la $OUT
