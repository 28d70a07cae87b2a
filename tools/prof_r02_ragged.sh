#!/bin/bash
# PMC study of the ragged multiply against the uniform flat kernel at the same mean shape.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02_ragged
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- python3 tools/prof_ragged_small.py > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SMEM" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/pmc$i -o r -- python3 tools/prof_ragged_small.py > $OUT/pmc$i.log 2>&1
  echo "group $i rc=$?"
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  (head -1 $f; grep -E "k_mul_ragged|k_mul_flat|k_touch" $f) > $OUT/pmc$i.csv
done
rm -rf $OUT/trace $OUT/pmc[0-9]
ls -la $OUT
