#!/bin/bash
# PMC counters of the ragged multiply kernels on the long-tailed mean-8 batch (dev tool; on the GPU box).
# usage: bash tools/prof_ragged_pmc.sh OUTDIR [knob=value ...]   -- knobs as CSGN_* environment variables; MEAN=16 for the other batch
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd $GRAFT_REPO_ROOT
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
         "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM" \
         "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_COALESCEABLE_WAVEFRONTS" \
         "TA_FLAT_READ_WAVEFRONTS TA_FLAT_WRITE_WAVEFRONTS TA_ADDR_STALLED_BY_TD_CYCLES TD_TD_BUSY" \
         "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
         "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TOTAL_READ" \
         "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY" \
         "TCP_GATE_EN1 TCP_GATE_EN2 TCP_TD_TCP_STALL_CYCLES TCP_WRITE_TAGCONFLICT_STALL_CYCLES"; do
  tag=$(echo $C | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/p_$tag -o r -- python3 tools/prof_ragged_valu.py > $OUT/run.log 2>&1
  f=$(find $OUT/p_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_kernels.py $f k_mul_ragged >> $OUT/pmc.txt || echo "no output for $C" >> $OUT/pmc.txt
  rm -rf $OUT/p_$tag
done
cat $OUT/pmc.txt
