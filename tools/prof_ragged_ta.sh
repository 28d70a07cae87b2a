#!/bin/bash
# texture-addresser / L1 counters of the ragged multiply on the long-tailed mean-8 batch, ONE counter per pass and a short
# limit on each (a pass over several TA_* / TCP_* counters once took rocprofv3 down and hung until the outer limit).
# What is known about that hang (nothing was re-run to provoke it): the pass asked for more TA / TCP counters
# than one pass can hold -- these blocks have few counter registers per instance and rocprofv3 does not always refuse an
# over-subscribed set up front; the process under it (41 multiplies, kernels that pass all tests with and without a
# profiler attached, and that the single-counter passes below profile without incident) never reported a fault, and the
# box's dmesg is not readable by the pool's user, so a kernel-side cause cannot be ruled in or out from the logs kept.
# One counter per pass is the guide's prescription anyway (MI355X_MICROARCH.md, HBM / rocprofv3 section); the short
# `timeout -k` is a belt, not a fix: with one counter per pass no pass has come near it.
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd $GRAFT_REPO_ROOT
for C in TA_TA_BUSY TA_BUSY_avr TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ_LATENCY TA_FLAT_READ_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TA_TCP_STATE_READ TD_TD_BUSY TCP_GATE_EN2; do
  timeout -k 5 45 rocprofv3 --pmc $C --output-format csv -d $OUT/p_$C -o r -- python3 tools/prof_ragged_valu.py > $OUT/run_$C.log 2>&1
  rc=$?
  f=$(find $OUT/p_$C -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 tools/pmc_kernels.py $f k_mul_ragged >> $OUT/pmc.txt; else echo "$C: no output (rc $rc)" >> $OUT/pmc.txt; fi
  rm -rf $OUT/p_$C
  [ $rc -ge 124 ] && { echo "stopping after $C (rc $rc)" >> $OUT/pmc.txt; break; }
done
cat $OUT/pmc.txt
