#!/bin/bash
# SQ-level PMC passes over bench.py (separate passes, kernel-trace only)
TAG=${1:-pmc}
W=${2:-full10m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_INSTS_BRANCH" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p$i -- python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/p$i.log 2>&1; echo "pass $i rc=$?"
  tail -3 $OUT/p$i.log | cut -c1-300
done
