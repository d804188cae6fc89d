#!/bin/bash
# PMC passes over the headline bench (separate passes, kernel-trace + pmc only -- never with the trace domains gpurun refuses).
# (the TA_* group -- TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_WRITE_WAVEFRONTS -- returned once in
#  round 2 and hung twice until its timeout: not collected any more)
# Usage: bash scripts/gpu_pmc_r02.sh tag [workload]      (RUHVRO_HIP_VARIANT is honoured)
TAG=${1:-pmc}
W=${2:-full10m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
           "TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES TCP_TOTAL_WRITE TCP_TCC_WRITE_REQ" \
           "TCP_WRITE_TAGCONFLICT_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p$i -- python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 > $OUT/p$i.log 2>&1; echo "pass $i rc=$?"
  for f in $(find $OUT/p$i -name "*.db"); do python scripts/rocpd_summary.py $f 2>&1 | grep -E "^(rh_|kernel)" > $OUT/p$i.txt; done
  rm -rf $OUT/p$i
done
cat $OUT/p*.txt
