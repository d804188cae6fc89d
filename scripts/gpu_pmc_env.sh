#!/bin/bash
# PMC passes over the headline bench for one env configuration (separate passes, kernel-trace + pmc only).
# Usage: bash scripts/gpu_pmc_env.sh tag "VAR=val VAR2=val" [workload]
TAG=${1:-pmc}; ENVS=$2; W=${3:-full10m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for kv in $ENVS; do export "$kv"; done
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS" \
           "TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES TCP_TOTAL_WRITE TCP_TCC_WRITE_REQ" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p$i -- python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/p$i.log 2>&1; echo "pass $i rc=$?"
  for f in $(find $OUT/p$i -name "*.db"); do python scripts/rocpd_summary.py $f 2>&1 | grep -E "^(rh_spec|kernel)" > $OUT/p$i.txt; done
  rm -rf $OUT/p$i
done
cat $OUT/p*.txt
