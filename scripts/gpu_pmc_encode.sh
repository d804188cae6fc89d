#!/bin/bash
# SQ counter passes over the Arrow -> Avro bench line (separate passes, kernel-trace + pmc only).
# Usage: bash scripts/gpu_pmc_encode.sh tag [rows]
TAG=${1:-pmc_enc}
ROWS=${2:-4000000}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_MEM_VIOLATIONS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p$i -- python bench.py --direction encode --rows $ROWS --steps 2 --warmup 1 > $OUT/p$i.log 2>&1; echo "pass $i rc=$?"
  for f in $(find $OUT/p$i -name "*.db"); do python scripts/rocpd_summary.py $f 2>&1 | grep -E "^(rh_e|kernel)" > $OUT/p$i.txt; done
  rm -rf $OUT/p$i
done
cat $OUT/p*.txt
