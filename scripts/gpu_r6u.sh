#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r06_u && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
O=gpurun_out/r06_u
python scripts/workload_probe.py wide200 400000 --reps 2 --no-parity > /dev/null 2>&1     # (kernels into the cache)
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc$i -o p$i -- python scripts/workload_probe.py wide200 400000 --reps 2 --no-parity > $O/pmc$i.log 2>&1; echo "pass $i rc=$?"
  for f in $(find $O/pmc$i -name "*.db"); do python scripts/rocpd_summary.py $f; done 2>&1 | grep -E "^(rh_spec|kernel)" > $O/pmc$i.txt; rm -rf $O/pmc$i
done
cat $O/pmc*.txt
