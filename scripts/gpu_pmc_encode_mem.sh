#!/bin/bash
# Memory-path counter passes (TA / TCP / TCC) over the Arrow -> Avro bench line.  Usage: bash scripts/gpu_pmc_encode_mem.sh tag [rows]
TAG=${1:-pmc_encm}
ROWS=${2:-4000000}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
# (a TA_* pass -- TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_READ_WAVEFRONTS -- never
#  returned on this workload and ran into its 300 s timeout: not collected)
for set in "TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES TCP_TOTAL_READ TCP_TCC_READ_REQ" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TOTAL_ACCESSES TCP_TOTAL_CACHE_ACCESSES" \
           "TCP_TCC_READ_REQ_LATENCY TCP_TA_TCP_STATE_READ TCP_GATE_EN1 TCP_GATE_EN2" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p$i -- python bench.py --direction encode --rows $ROWS --steps 2 --warmup 1 > $OUT/p$i.log 2>&1; echo "pass $i rc=$?"
  for f in $(find $OUT/p$i -name "*.db"); do python scripts/rocpd_summary.py $f 2>&1 | grep -E "^rh_espec_emit" > $OUT/p$i.txt; done
  rm -rf $OUT/p$i
done
cat $OUT/p*.txt
