#!/bin/bash
# r04z3: field-split emit: phase clock per wavefront set, and the SQ counter groups of the split against the one-wave kernel of the same code object
OUT=gpurun_out/r04z3; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs"
for v in split one; do
  if [ $v = one ]; then export RUHVRO_HIP_SPLIT_EMIT=0; else unset RUHVRO_HIP_SPLIT_EMIT; fi
  RUHVRO_HIP_PROFILE=1 timeout 200 python bench.py --steps 4 --warmup 2 $B 2>&1 >/dev/null | grep -a "profile\] emit" | tail -2 > $OUT/clock_$v.txt; echo "== clock $v"; cat $OUT/clock_$v.txt
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
             "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
             "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU_MFMA_I8 SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY" \
             "TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES TCP_TOTAL_WRITE TCP_TCC_WRITE_REQ"; do
    i=$((i+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $set -d $OUT/p_$v$i -o p -- python bench.py --steps 2 --warmup 1 $B > $OUT/p_$v$i.log 2>&1; echo "pass $v $i rc=$?"
    for f in $(find $OUT/p_$v$i -name "*.db"); do python scripts/rocpd_summary.py $f 2>&1 | grep -E "^rh_spec_emit" >> $OUT/pmc_$v.txt; done
    rm -rf $OUT/p_$v$i
  done
done
python - <<'PY'
import re
def load(p):
    d={}
    for l in open(p):
        f=l.split()
        if len(f)>=4 and f[1][0].isalpha(): d[f[1]]=float(f[3])
    return d
a=load("gpurun_out/r04z3/pmc_split.txt"); b=load("gpurun_out/r04z3/pmc_one.txt")
for k in sorted(set(a)|set(b)): print("%-34s split %16.0f   one %16.0f   x%.3f" % (k, a.get(k,0), b.get(k,0), a.get(k,0)/b[k] if b.get(k) else 0))
PY
