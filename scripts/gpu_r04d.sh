#!/bin/bash
# r04d: emit diet 1 -- 8-byte string pieces at constant offsets (vs RH_V_OLDCOPY8), packed wave scan + lane-parallel workgroup prefix:
# bench A/B + parity, then SQ_INSTS_VALU / wave cycles per kernel for both
OUT=gpurun_out/r04d; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_ab.sh r04d "" "OLDCOPY8" "" "OLDCOPY8"
for v in "" "OLDCOPY8"; do
  export RUHVRO_HIP_VARIANT=$v; name=${v:-new}
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/p_$name -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs > $OUT/p_$name.log 2>&1; echo "pmc $name rc=$?"
  for f in $(find $OUT/p_$name -name "*.db"); do python scripts/rocpd_summary.py $f 2>&1 | grep -E "^rh_spec" > $OUT/pmc_$name.txt; done
  rm -rf $OUT/p_$name
  echo "== $name"; cat $OUT/pmc_$name.txt
done
