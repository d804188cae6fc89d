#!/bin/bash
# r04g: whole GPU suite on the tree with the batched string copy + packed scan; then A/B of LDS-DMA window staging (RH_V_LDSDMA,
# VERDICT r3 item 1b) and of the batched copy (RH_V_NOBATCH = the per-piece round trips of round 3), with LDS / VALU counters
OUT=gpurun_out/r04g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash scripts/gpu_ab.sh r04g "" "LDSDMA" "NOBATCH" "" "LDSDMA" "NOBATCH"
for v in "" "LDSDMA"; do
  export RUHVRO_HIP_VARIANT=$v; name=${v:-new}
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/p_$name -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs > $OUT/p_$name.log 2>&1; echo "pmc $name rc=$?"
  for f in $(find $OUT/p_$name -name "*.db"); do python scripts/rocpd_summary.py $f 2>&1 | grep -E "^rh_spec" > $OUT/pmc_$name.txt; done
  rm -rf $OUT/p_$name
  echo "== $name"; grep -E "INSTS|WAVE_CYCLES|rh_spec_(emit|size)  " $OUT/pmc_$name.txt
done
