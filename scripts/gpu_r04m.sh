#!/bin/bash
# r04m: static head fusion (walk.h read_head LA: adjacent heads decoded out of one window read) vs RUHVRO_HIP_NO_HEAD_FUSION=1
OUT=gpurun_out/r04m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_specialize.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
STEPS=30 bash scripts/gpu_env_ab.sh r04m "fused:" "nofusion:RUHVRO_HIP_NO_HEAD_FUSION=1" "fused2:" "nofusion2:RUHVRO_HIP_NO_HEAD_FUSION=1" "fused3:" "nofusion3:RUHVRO_HIP_NO_HEAD_FUSION=1"
for v in "fused:" "nofusion:RUHVRO_HIP_NO_HEAD_FUSION=1"; do
  name=${v%%:*}; envs=${v#*:}
  ( for kv in $envs; do export "$kv"; done
    timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/p_$name -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs > $OUT/p_$name.log 2>&1 )
  for f in $(find $OUT/p_$name -name "*.db"); do python scripts/rocpd_summary.py $f 2>&1 | grep -E "^rh_spec" > $OUT/pmc_$name.txt; done
  rm -rf $OUT/p_$name
  echo "== $name"; grep -E "INSTS|WAVE_CYCLES|rh_spec_(emit|size)  " $OUT/pmc_$name.txt
done
