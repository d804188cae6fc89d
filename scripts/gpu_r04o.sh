#!/bin/bash
# r04o: the single-pass form (one kernel: size walk, look-back scan across tiles, emit walk): parity, then bench A/B
OUT=gpurun_out/r04o; mkdir -p $OUT; export TMPDIR=/tmp
RUHVRO_HIP_SINGLE_PASS=1 timeout 120 python scripts/single_pass_check.py 300000 8 > $OUT/check.log 2>&1; echo "check rc=$?"; tail -5 $OUT/check.log
RUHVRO_HIP_SINGLE_PASS=1 timeout 120 python scripts/single_pass_check.py 100000 3 > $OUT/check3.log 2>&1; echo "check k=3 rc=$?"; tail -4 $OUT/check3.log
STEPS=20 timeout 600 bash scripts/gpu_env_ab.sh r04o "two_pass:" "single:RUHVRO_HIP_SINGLE_PASS=1" "two_pass2:" "single2:RUHVRO_HIP_SINGLE_PASS=1"
