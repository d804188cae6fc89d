#!/bin/bash
# r04q: single-pass form with the parallel look-back: parity, A/B vs two-pass, phase clock
OUT=gpurun_out/r04q; mkdir -p $OUT; export TMPDIR=/tmp
RUHVRO_HIP_SINGLE_PASS=1 timeout 120 python scripts/single_pass_check.py 300000 8 > $OUT/check.log 2>&1; echo "check rc=$?"; tail -4 $OUT/check.log
RUHVRO_HIP_SINGLE_PASS=1 timeout 120 python scripts/single_pass_check.py 100000 3 > $OUT/check3.log 2>&1; echo "check k=3 rc=$?"; tail -4 $OUT/check3.log
STEPS=20 timeout 600 bash scripts/gpu_env_ab.sh r04q "two_pass:" "single:RUHVRO_HIP_SINGLE_PASS=1" "two_pass2:" "single2:RUHVRO_HIP_SINGLE_PASS=1"
B="--no-cpu-baseline --no-end-to-end --no-projection --no-other-configs --overlap-streams 0"
RUHVRO_HIP_SINGLE_PASS=1 RUHVRO_HIP_PROFILE=1 timeout 200 python bench.py --steps 4 --warmup 2 $B > $OUT/prof.json 2> $OUT/prof.err; grep "single-pass cycles" $OUT/prof.err | tail -1; tail -2 $OUT/prof.err
