#!/bin/bash
# r04s: single-pass form, first generation of tiles staggered (RH_V_STAGGER)
OUT=gpurun_out/r04s; mkdir -p $OUT; export TMPDIR=/tmp; export RUHVRO_HIP_SINGLE_PASS=1
STEPS=20 timeout 600 bash scripts/gpu_env_ab.sh r04s "plain:" "stagger:RUHVRO_HIP_VARIANT=STAGGER" "plain2:" "stagger2:RUHVRO_HIP_VARIANT=STAGGER"
B="--no-cpu-baseline --no-end-to-end --no-projection --no-other-configs --overlap-streams 0"
RUHVRO_HIP_VARIANT=STAGGER RUHVRO_HIP_PROFILE=1 timeout 200 python bench.py --steps 4 --warmup 2 $B > $OUT/prof.json 2> $OUT/prof.err; grep "single-pass cycles" $OUT/prof.err | tail -1
