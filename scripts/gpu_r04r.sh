#!/bin/bash
# r04r: single-pass form, raised wave priority up to the look-back (RH_V_NOPRIO = without)
OUT=gpurun_out/r04r; mkdir -p $OUT; export TMPDIR=/tmp; export RUHVRO_HIP_SINGLE_PASS=1
STEPS=20 timeout 600 bash scripts/gpu_env_ab.sh r04r "prio:" "noprio:RUHVRO_HIP_VARIANT=NOPRIO" "twopass:RUHVRO_HIP_SINGLE_PASS=0" "prio2:" "noprio2:RUHVRO_HIP_VARIANT=NOPRIO" "twopass2:RUHVRO_HIP_SINGLE_PASS=0"
B="--no-cpu-baseline --no-end-to-end --no-projection --no-other-configs --overlap-streams 0"
for v in "" "NOPRIO"; do
RUHVRO_HIP_VARIANT=$v RUHVRO_HIP_PROFILE=1 timeout 200 python bench.py --steps 4 --warmup 2 $B > $OUT/prof_$v.json 2> $OUT/prof_$v.err; echo "variant=$v"; grep "single-pass cycles" $OUT/prof_$v.err | tail -1
done
