#!/bin/bash
# r04w: single-pass form, look-back polls through the XCD's L2 first (RH_V_SC0POLL) vs agent-scope loads only
OUT=gpurun_out/r04w; mkdir -p $OUT; export TMPDIR=/tmp; export RUHVRO_HIP_SINGLE_PASS=1
STEPS=20 timeout 600 bash scripts/gpu_env_ab.sh r04w "agent:" "sc0:RUHVRO_HIP_VARIANT=SC0POLL" "twopass:RUHVRO_HIP_SINGLE_PASS=0" "agent2:" "sc02:RUHVRO_HIP_VARIANT=SC0POLL" "twopass2:RUHVRO_HIP_SINGLE_PASS=0"
B="--no-cpu-baseline --no-end-to-end --no-projection --no-other-configs --overlap-streams 0"
for v in "" "SC0POLL"; do
RUHVRO_HIP_VARIANT=$v RUHVRO_HIP_PROFILE=1 timeout 200 python bench.py --steps 4 --warmup 2 $B > $OUT/prof_$v.json 2> $OUT/prof_$v.err; echo "variant=$v"; grep "single-pass cycles" $OUT/prof_$v.err | tail -1
done
