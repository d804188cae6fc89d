#!/bin/bash
# r04z2: field-split emit at other occupancies: registers held to 8 waves per SIMD (4 workgroups of 8 waves per CU), TILE = 128 (7 / 8 workgroups of 4 waves)
OUT=gpurun_out/r04z2; mkdir -p $OUT; export TMPDIR=/tmp
for r in a b; do
STEPS=20 bash scripts/gpu_env_ab.sh r04z2 "one_$r:RUHVRO_HIP_SPLIT_EMIT=0" "w6_$r:" "w8_$r:RUHVRO_HIP_SPLIT_WAVES=8" "t128w7_$r:RUHVRO_HIP_TILE=128 RUHVRO_HIP_SPLIT_WAVES=7" "t128w8_$r:RUHVRO_HIP_TILE=128 RUHVRO_HIP_SPLIT_WAVES=8"
done
