#!/bin/bash
# r04n: larger tiles on the round-4 kernels (TILE = 384 / 512 records per workgroup; round 2 swept them on other kernels)
OUT=gpurun_out/r04n; mkdir -p $OUT; export TMPDIR=/tmp
STEPS=20 bash scripts/gpu_env_ab.sh r04n "t256:" "t384:RUHVRO_HIP_TILE=384" "t512:RUHVRO_HIP_TILE=512" "t256b:" "t384b:RUHVRO_HIP_TILE=384" "t512b:RUHVRO_HIP_TILE=512"
