#!/bin/bash
OUT=gpurun_out/r04zn; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs"
RUHVRO_HIP_PERSIST=1 timeout 300 python scripts/parity_quick.py 1000000 > $OUT/parity_persist.log 2>&1; echo "parity persist rc=$?"
for v in 0 1; do
RUHVRO_HIP_PERSIST=$v RUHVRO_HIP_PROFILE=1 timeout 200 python bench.py --steps 4 --warmup 2 $B 2>&1 >/dev/null | grep -a "profile\] size" | grep -v "stage+barrier=0 " | tail -1 | tee $OUT/clock_$v.txt
done
STEPS=20 bash scripts/gpu_env_ab.sh r04zn "base_a:" "persist_a:RUHVRO_HIP_PERSIST=1" "base_b:" "persist_b:RUHVRO_HIP_PERSIST=1"
