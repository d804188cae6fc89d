#!/bin/bash
# r04a: baseline of the round-3 tree on this round's box + tile-size sweep (TILE = 64 / 128 / 256 records per workgroup) with the
# in-kernel phase clock of each (RUHVRO_HIP_PROFILE=1)
OUT=gpurun_out/r04a; mkdir -p $OUT; export TMPDIR=/tmp
STEPS=20 bash scripts/gpu_env_ab.sh r04a "t256:" "t128:RUHVRO_HIP_TILE=128" "t64:RUHVRO_HIP_TILE=64" "t256b:" "t128b:RUHVRO_HIP_TILE=128" "t64b:RUHVRO_HIP_TILE=64"
for t in 256 128 64; do
  RUHVRO_HIP_TILE=$t RUHVRO_HIP_PROFILE=1 timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs > $OUT/prof_t$t.json 2> $OUT/prof_t$t.err
  grep "ruhvro_hip profile" $OUT/prof_t$t.err | tail -2
done
