#!/bin/bash
# r04y: the emit kernel's wave scan run behind the window's LDS-DMA rows (one barrier for window + wave totals; first_bad looked at
# with the window bounds instead of in a round trip of its own) against the order of r04x (RH_V_NOOVL), three pairs + the phase clock
OUT=gpurun_out/r04y; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_ab.sh r04y "" "NOOVL" "" "NOOVL" "" "NOOVL"
for v in "" "NOOVL"; do
  name=${v:-new}
  RUHVRO_HIP_VARIANT=$v RUHVRO_HIP_PROFILE=1 timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs > $OUT/clock_$name.json 2> $OUT/clock_$name.err
  echo "== phase clock $name"; grep -a "profile" $OUT/clock_$name.err | tail -3
done
