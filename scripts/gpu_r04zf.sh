#!/bin/bash
# r04zf: phase clock of rh_espec_emit (Arrow -> Avro) at 2M and 10M rows: which stretches of the walk the time is in
OUT=gpurun_out/r04zf; mkdir -p $OUT; export TMPDIR=/tmp
for rows in 2000000 10000000; do
  RUHVRO_HIP_PROFILE=1 timeout 300 python bench.py --direction encode --rows $rows --steps 3 --warmup 1 2>&1 >/dev/null | grep -a "profile" | tail -2 | tee $OUT/clock_$rows.txt
done
