#!/bin/bash
# r04zj: Arrow -> Avro, timing-only: what the row's validity-bit loads (NOBITS) and the second offset load of every top-level span (ONESPAN) cost
OUT=gpurun_out/r04zj; mkdir -p $OUT; export TMPDIR=/tmp
for r in a b; do
for v in "" NOBITS ONESPAN "NOBITS,ONESPAN"; do
  name=${v:-default}; name=${name//,/+}
  ( export RUHVRO_HIP_VARIANT=$v; timeout 300 python bench.py --direction encode --rows 2000000 --steps 8 --warmup 2 > $OUT/bench_${name}_$r.json 2> $OUT/bench_${name}_$r.err )
  python -c "
import json; d=json.load(open('$OUT/bench_${name}_$r.json')); print('%-16s %s' % ('$name', {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}))"
done
done 2>&1 | tee $OUT/summary.txt
