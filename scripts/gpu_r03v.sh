#!/bin/bash
OUT=gpurun_out/r03v; mkdir -p $OUT
B="--no-cpu-baseline --no-end-to-end"
for w in full10m full1m cfg3_1m; do
  for st in 1 2 3; do
  timeout 200 python bench.py --workload $w --steps 60 --warmup 6 $B --streams $st --stats-every 4 > $OUT/b.json 2> $OUT/b.err || tail -5 $OUT/b.err
  python -c "
import json; d=json.load(open('$OUT/b.json')); print('$w streams $st', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}); p=d.get('config5_projection'); print({g: (round(v['ms_per_step'],4), round(v['implied_efficiency'],3)) for g,v in p['g'].items()} if p else '')"
  done
done
