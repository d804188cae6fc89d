#!/bin/bash
# r03s: what the kernel-timestamp steps cost the pipelined loop, and the GPU timeline of the asynchronous loop
OUT=gpurun_out/r03s; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-projection"
for w in full1m cfg3_1m flat4_1m; do
  for se in 2 1000; do
  timeout 200 python bench.py --workload $w --steps 60 --warmup 5 $B --stats-every $se > $OUT/b.json 2> $OUT/b.err || tail -5 $OUT/b.err
  python -c "
import json; d=json.load(open('$OUT/b.json')); print('$w stats-every $se', round(d['ms_per_step'],4), 'sync_call_ms', round(d['config'].get('sync_call_ms') or 0,4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()})"
  done
done
timeout 200 rocprofv3 --kernel-trace -d $OUT/p -o t -- python bench.py --workload full1m --steps 12 --warmup 3 $B --stats-every 4 > $OUT/p.log 2>&1
for f in $(find $OUT/p -name "*.db"); do python scripts/rocpd_timeline.py $f 60 2>&1 | tail -62; done
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
