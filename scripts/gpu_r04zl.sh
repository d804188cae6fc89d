#!/bin/bash
# r04zl: what the kernel timestamps inside the timed steps cost the headline (bench.py --stats-every N: every Nth step carries them)
OUT=gpurun_out/r04zl; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs"
for r in a b; do
for se in 1 2 4 5 10; do
  timeout 200 python bench.py --steps 20 --warmup 5 --stats-every $se $B > $OUT/b_${se}_$r.json 2> $OUT/b_${se}_$r.err
  python -c "
import json; d=json.load(open('$OUT/b_${se}_$r.json')); print('stats-every %2d  ms/step %.4f  value %.4g  %s  sync %.4f  launches %s' % ($se, d['ms_per_step'], d['value'], {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}, d['config']['sync_call_ms'], d['config'].get('timed_launches')))"
done
done 2>&1 | tee $OUT/summary.txt
