#!/bin/bash
OUT=gpurun_out/r03w; mkdir -p $OUT
timeout 600 python bench.py --no-cpu-baseline --no-end-to-end > $OUT/bench_default_noextras.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench_default_noextras.json')); print(round(d['ms_per_step'],4), d['value'], {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}, 'frac', round(d['roofline']['frac'],3), 'path', round(d['roofline']['path_frac'],3), 'traffic', d['roofline']['traffic'], 'read_frac', d['roofline']['read_frac']); print(json.dumps(d['overlapped'])); print(json.dumps(d['config5_projection']['g']))"
timeout 300 python -m pytest tests/test_dist_gloo.py -q -m gpu 2>&1 | tail -2
for w in full1m cfg3_1m flat4_1m; do
timeout 200 python bench.py --workload $w --steps 60 --warmup 6 --no-cpu-baseline --no-end-to-end --stats-every 4 > $OUT/bench_$w.json 2> $OUT/b.err || tail -5 $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', round(d['ms_per_step'],4), 'sync', round(d['config']['sync_call_ms'],4), 'overlapped', round(d['overlapped']['ms_per_step'],4))"
done
