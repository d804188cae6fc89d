#!/bin/bash
mkdir -p gpurun_out/r03f
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03f/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03f/pytest.log
tail -5 gpurun_out/r03f/pytest.log
B="--no-cpu-baseline --no-end-to-end"
for w in full10m full1m cfg3_1m flat4_1m; do
  python bench.py --workload $w --steps 50 --warmup 5 $B > gpurun_out/r03f/bench_$w.json 2> gpurun_out/r03f/bench_$w.err
  python -c "
import json; d=json.load(open('gpurun_out/r03f/bench_$w.json')); print('$w', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}, 'emit frac', round(d['roofline']['frac'],3), 'path', round(d['roofline']['path_frac'],3)); print(json.dumps(d.get('config5_projection'))[:600])"
done
