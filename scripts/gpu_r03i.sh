#!/bin/bash
mkdir -p gpurun_out/r03i
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03i/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03i/pytest.log
tail -4 gpurun_out/r03i/pytest.log
B="--no-cpu-baseline --no-end-to-end"
python bench.py --steps 50 --warmup 5 $B > gpurun_out/r03i/bench_full10m.json 2> gpurun_out/r03i/bench_full10m.err
python -c "
import json; d=json.load(open('gpurun_out/r03i/bench_full10m.json')); print('full10m', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}, 'emit frac', round(d['roofline']['frac'],3)); print({g: (round(v['ms_per_step'],4), round(v['implied_efficiency'],3)) for g,v in d['config5_projection']['g'].items()})"
for w in full1m cfg3_1m flat4_1m; do
  python bench.py --workload $w --steps 100 --warmup 5 $B --stats-every 10 > gpurun_out/r03i/bench_$w.json 2> gpurun_out/r03i/bench_$w.err
  python -c "
import json; d=json.load(open('gpurun_out/r03i/bench_$w.json')); print('$w', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()})"
  RUHVRO_HIP_HOSTPROF=1 python bench.py --workload $w --steps 6 --warmup 2 $B --stats-every 100 2>&1 >/dev/null | grep -a hostprof | tail -2
done
