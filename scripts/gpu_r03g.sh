#!/bin/bash
mkdir -p gpurun_out/r03g
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03g/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03g/pytest.log
tail -4 gpurun_out/r03g/pytest.log
bash scripts/gpu_ab.sh r03g "" "BITDECOMP" "" "BITDECOMP"
B="--no-cpu-baseline --no-end-to-end --stats-every 10"
for w in full1m cfg3_1m flat4_1m; do
  python bench.py --workload $w --steps 100 --warmup 5 $B > gpurun_out/r03g/bench_$w.json 2> gpurun_out/r03g/bench_$w.err
  python -c "
import json; d=json.load(open('gpurun_out/r03g/bench_$w.json')); print('$w', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()})"
done
