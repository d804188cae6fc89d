#!/bin/bash
mkdir -p gpurun_out/r03l
timeout 900 python -m pytest tests/test_gpu_encode.py -m gpu -x -q > gpurun_out/r03l/pytest_encode.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03l/pytest_encode.log
for rows in 2000000 10000000; do
python bench.py --direction encode --rows $rows --steps 20 --warmup 3 > gpurun_out/r03l/bench_encode_$rows.json 2> gpurun_out/r03l/bench_encode_$rows.err; echo "encode $rows rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r03l/bench_encode_$rows.json')); print(round(d['value']/1e9,3),'G rows/s', round(d['ms_per_step'],4), d['config']['kernel_ms'], 'emit frac', round(d['roofline']['frac'],3), 'path', round(d['roofline']['path_frac'],3), 'e2e', round(d['end_to_end']['value']/1e6,1), 'M rows/s')"
done
tail -3 gpurun_out/r03l/bench_encode_2000000.err
