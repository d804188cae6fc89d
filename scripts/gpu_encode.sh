#!/bin/bash
# Arrow -> Avro GPU parity (tests/test_gpu_encode.py) + a timing line; bounded so a hang cannot eat the budget
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_encode.py -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/encode_tests.log
timeout 200 python scripts/encode_bench.py 2>&1 | tail -5 | tee gpurun_out/encode_bench.log
