#!/bin/bash
# Arrow -> Avro GPU parity (tests/test_gpu_encode.py), a timing line and a rocprofv3 kernel summary; every step
# bounded so a hang cannot eat the GPU budget.   Usage: bash scripts/gpu_encode.sh [tag]
TAG=${1:-enc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_encode.py -q -m gpu 2>&1 | tail -40 | tee $OUT/encode_tests.log
timeout 200 python scripts/encode_bench.py 2>&1 | tail -4 | tee $OUT/encode_bench.json
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats -- python scripts/encode_bench.py > $OUT/prof_stats.log 2>&1; echo "rocprof rc=$?"
python scripts/rocpd_summary.py $(find $OUT/prof_stats -name "*.db" | head -1) > $OUT/rocprofv3_kernel_stats.txt 2>&1; head -9 $OUT/rocprofv3_kernel_stats.txt
