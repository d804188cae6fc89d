#!/bin/bash
# just the rocprofv3 kernel-stats pass of the bench command (bounded)
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 75 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.log; echo "rocprof rc=$?"
python scripts/rocpd_summary.py $(find $OUT/prof_stats -name "*.db" | head -1) > $OUT/rocprofv3_kernel_stats.txt 2>&1; head -8 $OUT/rocprofv3_kernel_stats.txt
