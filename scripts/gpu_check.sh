#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench line, rocprofv3 kernel stats + HBM PMC counters.
# Usage (from the repo root on the GPU box): bash scripts/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_stats.log 2>&1; echo "rocprof stats rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1; echo "rocprof fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -o write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_write.log 2>&1; echo "rocprof write rc=$?"
find $OUT -name '*.csv' | head -30
ls -la $OUT/prof_stats | head
