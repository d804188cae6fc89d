#!/bin/bash
# where does a 1M-record call spend its fixed cost: host stamps + device timeline
mkdir -p gpurun_out/r03e; export TMPDIR=/tmp
for w in full1m cfg3_1m flat4_1m; do
  python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-end-to-end > gpurun_out/r03e/bench_$w.json 2> gpurun_out/r03e/bench_$w.err
  RUHVRO_HIP_HOSTPROF=1 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end 2>&1 >/dev/null | grep -a hostprof | tail -3 > gpurun_out/r03e/hostprof_$w.txt
  timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/r03e/t_$w -o t -- python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r03e/trace_$w.log 2>&1
  for f in $(find gpurun_out/r03e/t_$w -name "*.db"); do python scripts/rocpd_timeline.py $f 24 > gpurun_out/r03e/timeline_$w.txt 2>&1; done
  rm -rf gpurun_out/r03e/t_$w
  python -c "
import json; d=json.load(open('gpurun_out/r03e/bench_$w.json')); print('$w', d['ms_per_step'], d['config']['kernel_ms'])"
  cat gpurun_out/r03e/hostprof_$w.txt; cat gpurun_out/r03e/timeline_$w.txt
done
