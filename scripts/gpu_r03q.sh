#!/bin/bash
# r03q: where a small device-resident call spends its microseconds: host marks + kernel timeline
OUT=gpurun_out/r03q; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_hostprof.sh > $OUT/hostprof.log 2>&1; cat $OUT/hostprof.log
B="--no-cpu-baseline --no-end-to-end --no-projection"
for w in full1m cfg3_1m flat4_1m; do
  timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/p_$w -o t -- python bench.py --workload $w --steps 6 --warmup 2 $B > $OUT/p_$w.log 2>&1
  for f in $(find $OUT/p_$w -name "*.db"); do python scripts/rocpd_timeline.py $f 2>&1 | tail -24; done
done
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
