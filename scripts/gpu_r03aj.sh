#!/bin/bash
# r03aj: GPU timeline of pipelined 1M-record calls on the shipping tree (kernels in start order with the idle gap before each)
OUT=gpurun_out/r03aj; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --stats-every 1000"
for w in full1m cfg3_1m; do
  timeout 200 rocprofv3 --kernel-trace -d $OUT/p_$w -o t -- python bench.py --workload $w --steps 12 --warmup 3 $B > $OUT/p_$w.log 2>&1
  for f in $(find $OUT/p_$w -name "*.db"); do python scripts/rocpd_timeline.py $f 200 2>&1 | grep -v "copyBuffer\|fillBuffer" | sed -n 22,42p; done
done
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
