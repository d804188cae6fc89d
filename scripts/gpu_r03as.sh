#!/bin/bash
# r03as: timeline of the per-rank step of config 5 at g = 8 (1.25M records, ONE chunk) on this one GPU: kernels and gaps
OUT=gpurun_out/r03as; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-projection --no-other-configs --overlap-streams 0 --stats-every 1000"
timeout 300 rocprofv3 --kernel-trace -d $OUT/p -o t -- python bench.py --records 1250000 --steps 12 --warmup 3 $B > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"
for f in $(find $OUT/p -name "*.db"); do python scripts/rocpd_timeline.py $f 40 > $OUT/timeline.txt; python scripts/rocpd_summary.py $f > $OUT/stats.txt; done
timeout 300 python bench.py --records 1250000 --steps 40 --warmup 5 $B > $OUT/bench_plain.json 2> $OUT/bench_plain.err
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
head -12 $OUT/stats.txt; tail -24 $OUT/timeline.txt
python -c "
import json; d=json.load(open('$OUT/bench_plain.json')); print(d['ms_per_step'], d['config'].get('sync_call_ms'), d['config']['kernel_ms'], d['config'].get('chunks'))"
