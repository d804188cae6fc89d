#!/bin/bash
# r04e: one stream, G groups back to back (size_g -> scan_g -> emit_g per group): does a group's second read of its input
# (and of its per-record counters) come out of the 256 MB Infinity Cache when the group is small enough?
OUT=gpurun_out/r04e; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-projection --no-other-configs --overlap-streams 0 --stats-every 1000"
for g in 0 2 4 8 0 8; do
  export RUHVRO_HIP_INTERNAL_STREAMS=1 RUHVRO_HIP_SPLIT_GROUPS=$g RUHVRO_HIP_SPLIT_STAGGER=0
  timeout 200 python bench.py --steps 30 --warmup 5 $B > $OUT/bench_g$g.json 2> $OUT/bench_g$g.err
  python -c "
import json; d=json.load(open('$OUT/bench_g$g.json')); print('groups=$g', 'ms/step', round(d['ms_per_step'],4), 'sync_call', round(d['config']['sync_call_ms'],4))"
done
export RUHVRO_HIP_SPLIT_GROUPS=8
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p8 -o t -- python bench.py --steps 6 --warmup 2 $B > $OUT/tl8.json 2> $OUT/tl8.err
for f in $(find $OUT/p8 -name "*.db"); do python scripts/rocpd_summary.py $f | head -8; python scripts/rocpd_timeline.py $f 40 > $OUT/timeline8.txt; done
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
tail -36 $OUT/timeline8.txt
