#!/bin/bash
# r04c: in-call overlap, staggered form (size pass g+1 beside emit pass g): streams x groups sweep + kernel timelines
OUT=gpurun_out/r04c; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-projection --no-other-configs --overlap-streams 0 --stats-every 1000"
for cfg in "1 0 1" "2 0 1" "2 4 1" "2 8 1" "3 0 1" "3 8 1" "2 0 0" "2 8 0"; do
  set -- $cfg
  export RUHVRO_HIP_INTERNAL_STREAMS=$1 RUHVRO_HIP_SPLIT_GROUPS=$2 RUHVRO_HIP_SPLIT_STAGGER=$3
  timeout 200 python bench.py --steps 30 --warmup 5 $B > $OUT/bench_s$1_g$2_st$3.json 2> $OUT/bench_s$1_g$2_st$3.err
  python -c "
import json; d=json.load(open('$OUT/bench_s$1_g$2_st$3.json')); print('streams=$1 groups=$2 stagger=$3', 'ms/step', round(d['ms_per_step'],4), 'sync_call', round(d['config']['sync_call_ms'],4))"
done
for cfg in "2 8 1" "3 0 0"; do
  set -- $cfg
  export RUHVRO_HIP_INTERNAL_STREAMS=$1 RUHVRO_HIP_SPLIT_GROUPS=$2 RUHVRO_HIP_SPLIT_STAGGER=$3
  timeout 300 rocprofv3 --kernel-trace -d $OUT/p_$1_$2_$3 -o t -- python bench.py --steps 6 --warmup 2 $B > $OUT/tl_$1_$2_$3.json 2> $OUT/tl_$1_$2_$3.err
  for f in $(find $OUT/p_$1_$2_$3 -name "*.db"); do python scripts/rocpd_timeline.py $f 70 > $OUT/timeline_$1_$2_$3.txt; done
done
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
