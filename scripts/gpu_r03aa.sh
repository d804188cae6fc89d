#!/bin/bash
# r03aa: the default bench line of the shipping tree (with the stamped traffic in place) + rocprofv3 kernel stats of the
# same command restricted to its single-stream region + the encode line with its stamped traffic
OUT=gpurun_out/r03aa; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_stats -o stats -- python bench.py --steps 20 --warmup 5 $B > $OUT/p_stats.log 2>&1; echo "stats rc=$?"
for f in $(find $OUT/p_stats -name "*.db"); do python scripts/rocpd_summary.py $f; done | grep -vE "^$" > $OUT/full10m_kernel_stats.txt; head -8 $OUT/full10m_kernel_stats.txt
timeout 300 python bench.py --direction encode --rows 2000000 --steps 5 --warmup 2 > $OUT/bench_encode_2m.json 2> $OUT/enc.err; echo "encode rc=$?"
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); r=d['roofline']; print(round(d['ms_per_step'],4), round(r['frac'],4), round(r['path_frac'],4), r['traffic'], r['read_frac'], d['overlapped']['ms_per_step'])
d=json.load(open('$OUT/bench_encode_2m.json')); print(d['roofline']['frac'], d['roofline']['traffic'])"
