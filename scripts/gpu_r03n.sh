#!/bin/bash
# r03n: state of the tree after re-entry: full GPU suite, default bench line, per-workload lines, kernel stats
OUT=gpurun_out/r03n; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
B="--no-cpu-baseline --no-end-to-end"
for w in full10m full1m cfg3_1m flat4_1m; do
  timeout 200 python bench.py --workload $w --steps 50 --warmup 5 $B > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}, 'emit frac', round(d['roofline']['frac'],3), 'path', round(d['roofline'].get('path_frac',0),3)); print(json.dumps(d.get('config5_projection'))[:700])"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_stats -o stats -- python bench.py --steps 10 --warmup 2 $B > $OUT/p_stats.log 2>&1; echo "stats rc=$?"
for f in $(find $OUT/p_stats -name "*.db"); do python scripts/rocpd_summary.py $f; done | grep -vE "^$" > $OUT/full10m_kernel_stats.txt; head -8 $OUT/full10m_kernel_stats.txt
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
