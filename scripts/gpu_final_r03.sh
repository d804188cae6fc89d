#!/bin/bash
# Final confirmation pass of round 3 on the shipping tree: GPU suite, smoke(), the default bench line, kernel stats of its
# single-stream region (the traffic stamp is unchanged: same kernels as profiles/r03ah_*)
OUT=gpurun_out/r03al; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_stats -o stats -- python bench.py --steps 20 --warmup 5 $B > $OUT/p_stats.log 2>&1; echo "stats rc=$?"
for f in $(find $OUT/p_stats -name "*.db"); do python scripts/rocpd_summary.py $f; done | grep -vE "^$" > $OUT/full10m_kernel_stats.txt; head -6 $OUT/full10m_kernel_stats.txt
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); r=d['roofline']; print(round(d['ms_per_step'],4), d['value'], d['config']['kernel_ms'], round(r['frac'],4), round(r['path_frac'],4), r['traffic'], r['read_frac'], d['config']['sync_call_ms'])
print(d['overlapped']['ms_per_step'], d['overlapped']['value'], d['overlapped']['path_frac'])
print({g:(round(v['ms_per_step'],4), round(v['implied_efficiency'],3), round(v['overlapped']['ms_per_step'],4), round(v['overlapped']['implied_efficiency'],3)) for g,v in d['config5_projection']['g'].items()})
for k,v in d['other_configs'].items(): print(k, round(v['ms_per_step'],4), round(v.get('sync_call_ms',0),4), round(v.get('overlapped_ms_per_step',0),4), round(v['emit_frac'],3))
print({k:(round(v['value']/1e6,1), round(v['wall_ms'],2)) for k,v in d['end_to_end'].items() if isinstance(v,dict)})
print(d['cpu_baseline']['value'], d['cpu_baseline']['wide']['value'])"
