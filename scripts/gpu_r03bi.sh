#!/bin/bash
# r03bi: wave-private null counts of domain-0 fields, occupancy-aware window: whole GPU suite, smoke(), re-stamp of the HBM traffic, kernel stats, default line
OUT=gpurun_out/r03bi; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_n4_types.py tests/test_named_refs.py -m gpu -x -q > $OUT/pytest_n4.log 2>&1; echo "n4 rc=$?"; tail -3 $OUT/pytest_n4.log
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p_fetch -o fetch -- python bench.py --steps 3 --warmup 1 $B > $OUT/p_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/p_write -o write -- python bench.py --steps 3 --warmup 1 $B > $OUT/p_write.log 2>&1; echo "write rc=$?"
KEY=$(python -c "from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS; print(cabi.kernel_key(SCHEMAS['full']))")
python scripts/rocpd_summary.py --traffic-json $(find $OUT/p_fetch -name "*.db" | head -1) $(find $OUT/p_write -name "*.db" | head -1) $KEY > $OUT/hbm_traffic.json
for f in $(find $OUT/p_fetch -name "*.db"); do python scripts/rocpd_summary.py $f; done | grep -E "FETCH_SIZE" > $OUT/full10m_fetch.txt
for f in $(find $OUT/p_write -name "*.db"); do python scripts/rocpd_summary.py $f; done | grep -E "WRITE_SIZE" > $OUT/full10m_write.txt
cp $OUT/hbm_traffic.json profiles/hbm_traffic.json
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_stats -o stats -- python bench.py --steps 20 --warmup 5 $B > $OUT/p_stats.log 2>&1; echo "stats rc=$?"
for f in $(find $OUT/p_stats -name "*.db"); do python scripts/rocpd_summary.py $f; done | grep -vE "^$" > $OUT/full10m_kernel_stats.txt; head -6 $OUT/full10m_kernel_stats.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
python -c "
import json; d=json.load(open('$OUT/bench_default.json')); r=d['roofline']; print(round(d['ms_per_step'],4), d['value'], d['config']['kernel_ms'], round(r['frac'],4), round(r['path_frac'],4), r['traffic'], r['read_frac'], d['config']['sync_call_ms'])
print(d['overlapped']['ms_per_step'], d['overlapped']['value'], d['overlapped']['path_frac'])
print({g:(round(v['ms_per_step'],4), round(v['implied_efficiency'],3)) for g,v in d['config5_projection']['g'].items()})
for k,v in d['other_configs'].items(): print(k, round(v['ms_per_step'],4), round(v.get('sync_call_ms',0),4), round(v['emit_frac'],3))"
