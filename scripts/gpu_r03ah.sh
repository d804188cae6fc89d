#!/bin/bash
# r03ah: GPU suite (already run in r03ag) is skipped here: re-stamp of the HBM traffic after the null-count change + kernel stats + default line
OUT=gpurun_out/r03ah; mkdir -p $OUT; export TMPDIR=/tmp
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
import json; d=json.load(open('$OUT/bench_default.json')); r=d['roofline']; print(round(d['ms_per_step'],4), d['config']['kernel_ms'], round(r['frac'],4), round(r['path_frac'],4), r['traffic'], r['read_frac'], d['overlapped']['ms_per_step'], d['overlapped']['path_frac'])"
