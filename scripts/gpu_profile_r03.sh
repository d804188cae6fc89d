#!/bin/bash
# Round-3 measurement pass on the MI355X box: the numbers and rocprofv3 summaries that go to profiles/.
# Every profiler pass is its own run (kernel-trace + stats, then one PMC counter per pass), each under `timeout`.
TAG=${1:-prof_r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0"   # profiler passes: single-stream launches only (kernels that share the chip have inflated durations by design)
summ() { for f in $(find $1 -name "*.db"); do python scripts/rocpd_summary.py $f; done; }
# A. the default bench line (what the driver runs)
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
# B. kernel stats of the same command
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_stats -o stats -- python bench.py --steps 10 --warmup 2 $B > $OUT/p_stats.log 2>&1; echo "stats rc=$?"
summ $OUT/p_stats | grep -vE "^$" > $OUT/full10m_kernel_stats.txt; head -8 $OUT/full10m_kernel_stats.txt
# C. HBM traffic (separate PMC passes), stamped with the kernel content hash
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p_fetch -o fetch -- python bench.py --steps 3 --warmup 1 $B > $OUT/p_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/p_write -o write -- python bench.py --steps 3 --warmup 1 $B > $OUT/p_write.log 2>&1; echo "write rc=$?"
KEY=$(python -c "from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS; print(cabi.kernel_key(SCHEMAS['full']))")
python scripts/rocpd_summary.py --traffic-json $(find $OUT/p_fetch -name "*.db" | head -1) $(find $OUT/p_write -name "*.db" | head -1) $KEY > $OUT/hbm_traffic.json; head -c 900 $OUT/hbm_traffic.json
summ $OUT/p_fetch | grep -E "FETCH_SIZE" > $OUT/full10m_fetch.txt; summ $OUT/p_write | grep -E "WRITE_SIZE" > $OUT/full10m_write.txt
# D. BASELINE configs 2 and 3 (cache-resident at 1M records: flagged in profiles/README.md)
for w in flat4_1m cfg3_1m full1m; do
  timeout 300 python bench.py --workload $w $B > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_$w -o stats -- python bench.py --workload $w --steps 20 --warmup 3 $B > $OUT/p_$w.log 2>&1
  summ $OUT/p_$w | grep -vE "^$" | head -8 > $OUT/${w}_kernel_stats.txt
done
# E. the other direction
for rows in 2000000 10000000; do
  timeout 300 python bench.py --direction encode --rows $rows --steps 5 --warmup 2 > $OUT/bench_encode_$rows.json 2> $OUT/bench_encode_$rows.err; echo "encode $rows rc=$?"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_enc -o stats -- python bench.py --direction encode --rows 2000000 --steps 5 --warmup 2 > $OUT/p_enc.log 2>&1
summ $OUT/p_enc | grep -vE "^$" | head -10 > $OUT/encode_kernel_stats.txt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p_encf -o fetch -- python bench.py --direction encode --rows 2000000 --steps 3 --warmup 1 > $OUT/p_encf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/p_encw -o write -- python bench.py --direction encode --rows 2000000 --steps 3 --warmup 1 > $OUT/p_encw.log 2>&1
EKEY=$(python -c "from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS; print(cabi.kernel_key(SCHEMAS['full'], True))")
python scripts/rocpd_summary.py --traffic-json $(find $OUT/p_encf -name "*.db" | head -1) $(find $OUT/p_encw -name "*.db" | head -1) $EKEY > $OUT/encode_hbm_traffic.json
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
ls $OUT
