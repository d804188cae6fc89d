#!/bin/bash
# r04zi: Arrow -> Avro kernels with software-pipelined list bodies: traffic re-stamp (2M rows), kernel stats, bench lines at 2M / 10M rows
OUT=gpurun_out/r04zi; mkdir -p $OUT; export TMPDIR=/tmp
summ() { for f in $(find $1 -name "*.db"); do python scripts/rocpd_summary.py $f; done; }
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p_encf -o fetch -- python bench.py --direction encode --rows 2000000 --steps 3 --warmup 1 > $OUT/p_encf.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/p_encw -o write -- python bench.py --direction encode --rows 2000000 --steps 3 --warmup 1 > $OUT/p_encw.log 2>&1; echo "write rc=$?"
EKEY=$(python -c "from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS; print(cabi.kernel_key(SCHEMAS['full'], True))")
python scripts/rocpd_summary.py --traffic-json $(find $OUT/p_encf -name "*.db" | head -1) $(find $OUT/p_encw -name "*.db" | head -1) $EKEY > $OUT/encode_hbm_traffic.json
summ $OUT/p_encf | grep -E "FETCH_SIZE" > $OUT/encode_2m_fetch.txt; summ $OUT/p_encw | grep -E "WRITE_SIZE" > $OUT/encode_2m_write.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_enc -o stats -- python bench.py --direction encode --rows 2000000 --steps 5 --warmup 2 > $OUT/p_enc.log 2>&1
summ $OUT/p_enc | grep -vE "^$" | head -10 > $OUT/encode_kernel_stats.txt; head -5 $OUT/encode_kernel_stats.txt
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
cp $OUT/encode_hbm_traffic.json profiles/encode_hbm_traffic.json
for rows in 2000000 10000000; do
  timeout 300 python bench.py --direction encode --rows $rows --steps 8 --warmup 2 > $OUT/bench_encode_$rows.json 2> $OUT/bench_encode_$rows.err; echo "encode $rows rc=$?"
  python -c "
import json; d=json.load(open('$OUT/bench_encode_$rows.json')); r=d['roofline']; print(d['value'], round(d['ms_per_step'],4), d['config']['kernel_ms'], round(r['frac'],4), r['traffic'])"
done
cat $OUT/encode_hbm_traffic.json | head -30
