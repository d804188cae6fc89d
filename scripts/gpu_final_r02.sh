OUT=gpurun_out/prof_r02k; mkdir -p $OUT; export TMPDIR=/tmp
summ() { for f in $(find $1 -name "*.db"); do python scripts/rocpd_summary.py $f; done; }
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for rows in 2000000 10000000; do timeout 300 python bench.py --direction encode --rows $rows --steps 5 --warmup 2 > $OUT/bench_encode_$rows.json 2> $OUT/bench_encode_$rows.err; echo "encode $rows rc=$?"; done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_enc -o stats -- python bench.py --direction encode --rows 2000000 --steps 5 --warmup 2 > $OUT/p_enc.log 2>&1
summ $OUT/p_enc | grep -vE "^$" | head -10 > $OUT/encode_kernel_stats.txt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p_encf -o fetch -- python bench.py --direction encode --rows 2000000 --steps 3 --warmup 1 > $OUT/p_encf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/p_encw -o write -- python bench.py --direction encode --rows 2000000 --steps 3 --warmup 1 > $OUT/p_encw.log 2>&1
EKEY=$(python -c "from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS; print(cabi.kernel_key(SCHEMAS['full'], True))")
python scripts/rocpd_summary.py --traffic-json $(find $OUT/p_encf -name "*.db" | head -1) $(find $OUT/p_encw -name "*.db" | head -1) $EKEY > $OUT/encode_hbm_traffic.json
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > $OUT/bench_decode_quick.json 2>/dev/null
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
