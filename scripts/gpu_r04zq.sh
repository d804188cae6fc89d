#!/bin/bash
# r04zq: Arrow -> Avro list bodies pipelined to depth two (the next item's BYTES requested an iteration ahead as well) against depth one (r04zg) and the loop of round 3
# offsets requested with the row's string fetches; GPU encode tests, then A/B at 2M and 10M rows inside one call
OUT=gpurun_out/r04zq; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_encode.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for r in a b; do
for v in "depth2:X=1" "depth1:RUHVRO_HIP_ENC_DEPTH=1" "old:RUHVRO_HIP_NO_ENC_PIPE=1"; do
  name=${v%%:*}; kv=${v#*:}
  for rows in 2000000 10000000; do
    ( export $kv; timeout 300 python bench.py --direction encode --rows $rows --steps 8 --warmup 2 > $OUT/bench_${name}_${rows}_$r.json 2> $OUT/bench_${name}_${rows}_$r.err )
    python -c "
import json; d=json.load(open('$OUT/bench_${name}_${rows}_$r.json')); r=d['roofline']; print('%-8s %9d rows  ms/step %.4f  %s  emit frac %.4f' % ('$name', $rows, d['ms_per_step'], {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}, r['frac']))"
  done
done
done 2>&1 | tee $OUT/summary.txt
