#!/bin/bash
# r03ac: item offsets of list bodies preloaded behind e_list_begin (e_span_preload): parity (encode tests) + A/B against RUHVRO_HIP_NO_PRELOAD=1
OUT=gpurun_out/r03ac; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_n4_types.py tests/test_named_refs.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for rows in 2000000 10000000; do
for v in "A=1" "RUHVRO_HIP_NO_PRELOAD=1" "A=1"; do
  env $v timeout 300 python bench.py --direction encode --rows $rows --steps 6 --warmup 2 > $OUT/b.json 2> $OUT/b.err || tail -3 $OUT/b.err
  python -c "
import json; d=json.load(open('$OUT/b.json')); print('$rows $v', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['config']['kernel_ms'].items()}, 'frac', round(d['roofline']['frac'],3))"
done; done
