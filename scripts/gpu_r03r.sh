#!/bin/bash
# r03r: RH_ASYNC: tests, then every workload with pipelined and with synchronous calls
OUT=gpurun_out/r03r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_async_device.py tests/test_engine_branches.py tests/test_device_export.py -x -q -m gpu > $OUT/pytest_async.log 2>&1; echo "async tests rc=$?"; tail -15 $OUT/pytest_async.log
B="--no-cpu-baseline --no-end-to-end"
for w in full10m full1m cfg3_1m flat4_1m; do
  for mode in "" "--sync-calls"; do
  timeout 200 python bench.py --workload $w --steps 50 --warmup 5 $B $mode > $OUT/bench_$w$mode.json 2> $OUT/bench_$w$mode.err || tail -5 $OUT/bench_$w$mode.err
  python -c "
import json; d=json.load(open('$OUT/bench_$w$mode.json')); print('$w $mode', round(d['ms_per_step'],4), 'sync_call_ms', round(d['config'].get('sync_call_ms') or 0,4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}, 'emit frac', round(d['roofline']['frac'],3), 'path', round(d['roofline'].get('path_frac',0),3)); p=d.get('config5_projection'); print({g: (round(v['ms_per_step'],4), round(v['implied_efficiency'],3)) for g,v in p['g'].items()} if p else '')"
  done
done
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
