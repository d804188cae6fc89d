#!/bin/bash
# r03o: item-dense list handling of the emit kernel: parity (full GPU suite) + A/B against RUHVRO_HIP_NO_DENSE=1
OUT=gpurun_out/r03o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python scripts/parity_quick.py > $OUT/parity.log 2>&1; echo "parity rc=$?"; tail -2 $OUT/parity.log
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
B="--no-cpu-baseline --no-end-to-end"
run() {  # name env
  env $2 timeout 200 python bench.py --steps 30 --warmup 5 $B > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python -c "
import json; d=json.load(open('$OUT/bench_$1.json')); print('%-12s' % '$1', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}, 'emit frac', round(d['roofline']['frac'],3), 'path', round(d['roofline'].get('path_frac',0),3), 'lds', d['config']['emit_lds_bytes_per_workgroup'])"
}
run dense A=1
run nodense RUHVRO_HIP_NO_DENSE=1
run dense2 A=1
run nodense2 RUHVRO_HIP_NO_DENSE=1
RUHVRO_HIP_PROFILE=1 timeout 200 python bench.py --steps 3 --warmup 1 $B 2>&1 >/dev/null | grep profile | tail -2
