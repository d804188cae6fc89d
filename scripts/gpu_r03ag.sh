#!/bin/bash
# r03ag: null counts spread over 64 addresses per (node, chunk): GPU suite, then the workloads that carry nullable columns
OUT=gpurun_out/r03ag; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0"
for w in "cfg3_1m --records 10000000" "cfg3_1m" "full1m" "full10m" "flat4_1m"; do
  python bench.py --workload $w --steps 40 $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],4), 'sync', round(d['config']['sync_call_ms'],4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}, 'frac', round(d['roofline']['frac'],3))"
done
