#!/bin/bash
mkdir -p gpurun_out/r03k
B="--no-cpu-baseline --no-end-to-end"
run() { python bench.py --steps 50 --warmup 5 $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 full10m', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}, {g: (round(v['ms_per_step'],4), round(v['implied_efficiency'],3)) for g,v in d['config5_projection']['g'].items()})"
for w in full1m cfg3_1m flat4_1m; do python bench.py --workload $w --steps 100 --warmup 5 $B --stats-every 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 $w', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()})"; done; }
RUHVRO_HIP_SPIN_US=0 run nospin
run spin
HSA_ENABLE_INTERRUPT=0 RUHVRO_HIP_SPIN_US=0 run hsa_poll
RUHVRO_HIP_SPIN_US=0 run nospin
timeout 600 python -m pytest tests -m gpu -x -q -k "baseline_configs or chunk_semantics or generated_records or multi_gpu or engine" 2>&1 | tail -3
