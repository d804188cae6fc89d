#!/bin/bash
mkdir -p gpurun_out/r03j; export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r03j/p -o s -- python bench.py --steps 20 --warmup 3 $B > gpurun_out/r03j/p.log 2>&1
for f in $(find gpurun_out/r03j/p -name "*.db"); do python scripts/rocpd_summary.py $f | head -12; python scripts/rocpd_timeline.py $f 30; done
rm -rf gpurun_out/r03j/p
python bench.py --steps 20 --warmup 3 $B | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('full10m', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['config']['kernel_ms'].items()})"
