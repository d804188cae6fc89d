#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r06_y && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
O=gpurun_out/r06_y
timeout 600 python scripts/giant_probe.py > $O/giant_probe.txt 2>&1; tail -2 $O/giant_probe.txt
timeout 1500 python -m pytest tests/test_round6.py tests/test_round5.py -q -x -s -k "slide or past_the_window or giant or workloads or 60" 2>&1 | grep -v "^$" | tail -12
timeout 600 python scripts/workload_probe.py full_realistic 10000000 --reps 10 --parity-max 2000000 2>/dev/null | grep "^{" > $O/probe.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r06_y/probe.jsonl"):
    d=json.loads(l); print(d["workload"], d["kernel_ms"], round(d["path_frac"],4), d.get("parity_check",{}).get("result"))
PY
