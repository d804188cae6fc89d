#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r06_t && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
O=gpurun_out/r06_t
: > $O/probes.jsonl
timeout 900 python scripts/workload_probe.py wide200 400000 --reps 10 --no-parity 2>/dev/null | grep "^{" >> $O/probes.jsonl
RUHVRO_HIP_TOPUP_EVERY=4 timeout 900 python scripts/workload_probe.py wide200 400000 --reps 10 --no-parity 2>/dev/null | grep "^{" >> $O/probes.jsonl
RUHVRO_HIP_TOPUP_EVERY=1 timeout 900 python scripts/workload_probe.py wide200 400000 --reps 10 --no-parity 2>/dev/null | grep "^{" >> $O/probes.jsonl
timeout 900 python scripts/workload_probe.py wide200 1000000 --reps 10 --parity-max 200000 2>/dev/null | grep "^{" >> $O/probes.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r06_t/probes.jsonl"):
    d=json.loads(l); print(d["workload"], d["records"], d["env"].get("RUHVRO_HIP_TOPUP_EVERY"), d["kernel_ms"], round(d["path_frac"],4), d.get("parity_check",{}).get("result"))
PY
timeout 1500 python -m pytest tests/test_round6.py -q -x -k "wide or deep or slide" 2>&1 | tail -5
