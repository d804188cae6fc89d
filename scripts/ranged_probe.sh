#!/bin/bash
# Workloads whose tiles go to the ranged kernel pair, one JSON line each (scripts/workload_probe.py), then the round-6 tests.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/ranged_probe && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
O=gpurun_out/ranged_probe; : > $O/probes.jsonl
for w in full_realistic full_realistic_nogiant full_skewed; do
  timeout 600 python scripts/workload_probe.py $w 10000000 --reps 10 --parity-max 2000000 2>/dev/null | grep "^{" >> $O/probes.jsonl
done
timeout 600 python scripts/workload_probe.py wide200 1000000 --reps 10 --parity-max 200000 2>/dev/null | grep "^{" >> $O/probes.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/ranged_probe/probes.jsonl"):
    d=json.loads(l); print(d["workload"], d["kernel_ms"], round(d["path_frac"],4), d.get("parity_check",{}).get("result"), d["per_call"])
PY
timeout 600 python scripts/giant_probe.py 2>/dev/null | tail -1
timeout 1500 python -m pytest tests/test_round6.py -q -x -k "slide or past or workloads or lane_windows" 2>&1 | tail -4
