#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r06_r && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
O=gpurun_out/r06_r
: > $O/probes.jsonl
for w in full_realistic_nogiant full_skewed full_realistic; do
  timeout 600 python scripts/workload_probe.py $w 10000000 --reps 10 --parity-max 1000000 2>/dev/null | grep "^{" >> $O/probes.jsonl
done
RUHVRO_HIP_WIN_BYTES=16384 RUHVRO_HIP_RANGED=1 timeout 600 python scripts/workload_probe.py full 10000000 --reps 10 --parity-max 1000000 2>/dev/null | grep "^{" >> $O/probes.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r06_r/probes.jsonl"):
    d=json.loads(l); print(d["workload"], d["env"], d["kernel_ms"], round(d["path_frac"],4), d.get("parity_check",{}).get("result"), d["per_call"])
PY
timeout 1500 python -m pytest tests/test_round6.py -q -x 2>&1 | tail -5
