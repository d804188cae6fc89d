#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r06_s && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
O=gpurun_out/r06_s
: > $O/probes.jsonl
for w in full_realistic_nogiant full_skewed; do
  for win in 20480 28672 49152 77824; do
    RUHVRO_HIP_WIN_BYTES=$win timeout 600 python scripts/workload_probe.py $w 10000000 --reps 10 --no-parity 2>/dev/null | grep "^{" >> $O/probes.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_s/probes.jsonl"):
    d=json.loads(l); print(d["workload"], d["env"].get("RUHVRO_HIP_WIN_BYTES"), d["lds_bytes"], d["kernel_ms"], round(d["path_frac"],4))
PY
