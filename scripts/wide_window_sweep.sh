#!/bin/bash
cd "$(dirname "$0")/.." 2>/dev/null; cd $GRAFT_REPO_ROOT 2>/dev/null
mkdir -p gpurun_out/r06_zz && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
O=gpurun_out/r06_zz; : > $O/probes.jsonl
for win in 6144 8192 10240 12288 0 24576 32768; do
  if [ $win = 0 ]; then env -u RUHVRO_HIP_WIN_BYTES timeout 600 python scripts/workload_probe.py wide200 1000000 --reps 10 --no-parity 2>/dev/null | grep "^{" >> $O/probes.jsonl
  else RUHVRO_HIP_WIN_BYTES=$win timeout 600 python scripts/workload_probe.py wide200 1000000 --reps 10 --no-parity 2>/dev/null | grep "^{" >> $O/probes.jsonl; fi
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_zz/probes.jsonl"):
    d=json.loads(l); print(d["workload"], d["env"].get("RUHVRO_HIP_WIN_BYTES"), d["lds_bytes"], d["kernel_ms"], round(d["path_frac"],4))
PY
