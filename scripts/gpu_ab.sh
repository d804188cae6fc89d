#!/bin/bash
# Bench-only A/B of kernel variants (RUHVRO_HIP_VARIANT) on the full schema + a buffer-identity check of each variant
# against the oracle at 300k records.  Usage: bash scripts/gpu_ab.sh tag "V1" "V2,V3" ...   ("" = default)
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in "$@"; do
  name=${v:-default}; name=${name//,/+}
  export RUHVRO_HIP_VARIANT=$v
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  timeout 200 python scripts/parity_quick.py > $OUT/parity_$name.log 2>&1; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    print("%-70s parity rc=$rc  rec/s=%.3e  %s  emit frac=%.3f" % ("$name", d["value"], {k: round(v, 4) for k, v in d["config"]["kernel_ms"].items()}, d["roofline"]["frac"]))
except Exception as e:
    print("$name", "parity rc=$rc", "bench failed:", e)
PY
done 2>&1 | tee $OUT/summary.txt
