#!/bin/bash
# A/B of Arrow -> Avro kernel variants (RUHVRO_HIP_VARIANT) on the full schema: the encode bench line + a byte-identity
# check of each variant against the generator's datums.  Usage: bash scripts/gpu_ab_encode.sh tag "V1" "V2,V3" ...
TAG=${1:-abe}; shift
ROWS=${ROWS:-4000000}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in "$@"; do
  name=${v:-default}; name=${name//,/+}
  export RUHVRO_HIP_VARIANT=$v
  timeout 200 python bench.py --direction encode --rows $ROWS --steps 4 --warmup 2 > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  timeout 200 python scripts/parity_quick_encode.py > $OUT/parity_$name.log 2>&1; rc=$?
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    print("%-40s parity rc=$rc  %s  emit frac=%.3f" % ("$name", {k: round(v, 4) for k, v in d["config"]["kernel_ms"].items()}, d["roofline"]["frac"]))
except Exception as e:
    print("$name", "parity rc=$rc", "bench failed:", e)
PY
done 2>&1 | tee $OUT/summary.txt
