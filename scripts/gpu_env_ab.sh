#!/bin/bash
# Bench-only A/B of engine knobs / kernel variants given as env assignments, + a buffer-identity check of each.
# Usage: bash scripts/gpu_env_ab.sh tag "NAME:VAR=val VAR2=val" ...        (NAME: with nothing after it = defaults)
TAG=${1:-envab}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( for kv in $envs; do export "$kv"; done
    timeout 300 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-end-to-end ${BENCH_ARGS:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err
    timeout 200 python scripts/parity_quick.py > $OUT/parity_$name.log 2>&1; echo $? > $OUT/parity_$name.rc )
  rc=$(cat $OUT/parity_$name.rc)
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    print("%-28s parity rc=$rc  ms/step=%.4f  %s  emit frac=%.3f lds=%s" % ("$name", d["ms_per_step"], {k: round(v, 4) for k, v in d["config"]["kernel_ms"].items()}, d["roofline"]["frac"], d["config"].get("emit_lds_bytes_per_workgroup")))
except Exception as e:
    print("$name", "parity rc=$rc", "bench failed:", e)
PY
done 2>&1 | tee $OUT/summary.txt
