#!/bin/bash
# A/B of the staged kernel variants on the GPU box.  Before sending the tree, prebuild the
# variant kernels HERE so the box only loads code objects:
#     for v in "" EMIT_TRUST STAGE_SWITCH STAGE_LANE_PRED PREFIX_SELECT COPY_FLAT NACC_LDS CUR_ABS EMIT_TRUST,STAGE_SWITCH,PREFIX_SELECT,COPY_FLAT; do
#         RUHVRO_HIP_VARIANT=$v python -m pyruhvro_amd.prebuild; done
# Usage on the box:  bash scripts/gpu_variants.sh [tag] [variant ...]     (default: each staged variant, then all)
TAG=${1:-variants}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
VARS=("$@")
if [ ${#VARS[@]} -eq 0 ]; then VARS=("" EMIT_TRUST STAGE_SWITCH STAGE_LANE_PRED PREFIX_SELECT COPY_FLAT NACC_LDS CUR_ABS "EMIT_TRUST,STAGE_SWITCH,PREFIX_SELECT,COPY_FLAT"); fi
for v in "${VARS[@]}"; do
  name=${v:-default}; name=${name//,/+}
  export RUHVRO_HIP_VARIANT=$v
  timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_$name.log 2>&1; rc=$?
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    print("%-60s pytest rc=$rc  rec/s=%.3e  %s  emit frac=%.3f" % ("$name", d["value"], d["config"]["kernel_ms"], d["roofline"]["frac"]))
except Exception as e:
    print("$name", "pytest rc=$rc", "bench failed:", e)
PY
done
