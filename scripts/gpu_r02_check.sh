#!/bin/bash
# Round-2 iteration pass: GPU suite, every bench workload (specialised kernels), config 1 through the Python surface.
# RUHVRO_HIP_TWO_SYNC=1 reproduces round 1's host round trip between scan and emit for an A/B of the fused submission.
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
fi
for mode in 0 1; do
 for w in full10m full1m cfg3_1m flat4_1m; do
  RUHVRO_HIP_TWO_SYNC=$mode timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > $OUT/bench_${w}_ts$mode.json 2> $OUT/bench_${w}_ts$mode.err || tail -5 $OUT/bench_${w}_ts$mode.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_${w}_ts$mode.json"))
    print("two_sync=$mode %-9s rec/s=%.3e ms/step=%.4f kernels=%s emit frac=%.3f path GB/s=%.0f" % ("$w", d["value"], d["ms_per_step"], {k: round(v, 4) for k, v in d["config"]["kernel_ms"].items()}, d["roofline"]["frac"], d["config"]["path_alg_GBps"]))
except Exception as e:
    print("$w failed", e)
PY
 done
 RUHVRO_HIP_TWO_SYNC=$mode bash scripts/gpu_small.sh > $OUT/small_ts$mode.jsonl 2>&1; cut -c1-400 $OUT/small_ts$mode.jsonl
done
