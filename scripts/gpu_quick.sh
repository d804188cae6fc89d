#!/bin/bash
# Fast iteration pass on the GPU box: parity tests, then kernel timings of every bench workload.
TAG=${1:-quick}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
for w in full10m cfg3_1m flat4_1m; do
 for kf in ${KERNELS:-specialized generic}; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --kernel $kf > $OUT/bench_$w.json 2> $OUT/bench_$w.err || tail -5 $OUT/bench_$w.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$w.json"))
    print("$w", "$kf", d["config"]["emit_lds_bytes_per_workgroup"], "rec/s=%.3e"%d["value"], d["config"]["kernel_ms"], "emit frac=%.3f"%d["roofline"]["frac"], "path GB/s=%.0f"%d["config"]["path_alg_GBps"])
except Exception as e:
    print("$w failed", e)
PY
 done
done
