#!/bin/bash
# one optimisation iteration on the GPU box: whole GPU suite, then the headline workload's kernel times
TAG=${1:-iter}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > $OUT/bench.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("rec/s=%.3e"%d["value"], d["config"]["kernel_ms"], "emit frac=%.3f"%d["roofline"]["frac"])
PY
