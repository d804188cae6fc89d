#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
: > gpurun_out/r6o.txt
timeout 1500 python -m pytest tests/test_round6.py -q 2>&1 | tail -8 >> gpurun_out/r6o.txt
timeout 600 python scripts/giant_probe.py >> gpurun_out/r6o.txt 2>&1
cat gpurun_out/r6o.txt
