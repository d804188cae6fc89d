#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
: > gpurun_out/r6o.txt
timeout 1500 python -m pytest tests/test_round6.py -q -k "slide" 2>&1 | tail -12 >> gpurun_out/r6o.txt
cat gpurun_out/r6o.txt
