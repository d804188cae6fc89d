#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
timeout 1500 python -m pytest tests/test_round6.py -q -x -k "lane_windows or wide_schema_mostly" 2>&1 | tail -15
