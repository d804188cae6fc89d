#!/bin/bash
# Giant-record path after a change to spec_body.h item_scan: probe (4 arrays of 9,000 strings among 20,000 records), the
# realistic workload with its giant arrays, and every test that walks a record larger than the window.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/giant_check && export TMPDIR=/tmp
O=gpurun_out/giant_check
timeout 600 python scripts/giant_probe.py 2>/dev/null | tail -1 | tee $O/giant_probe.txt
timeout 600 python scripts/workload_probe.py full_realistic 10000000 --reps 10 --parity-max 2000000 2>/dev/null | grep "^{" | tee $O/full_realistic.json | cut -c1-400
timeout 1500 python -m pytest -m gpu tests/test_round6.py tests/test_round5.py -q -x -k "giant or slide or past or workloads or nested or item" 2>&1 | tail -4 | tee $O/tests.txt
