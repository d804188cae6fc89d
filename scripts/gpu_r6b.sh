#!/bin/bash
# round 6, K1 (wider single-read varint forms): parity suite + the distributions again
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -k "not wide97 and not wide200 and not wide400" -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r6b_suite.txt
P="python scripts/workload_probe.py"
O=gpurun_out/r6b.jsonl; : > $O
run() { echo "== $*" >&2; timeout 600 env "${ENVV[@]}" $P "$@" >> $O 2>gpurun_out/r6b_err.log || echo "{\"failed\": \"$*\"}" >> $O; }
ENVV=(A=1); run full 10000000 --parity-max 1000000
ENVV=(A=1); run full_realistic 10000000 --parity-max 1000000
ENVV=(A=1); run full 1000000
ENVV=(A=1); run cfg3 1000000
cat gpurun_out/r6b_suite.txt; cat $O
