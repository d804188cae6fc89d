#!/bin/bash
# round 6, K2: ranged tiles / sliding window / cooperative copies -- new tests first, then the suite, then the distributions
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_round6.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/r6c_round6.txt
cat gpurun_out/r6c_round6.txt
P="python scripts/workload_probe.py"
O=gpurun_out/r6c.jsonl; : > $O
run() { echo "== $*" >&2; timeout 600 env "${ENVV[@]}" $P "$@" >> $O 2>gpurun_out/r6c_err.log || echo "{\"failed\": \"$*\"}" >> $O; }
ENVV=(A=1); run full 10000000 --parity-max 1000000
ENVV=(RUHVRO_HIP_VARIANT=LEN16); run full 10000000 --no-parity
ENVV=(RUHVRO_HIP_VARIANT=INT28); run full 10000000 --no-parity
ENVV=(RUHVRO_HIP_VARIANT=LEN16,INT28); run full 10000000 --no-parity
ENVV=(A=1); run full 10000000 --no-parity
ENVV=(A=1); run full_realistic 10000000 --parity-max 1000000
ENVV=(RUHVRO_HIP_WIN_BYTES=40960); run full_realistic 10000000 --no-parity
ENVV=(A=1); run full_skewed 10000000 --parity-max 1000000
ENVV=(RUHVRO_HIP_WIN_BYTES=40960); run full_skewed 10000000 --no-parity
ENVV=(RUHVRO_HIP_WIN_BYTES=65536); run full_skewed 10000000 --no-parity
ENVV=(A=1); run full_realistic_heavy 1000000 --parity-max 200000
cat $O
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_round6.py 2>&1 | tail -40 > gpurun_out/r6c_suite.txt
cat gpurun_out/r6c_suite.txt
