#!/bin/bash
# round 6: sliding fix, A/B of the ranged / cooperative-copy code on the friendly workload, the two suite failures, wide schemas
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
timeout 1200 python -m pytest tests/test_round6.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/r6d_round6.txt
cat gpurun_out/r6d_round6.txt
timeout 900 python -m pytest "tests/test_single_pass.py::test_random_schemas_single_pass" -m gpu -q -x -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r6d_single.txt
cat gpurun_out/r6d_single.txt
timeout 900 python -m pytest tests/test_round4.py::test_poisoned_pools_both_directions -m gpu -q -x -p no:cacheprovider 2>&1 | tail -80 > gpurun_out/r6d_poison.txt
cat gpurun_out/r6d_poison.txt
P="python scripts/workload_probe.py"
O=gpurun_out/r6d.jsonl; : > $O
run() { echo "== $*" >&2; timeout 900 env "${ENVV[@]}" $P "$@" >> $O 2>gpurun_out/r6d_err.log || echo "{\"failed\": \"$*\"}" >> $O; }
ENVV=(A=1); run full 10000000 --no-parity
ENVV=(RUHVRO_HIP_VARIANT=NORANGED); run full 10000000 --no-parity
ENVV=(RUHVRO_HIP_VARIANT=NOCOOP); run full 10000000 --no-parity
ENVV=(RUHVRO_HIP_VARIANT=NORANGED,NOCOOP); run full 10000000 --no-parity
ENVV=(RUHVRO_HIP_VARIANT=NORANGED,NOCOOP,LEN16,INT28); run full 10000000 --no-parity
ENVV=(A=1); run full 10000000 --no-parity
ENVV=(RUHVRO_HIP_WIN_BYTES=40960); run full_skewed 10000000 --no-parity
ENVV=(RUHVRO_HIP_WIN_BYTES=24576); run full_skewed 10000000 --no-parity
ENVV=(RUHVRO_HIP_WIN_BYTES=40960); run full_realistic 10000000 --no-parity
cat $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "wide97 or wide200" 2>&1 | tail -30 > gpurun_out/r6d_wide.txt
cat gpurun_out/r6d_wide.txt
