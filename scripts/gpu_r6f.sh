#!/bin/bash
# round 6: item scan (giant lists), direct tiles, 40 KB window cap, generic kernels with fast walks + counter hand-over
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_round6.py tests/test_round5.py -m gpu -q -x -s -p no:cacheprovider > gpurun_out/r6f_round6.txt 2>&1
tail -15 gpurun_out/r6f_round6.txt | cut -c1-250; grep -h "giant record:" gpurun_out/r6f_round6.txt
P="python scripts/workload_probe.py"
O=gpurun_out/r6f.jsonl; : > $O
run() { echo "== $*" >&2; timeout 900 env "${ENVV[@]}" $P "$@" >> $O 2>gpurun_out/r6f_err.log || echo "{\"failed\": \"$*\"}" >> $O; }
ENVV=(A=1); run full 10000000 --parity-max 1000000
ENVV=(A=1); run full_realistic 10000000 --parity-max 1000000
ENVV=(A=1); run full_skewed 10000000 --parity-max 1000000
ENVV=(A=1); run wide200 1000000 --parity-max 200000
ENVV=(A=1); run full_realistic_heavy 1000000 --parity-max 200000
ENVV=(A=1); run full 10000000 --kernel generic --parity-max 1000000 --reps 8
ENVV=(A=1); run cfg3 1000000 --kernel generic
ENVV=(A=1); run wide200 1000000 --kernel generic --no-parity --reps 5
ENVV=(A=1); run full 1000000 --no-parity
cat $O
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_round6.py --deselect tests/test_round5.py > gpurun_out/r6f_suite.txt 2>&1
tail -30 gpurun_out/r6f_suite.txt | cut -c1-250
