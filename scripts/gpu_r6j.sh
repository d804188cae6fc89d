#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out/r6j.txt; : > $O; : > gpurun_out/r6j_err.log
run() { echo "== ${ENVV[*]} :: $*" >> $O; timeout 600 env "${ENVV[@]}" python scripts/giant_probe.py "$@" >> $O 2>> gpurun_out/r6j_err.log || echo "FAILED" >> $O; }
ENVV=(A=1); run 20000 1 90000
ENVV=(A=1); run 20000 1 2000000
cat $O
P="python scripts/workload_probe.py"
O2=gpurun_out/r6j.jsonl; : > $O2
runp() { echo "== $*" >&2; timeout 900 env "${ENVV[@]}" $P "$@" >> $O2 2>>gpurun_out/r6j_err.log || echo "{\"failed\": \"$*\"}" >> $O2; }
ENVV=(A=1); runp full_realistic_nogiant 10000000 --parity-max 1000000
ENVV=(A=1); runp full_realistic 10000000 --no-parity
ENVV=(A=1); runp full_skewed 10000000 --no-parity
ENVV=(A=1); runp wide200 1000000 --no-parity
ENVV=(A=1); runp full 10000000 --no-parity
cat $O2 | cut -c1-600
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r6j_suite.txt 2>&1
tail -25 gpurun_out/r6j_suite.txt | cut -c1-250; grep -h "giant record:" gpurun_out/r6j_suite.txt
