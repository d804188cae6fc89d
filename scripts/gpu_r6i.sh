#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
O=gpurun_out/r6i.txt; : > $O; : > gpurun_out/r6i_err.log
run() { echo "== ${ENVV[*]} :: $*" >> $O; timeout 600 env "${ENVV[@]}" python scripts/giant_probe.py "$@" >> $O 2>> gpurun_out/r6i_err.log || echo "FAILED" >> $O; }
ENVV=(A=1); run 20000 1 9000
ENVV=(A=1); run 20000 1 90000
ENVV=(A=1); run 20000 1 2000000
ENVV=(RUHVRO_HIP_PROFILE=1); run 20000 1 90000
cat $O; grep "ruhvro_hip profile" gpurun_out/r6i_err.log | tail -2 | cut -c1-500
P="python scripts/workload_probe.py"
O2=gpurun_out/r6i.jsonl; : > $O2
runp() { echo "== $*" >&2; timeout 900 env "${ENVV[@]}" $P "$@" >> $O2 2>>gpurun_out/r6i_err.log || echo "{\"failed\": \"$*\"}" >> $O2; }
ENVV=(A=1); runp full_realistic 10000000 --parity-max 1000000
ENVV=(A=1); runp full_realistic_heavy 1000000 --parity-max 100000 --reps 5
ENVV=(A=1); runp full_skewed 10000000 --no-parity
ENVV=(RUHVRO_HIP_VARIANT=NODIRECT); runp full_skewed 10000000 --no-parity
ENVV=(A=1); runp wide200 1000000 --parity-max 100000
ENVV=(A=1); runp full 10000000 --no-parity
cat $O2 | cut -c1-600
timeout 1500 python -m pytest tests/test_round6.py tests/test_round5.py -m gpu -q -x -s -p no:cacheprovider -k "not wide_schema_kernels" > gpurun_out/r6i_round6.txt 2>&1
tail -8 gpurun_out/r6i_round6.txt | cut -c1-250; grep -h "giant record:" gpurun_out/r6i_round6.txt
