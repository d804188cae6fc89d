#!/bin/bash
# round 6, first measurement: the walks off their tuned value distribution (before any kernel change)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
P="python scripts/workload_probe.py"
O=gpurun_out/r6a.jsonl; : > $O
run() { echo "== $*" >&2; timeout 600 env "${ENVV[@]}" $P "$@" >> $O 2>gpurun_out/r6a_err.log || echo "{\"failed\": \"$*\"}" >> $O; }
ENVV=(A=1); run full 10000000 --parity-max 1000000
ENVV=(A=1); run full_realistic 10000000 --parity-max 1000000
ENVV=(A=1); run full_skewed 10000000 --parity-max 1000000
ENVV=(RUHVRO_HIP_NO_TRUST=1); run full 10000000 --no-parity
ENVV=(RUHVRO_HIP_WIN_BYTES=0); run full 10000000 --no-parity
ENVV=(RUHVRO_HIP_WIN_BYTES=0); run full_skewed 10000000 --no-parity
ENVV=(RUHVRO_HIP_WIN_BYTES=40960); run full_skewed 10000000 --no-parity
ENVV=(RUHVRO_HIP_WIN_BYTES=0); run full_realistic 10000000 --no-parity
ENVV=(A=1); run full 10000000 --kernel generic --parity-max 1000000 --reps 8
ENVV=(A=1); run cfg3 1000000 --kernel generic
ENVV=(A=1); run cfg3 1000000
cat $O
