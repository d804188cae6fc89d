#!/bin/bash
# round 6: where do giant records / direct tiles spend their time?  (focused A/B, one box)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
P="python scripts/workload_probe.py"
O=gpurun_out/r6g.jsonl; : > $O
run() { echo "== $*" >&2; timeout 900 env "${ENVV[@]}" $P "$@" >> $O 2>gpurun_out/r6g_err.log || echo "{\"failed\": \"$*\"}" >> $O; }
ENVV=(A=1); run full_realistic_heavy 200000 --parity-max 50000 --reps 5
ENVV=(RUHVRO_HIP_VARIANT=NOSCAN); run full_realistic_heavy 200000 --no-parity --reps 5
ENVV=(RUHVRO_HIP_WIN_BYTES=98304); run full_realistic_heavy 200000 --no-parity --reps 5
ENVV=(RUHVRO_HIP_NO_DENSE=1); run full_realistic_heavy 200000 --no-parity --reps 5
ENVV=(RUHVRO_HIP_PROFILE=1); run full_realistic_heavy 200000 --no-parity --reps 3
ENVV=(A=1); run full_realistic 10000000 --no-parity
ENVV=(A=1); run full_skewed 10000000 --no-parity
ENVV=(A=1); run wide200 1000000 --no-parity
ENVV=(A=1); run full 10000000 --no-parity
ENVV=(A=1); run full 10000000 --kernel generic --no-parity --reps 8
cat $O | cut -c1-700
grep "ruhvro_hip profile" gpurun_out/r6g_err.log | tail -6
timeout 1200 python -m pytest tests/test_round6.py tests/test_round5.py -m gpu -q -x -s -p no:cacheprovider -k "not wide_schema_kernels" > gpurun_out/r6g_round6.txt 2>&1
tail -8 gpurun_out/r6g_round6.txt | cut -c1-250; grep -h "giant record:" gpurun_out/r6g_round6.txt
