#!/bin/bash
# round 6: hot / ranged kernel split, wide schemas -- round-6 tests, probes, then the whole suite
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_round6.py tests/test_round5.py -m gpu -q -x -p no:cacheprovider -k "not wide_schemas" > gpurun_out/r6e_round6.txt 2>&1
tail -15 gpurun_out/r6e_round6.txt
timeout 900 python -X faulthandler -m pytest tests/test_round6.py -m gpu -q -x -p no:cacheprovider -k "wide_schemas and generic" > gpurun_out/r6e_wide_generic.txt 2>&1
head -60 gpurun_out/r6e_wide_generic.txt | cut -c1-200; tail -5 gpurun_out/r6e_wide_generic.txt
timeout 900 python -X faulthandler -m pytest tests/test_round6.py -m gpu -q -x -p no:cacheprovider -k "wide_schemas and specialized" > gpurun_out/r6e_wide_spec.txt 2>&1
head -60 gpurun_out/r6e_wide_spec.txt | cut -c1-200; tail -5 gpurun_out/r6e_wide_spec.txt
P="python scripts/workload_probe.py"
O=gpurun_out/r6e.jsonl; : > $O
run() { echo "== $*" >&2; timeout 900 env "${ENVV[@]}" $P "$@" >> $O 2>gpurun_out/r6e_err.log || echo "{\"failed\": \"$*\"}" >> $O; }
ENVV=(A=1); run full 10000000 --parity-max 1000000
ENVV=(A=1); run full 1000000 --no-parity
ENVV=(A=1); run cfg3 1000000 --no-parity
ENVV=(RUHVRO_HIP_RANGED=1); run full 10000000 --no-parity
ENVV=(A=1); run full_skewed 10000000 --parity-max 1000000
ENVV=(RUHVRO_HIP_WIN_BYTES=40960); run full_skewed 10000000 --no-parity
ENVV=(A=1); run full_realistic 10000000 --parity-max 1000000
ENVV=(A=1); run wide200 1000000 --parity-max 200000
ENVV=(A=1); run wide200 1000000 --kernel generic --no-parity --reps 5
cat $O
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_round6.py --deselect tests/test_round5.py > gpurun_out/r6e_suite.txt 2>&1
tail -40 gpurun_out/r6e_suite.txt
