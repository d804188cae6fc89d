#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp RUHVRO_HIP_SKIP_WARM=1
O=gpurun_out/r6h.txt; : > $O
run() { echo "== ${ENVV[*]} :: $*" >> $O; timeout 600 env "${ENVV[@]}" python scripts/giant_probe.py "$@" >> $O 2>> gpurun_out/r6h_err.log || echo "FAILED" >> $O; }
ENVV=(A=1); run 20000 0 9000
ENVV=(A=1); run 20000 1 9000
ENVV=(A=1); run 20000 1 90000
ENVV=(A=1); run 20000 8 9000
ENVV=(RUHVRO_HIP_VARIANT=NOSCAN); run 20000 1 90000
ENVV=(RUHVRO_HIP_WIN_BYTES=8192); run 20000 1 90000
ENVV=(RUHVRO_HIP_WIN_BYTES=98304); run 20000 1 90000
ENVV=(RUHVRO_HIP_NO_DENSE=1); run 20000 1 90000
ENVV=(RUHVRO_HIP_PROFILE=1); run 20000 1 90000
cat $O; grep "ruhvro_hip profile" gpurun_out/r6h_err.log | tail -4 | cut -c1-500
timeout 900 python -m pytest tests/test_round5.py -m gpu -q -x -s -p no:cacheprovider -k "giant or first_large" > gpurun_out/r6h_giant.txt 2>&1
tail -5 gpurun_out/r6h_giant.txt | cut -c1-300; grep -h "giant record:" gpurun_out/r6h_giant.txt
