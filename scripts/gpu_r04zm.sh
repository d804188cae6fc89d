#!/bin/bash
# r04zm: persistent size pass with the next tile's window held in REGISTERS while the current one is walked out of LDS (spec_size_p)
OUT=gpurun_out/r04zm; mkdir -p $OUT; export TMPDIR=/tmp
RUHVRO_HIP_PERSIST=1 timeout 300 python scripts/parity_quick.py 1000000 > $OUT/parity_persist.log 2>&1; echo "parity persist rc=$?"; tail -2 $OUT/parity_persist.log
RUHVRO_HIP_PERSIST=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest persist rc=$?"; tail -3 $OUT/pytest.log
STEPS=20 bash scripts/gpu_env_ab.sh r04zm "base_a:" "persist_a:RUHVRO_HIP_PERSIST=1" "p768_a:RUHVRO_HIP_PERSIST=1 RUHVRO_HIP_PERSIST_GRID=768" "p2048_a:RUHVRO_HIP_PERSIST=1 RUHVRO_HIP_PERSIST_GRID=2048" "base_b:" "persist_b:RUHVRO_HIP_PERSIST=1" "p768_b:RUHVRO_HIP_PERSIST=1 RUHVRO_HIP_PERSIST_GRID=768" "p2048_b:RUHVRO_HIP_PERSIST=1 RUHVRO_HIP_PERSIST_GRID=2048"
