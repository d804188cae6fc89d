#!/bin/bash
# r04zo: per-record counters handed from the size pass to the emit pass as ONE byte each where a wavefront's counters allow it
# (12 instead of 24 bytes per record on the benchmark schema); RH_V_NOLC8 = 16 bits always (r04x)
OUT=gpurun_out/r04zo; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_async_device.py tests/test_round4.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash scripts/gpu_ab.sh r04zo "" "NOLC8" "" "NOLC8" "" "NOLC8"
