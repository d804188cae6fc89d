#!/bin/bash
# r04j: the CPython boundary publishes its extraction progress while it extracts: Python surface at 1M / 10M, parity of the streaming path
OUT=gpurun_out/r04j; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_multi_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
PYRUHVRO_STREAM_MIN=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_forced.log 2>&1; echo "pytest (streaming forced on every call) rc=$?"; tail -2 $OUT/pytest_forced.log
timeout 300 python scripts/py_surface_profile.py 10000000 > $OUT/stream10m.txt 2> $OUT/stream10m.err; tail -3 $OUT/stream10m.txt
timeout 300 python scripts/py_surface_profile.py 1000000 > $OUT/stream1m.txt 2> $OUT/stream1m.err; tail -3 $OUT/stream1m.txt
RUHVRO_HIP_TIMELINE=1 timeout 300 python scripts/py_surface_profile.py 10000000 > $OUT/tl.txt 2> $OUT/tl.err; grep timeline $OUT/tl.err | tail -48 | head -12
