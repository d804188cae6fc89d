#!/bin/bash
# r04t: single-pass form as the default of device-resident calls: its tests, then the whole GPU suite
OUT=gpurun_out/r04t; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_single_pass.py -m gpu -x -q > $OUT/pytest_single.log 2>&1; echo "single rc=$?"; tail -15 $OUT/pytest_single.log
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "suite rc=$?"; tail -4 $OUT/pytest.log
