#!/bin/bash
OUT=gpurun_out/r03t; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
bash scripts/gpu_profile_r03.sh prof_r03 2>&1 | tail -40
