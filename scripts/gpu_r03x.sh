#!/bin/bash
OUT=gpurun_out/r03x; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 300 python scripts/k_lt_g_split.py 10000000 2>/dev/null | tee $OUT/k_lt_g_split.json
timeout 300 python scripts/k_lt_g_split.py 1000000 2>/dev/null | tee -a $OUT/k_lt_g_split.json
