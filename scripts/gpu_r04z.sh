#!/bin/bash
# r04z: field-split emit (spec_body.h spec_emit2: two wavefronts per 64 records, one per half of the record's top-level fields)
# parity of the GPU suite's core, then A/B against the one-wave emit kernel of the same code object (RUHVRO_HIP_SPLIT_EMIT=0)
OUT=gpurun_out/r04z; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python scripts/parity_quick.py > $OUT/parity_quick.log 2>&1; echo "parity_quick rc=$?"; tail -2 $OUT/parity_quick.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_async_device.py tests/test_round4.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
STEPS=20 bash scripts/gpu_env_ab.sh r04z "split:" "one:RUHVRO_HIP_SPLIT_EMIT=0" "split_b:" "one_b:RUHVRO_HIP_SPLIT_EMIT=0" "split_c:" "one_c:RUHVRO_HIP_SPLIT_EMIT=0"
