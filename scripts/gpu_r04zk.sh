#!/bin/bash
# r04zk: the GPU suite with every pooled block handed out filled with 0xA5 (RUHVRO_HIP_POISON=1): nothing may lean on what a block holds
OUT=gpurun_out/r04zk; mkdir -p $OUT; export TMPDIR=/tmp
RUHVRO_HIP_POISON=1 timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest_poison.log 2>&1; echo "pytest POISON rc=$?"; tail -5 $OUT/pytest_poison.log
