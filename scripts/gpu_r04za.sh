#!/bin/bash
# r04za: what the emit kernel's stores cost, timing-only builds (results are NOT valid: parity rc != 0 is expected for WRAP / NOST):
#   WRAP = every store's byte offset wrapped into the first 16 KiB of its buffer (320 buffers x 16 KiB stay in the L2s: same store
#          instructions, same TA / TCP work, no HBM writes)        NOST = no store instructions at all (values and addresses still computed)
OUT=gpurun_out/r04za; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_ab.sh r04za "" "WRAP" "NOST" "" "WRAP" "NOST"
