#!/bin/bash
# r04zb: cache-policy hints: Arrow buffers written with non-temporal stores (NTST), the window's LDS-DMA rows read non-temporally (NTLD), both
OUT=gpurun_out/r04zb; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_ab.sh r04zb "" "NTST" "NTLD" "NTST,NTLD" "" "NTST" "NTLD" "NTST,NTLD"
