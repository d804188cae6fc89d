#!/bin/bash
# r04ze: wave priority: a workgroup's prologue (bounds, window DMA issue [PRIO]; + scan and prefix [PRIOLONG]) at s_setprio 3, its walk at 0
OUT=gpurun_out/r04ze; mkdir -p $OUT; export TMPDIR=/tmp
bash scripts/gpu_ab.sh r04ze "" "WALKPRIO" "" "WALKPRIO" "" "WALKPRIO"
