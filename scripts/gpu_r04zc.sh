#!/bin/bash
# r04zc: where a 1M-record call through the Python surface spends its 8-10 ms (engine timeline + boundary phases)
OUT=gpurun_out/r04zc; mkdir -p $OUT; export TMPDIR=/tmp
RUHVRO_HIP_TIMELINE=1 timeout 300 python scripts/py_surface_profile.py 1000000 > $OUT/py1m.txt 2>&1; awk '/wall_ms/{c++} c==4' $OUT/py1m.txt | head -40; grep wall_ms $OUT/py1m.txt
