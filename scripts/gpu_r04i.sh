#!/bin/bash
# r04i: where the Python surface spends its time at 10M records: streaming hand-over (default) vs the classic path, engine timeline
OUT=gpurun_out/r04i; mkdir -p $OUT
RUHVRO_HIP_TIMELINE=1 timeout 300 python scripts/py_surface_profile.py 10000000 > $OUT/stream.txt 2> $OUT/stream.err; tail -3 $OUT/stream.txt; grep -i "timeline\|gathered\|h2d\|d2h" $OUT/stream.err | tail -40
PYRUHVRO_STREAM_MIN=-1 timeout 300 python scripts/py_surface_profile.py 10000000 > $OUT/classic.txt 2> $OUT/classic.err; tail -3 $OUT/classic.txt
