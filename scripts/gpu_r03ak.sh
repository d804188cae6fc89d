#!/bin/bash
# r03ak: streaming hand-over of list[bytes] (rh_opts.ready / gathered): the GPU suite with streaming forced on every call, then default, then timing
OUT=gpurun_out/r03ak; mkdir -p $OUT
PYRUHVRO_STREAM_MIN=1 timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_stream.log 2>&1; echo "pytest(stream all) rc=$?"; tail -3 $OUT/pytest_stream.log
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
echo "--- streaming (default)"; python scripts/pyprof_list_bytes.py 2>/dev/null
echo "--- serial (PYRUHVRO_STREAM_MIN=-1)"; PYRUHVRO_STREAM_MIN=-1 python scripts/pyprof_list_bytes.py 2>/dev/null
