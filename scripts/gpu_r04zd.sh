#!/bin/bash
# r04zd: host-call pool / stream cache: the GPU tests that go through host calls, then the bench line's end_to_end block
OUT=gpurun_out/r04zd; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-projection --no-other-configs > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04zd/bench.json")); e=d["end_to_end"]
print({k: round(e[k]["value"]/1e6,1) for k in ("packed_pageable","record_slices","packed_8_logical_shards")})
for m,v in e["python_list_bytes"].items(): print(m, round(v["value"]/1e6,1), "M rec/s", v["wall_ms"], "gil", v["gil_held_ms"], "vs_record_slices", round(v["vs_record_slices"],3), v.get("phase_ms"))
print(e["config1_python_10k"])
PY
