#!/bin/bash
# r04h: LDS-DMA staging as the default: window-size / long-record tests (rows beyond 64 KB of LDS), the default bench line (parity_check,
# Python surface at 1M / 10M, dist absent), gather scaling on the box's host
OUT=gpurun_out/r04h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_specialize.py tests/test_n4_types.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -2 $OUT/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04h/bench_default.json"))
print("ms/step", d["ms_per_step"], d["config"]["kernel_ms"], "frac", d["roofline"]["frac"], "path", d["roofline"]["path_frac"])
print("parity_check", d.get("parity_check"))
print("cpu", d["cpu_baseline"])
e=d["end_to_end"]
for k in ("packed_pageable","record_slices","packed_8_logical_shards"): print(k, round(e[k]["value"]/1e6,1), "M rec/s", e[k]["stage_ms"])
print(json.dumps(e["python_list_bytes"], indent=0))
print(e["config1_python_10k"])
print({k: (round(v["ms_per_step"],4), round(v.get("emit_frac",0),3)) for k,v in d["other_configs"].items()})
PY
timeout 600 python scripts/gather_scaling.py 10000000 5 > $OUT/gather_scaling.json 2> $OUT/gather.err; echo "gather rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r04h/gather_scaling.json')); print(d['host_cpus']); print(json.dumps(d['best'], indent=0))"
