#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench line, rocprofv3 kernel stats + HBM PMC counters (separate passes),
# end-to-end (PCIe-inclusive) stage timings.   Usage (repo root, on the GPU box): bash scripts/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end > $OUT/prof_stats.log 2>&1; echo "rocprof stats rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/prof_fetch.log 2>&1; echo "rocprof fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -o write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/prof_write.log 2>&1; echo "rocprof write rc=$?"
timeout 300 python - > $OUT/e2e.log 2>&1 <<'PY'
# PCIe-inclusive path: host buffers in, host Arrow buffers out (rh_decode_packed), per-stage rh_stats
import json, time, torch
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
from pyruhvro_amd import cabi
for name, n in (("full", 10_000_000), ("full", 1_000_000)):
    data, offsets = fastgen.generate(name, n)
    for rep in range(3):
        t = time.perf_counter()
        out, st = cabi.decode_packed(data, offsets, SCHEMAS[name], 8, want_stats=True)
        wall = time.perf_counter() - t
        del out
    st["wall_ms"] = wall * 1e3
    st["records_per_s_end_to_end"] = n / wall
    print(json.dumps({"workload": f"{name} x {n}, rh_decode_packed (H2D + kernels + D2H), third call", **st}))
PY
cat $OUT/e2e.log | cut -c1-600
