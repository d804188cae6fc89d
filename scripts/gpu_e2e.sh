#!/bin/bash
# PCIe-inclusive stage timings of rh_decode_packed (host buffers in, host Arrow buffers out)
timeout 300 python - <<'PY'
import json, time, torch
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
from pyruhvro_amd import cabi
for name, n in (("full", 10_000_000), ("full", 1_000_000)):
    data, offsets = fastgen.generate(name, n)
    for rep in range(4):
        t = time.perf_counter()
        out, st = cabi.decode_packed(data, offsets, SCHEMAS[name], 8, want_stats=True)
        wall = time.perf_counter() - t
        del out
    st["wall_ms"] = wall * 1e3
    st["records_per_s_end_to_end"] = n / wall
    print(json.dumps({"workload": f"{name} x {n}, rh_decode_packed (H2D + kernels + D2H), fourth call", **st}))
PY
