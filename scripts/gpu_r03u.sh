#!/bin/bash
# r03u: BASELINE config 1 through the Python surface (10k records, list[bytes] -> RecordBatches), stage by stage
timeout 300 python - <<'PY'
import time, os, sys
sys.path.insert(0, '.')
import torch
import pyruhvro_amd as P
from avrogen import synth
from avrogen.schemas import SCHEMAS
recs = synth.records("full", 10000)
S = SCHEMAS["full"]
for mode in ("auto", "specialized", "generic"):
    P.set_kernel_mode(mode)
    for _ in range(20): P.deserialize_array_threaded(recs, S, 8)
    t = time.perf_counter()
    for _ in range(200): out = P.deserialize_array_threaded(recs, S, 8)
    ms = (time.perf_counter() - t) / 200 * 1e3
    _, st = P.deserialize_array_threaded_with_stats(recs, S, 8)
    print(mode, "ms/call", round(ms, 4), {k: round(float(v), 4) for k, v in st.items() if k.endswith("_ms")}, "spec", st["specialized"])
P.set_kernel_mode("auto")
os.environ["RUHVRO_HIP_HOSTPROF"] = "1"
PY
RUHVRO_HIP_HOSTPROF=1 timeout 100 python - 2>&1 <<'PY' | tail -4
import sys
sys.path.insert(0, '.')
import torch
import pyruhvro_amd as P
from avrogen import synth
from avrogen.schemas import SCHEMAS
recs = synth.records("full", 10000)
for _ in range(6): P.deserialize_array_threaded(recs, SCHEMAS["full"], 8)
PY
