"""Where a 10,000-record call through the Python surface (BASELINE config 1) spends its ~0.27 ms: the boundary's phases
(PYRUHVRO_PYPROF) and the engine's host phases (RUHVRO_HIP_HOSTPROF) of a few calls, then the mean wall time of 300."""
import os, sys, time
sys.path.insert(0, '.')
import torch
import pyruhvro_amd as P
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
data, offsets = fastgen.generate("full", n)
recs = fastgen.split(data, offsets)
S = SCHEMAS["full"]
for _ in range(20): P.deserialize_array_threaded(recs, S, 8)
t = time.perf_counter()
for _ in range(300): P.deserialize_array_threaded(recs, S, 8)
print("mean wall ms", (time.perf_counter() - t) / 300 * 1e3, flush=True)
for _ in range(3):
    out, st = P.deserialize_array_threaded_with_stats(recs, S, 8)
    print({k: round(float(v), 3) for k, v in st.items() if k.endswith("_ms")}, {k: round(v, 3) for k, v in P.last_decode_profile().items()}, flush=True)
