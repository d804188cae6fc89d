import time, sys, os
sys.path.insert(0, '.')
import torch
import pyruhvro_amd as P
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
data, offsets = fastgen.generate("full", n)
recs = fastgen.split(data, offsets)
S = SCHEMAS["full"]
for _ in range(2): P.deserialize_array_threaded(recs, S, 8)
for _ in range(3):
    t = time.perf_counter(); out, st = P.deserialize_array_threaded_with_stats(recs, S, 8); w = time.perf_counter() - t
    print("wall_ms", round(w*1e3, 2), {k: round(float(v), 2) for k, v in st.items() if k.endswith("_ms")}, {k: round(v, 2) for k, v in P.last_decode_profile().items()}, flush=True)
    del out
