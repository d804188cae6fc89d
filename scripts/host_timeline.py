"""RUHVRO_HIP_TIMELINE=1 python scripts/host_timeline.py [records]: the stage timeline (stderr) of the LAST of six calls through the
Python surface (list[bytes] -> RecordBatches), results dropped between calls, then of one call whose predecessor's result is kept."""
import os, sys, time
sys.path.insert(0, '.')
import torch
import pyruhvro_amd as P
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
data, offsets = fastgen.generate("full", n)
recs = fastgen.split(data, offsets)
S = SCHEMAS["full"]
for i in range(6):
    sys.stderr.write(f"=== call {i}\n"); sys.stderr.flush()
    t = time.perf_counter()
    out = P.deserialize_array_threaded(recs, S, 8)
    sys.stderr.write(f"=== call {i} wall {(time.perf_counter() - t) * 1e3:.3f} ms {P.last_decode_profile()}\n"); sys.stderr.flush()
    del out
keep = []
for i in range(4):
    sys.stderr.write(f"=== keep-alive call {i}\n"); sys.stderr.flush()
    t = time.perf_counter()
    keep.append(P.deserialize_array_threaded(recs, S, 8))
    sys.stderr.write(f"=== keep-alive call {i} wall {(time.perf_counter() - t) * 1e3:.3f} ms\n"); sys.stderr.flush()
