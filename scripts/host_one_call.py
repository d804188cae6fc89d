"""Six deserialize_array_threaded calls of [records] records (results dropped): the workload of a copy / kernel trace
(rocprofv3 --kernel-trace --memory-copy-trace) of the host path's pipeline."""
import sys, time
sys.path.insert(0, '.')
import torch
import pyruhvro_amd as P
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
data, offsets = fastgen.generate("full", n)
recs = fastgen.split(data, offsets)
for i in range(6):
    t = time.perf_counter(); out = P.deserialize_array_threaded(recs, SCHEMAS["full"], 8); w = time.perf_counter() - t
    print(f"call {i}: {w * 1e3:.3f} ms", flush=True)
    del out
    time.sleep(0.05)
