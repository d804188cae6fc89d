#!/bin/bash
# BASELINE config 1 (10k records of the full schema, num_chunks=8) and the Python list[bytes] surface at 1M records
timeout 200 python - <<'PY'
import time, json
import pyruhvro_amd as P
from avrogen import synth, fastgen
from avrogen.schemas import SCHEMAS
recs = synth.records("full", 10_000)
for _ in range(5): P.deserialize_array_threaded(recs, SCHEMAS["full"], 8)
best = 1e9
for _ in range(50):
    t = time.perf_counter(); out = P.deserialize_array_threaded(recs, SCHEMAS["full"], 8); best = min(best, time.perf_counter() - t)
_, st = P.deserialize_array_threaded_with_stats(recs, SCHEMAS["full"], 8)
print(json.dumps({"config": "10k records, full schema, num_chunks=8, list[bytes] in -> list[RecordBatch] out (host memory)",
                  "best_ms": best * 1e3, "records_per_s": 10_000 / best, "stats": st}))
data, offsets = fastgen.generate("full", 1_000_000)
recs = fastgen.split(data, offsets)
for _ in range(2): P.deserialize_array_threaded(recs, SCHEMAS["full"], 8)
t = time.perf_counter(); out = P.deserialize_array_threaded(recs, SCHEMAS["full"], 8); dt = time.perf_counter() - t
_, st = P.deserialize_array_threaded_with_stats(recs, SCHEMAS["full"], 8)
print(json.dumps({"config": "1M records, full schema, list[bytes] surface", "ms": dt * 1e3, "records_per_s": 1e6 / dt, "stats": st}))
PY
