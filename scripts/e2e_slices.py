"""Host in -> host out from record slices (rh_decode), the PCIe-inclusive rate with the engine's stage timings.
    python scripts/e2e_slices.py [records] [reps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
from pyruhvro_amd import cabi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
data, offsets = fastgen.generate("full", n)
ptrs = (np.uint64(data.ctypes.data) + offsets[:-1]).astype(np.uint64)
lens = np.diff(offsets).astype(np.uint64)
walls = []
for _ in range(reps):
    t = time.perf_counter()
    res, st = cabi.decode_slices(ptrs, lens, SCHEMAS["full"], 8, want_stats=True)
    walls.append(time.perf_counter() - t)
    del res
keys = ("pack_ms", "h2d_ms", "size_kernel_ms", "emit_kernel_ms", "d2h_ms", "total_ms")
print(json.dumps({"records": n, "wall_ms": [round(w * 1e3, 2) for w in walls], "best_records_per_s": n / min(walls),
                  "stage_ms_last": {k: round(float(st[k]), 3) for k in keys}, "side_by_side": bool(os.environ.get("RUHVRO_HIP_PACK_SIDE_BY_SIDE"))}))
