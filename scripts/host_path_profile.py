"""Where the Python surface spends a call at the metric's small sizes (10k and 1M records of the full schema, num_chunks=8):
mean wall of the whole call, of the native half alone (list -> Arrow C structs, no pyarrow import), of the import alone, the
engine's host phases (RUHVRO_HIP_HOSTPROF=1 lines, stderr) and, for the 1M call, the stage timeline (RUHVRO_HIP_TIMELINE=1)."""
import os, sys, time
sys.path.insert(0, '.')
import torch
import pyarrow as pa
import pyruhvro_amd as P
from pyruhvro_amd import _pyruhvro as nat
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
S = SCHEMAS["full"]
comp = P._get_schema(S)


def native_only(recs, k):
    addrs, _ = nat.decode(comp.capsule, recs, k, -1, 0, False, 0, None)
    return addrs


for n, reps in ((10_000, 300), (1_000_000, 20)):
    data, offsets = fastgen.generate("full", n)
    recs = fastgen.split(data, offsets)
    for _ in range(10): P.deserialize_array_threaded(recs, S, 8)
    t = time.perf_counter()
    for _ in range(reps): P.deserialize_array_threaded(recs, S, 8)
    whole = (time.perf_counter() - t) / reps * 1e3
    t_nat = t_imp = 0.0
    for _ in range(reps):
        t = time.perf_counter()
        addrs = native_only(recs, 8)
        t_nat += time.perf_counter() - t
        t = time.perf_counter()
        out = [pa.RecordBatch._import_from_c(a, comp.arrow_schema) for a in addrs]
        for a in addrs: nat.free_struct(a)
        t_imp += time.perf_counter() - t
        t = time.perf_counter()
        del out
        t_del = time.perf_counter() - t
    print(f"n={n}: whole call {whole:.3f} ms; native half {t_nat / reps * 1e3:.3f} ms; pyarrow import of 8 batches {t_imp / reps * 1e3:.3f} ms; "
          f"dropping them {t_del * 1e3:.3f} ms", flush=True)
    for _ in range(2):
        out, st = P.deserialize_array_threaded_with_stats(recs, S, 8)
        print("   ", {k: round(float(v), 3) for k, v in st.items() if k.endswith("_ms")}, {k: round(v, 3) for k, v in P.last_decode_profile().items()}, flush=True)
    # the same records through the C ABI's packed / slices entry points
    from pyruhvro_amd import cabi
    import numpy as np
    ptrs = (np.uint64(data.ctypes.data) + offsets[:-1]).astype(np.uint64)
    lens = np.diff(offsets).astype(np.uint64)
    for name, f in (("rh_decode_packed", lambda: cabi.decode_packed(data, offsets, S, 8)), ("rh_decode (slices)", lambda: cabi.decode_slices(ptrs, lens, S, 8))):
        for _ in range(3): f()
        t = time.perf_counter()
        for _ in range(max(reps // 2, 5)): f()
        print(f"    {name}: {(time.perf_counter() - t) / max(reps // 2, 5) * 1e3:.3f} ms", flush=True)
