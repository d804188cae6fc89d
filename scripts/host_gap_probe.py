"""Why is the wall time of deserialize_array_threaded larger than the boundary's own total?  Bench conditions (a 10M-element list
alive), per rep: wall, the boundary's total, gc collections during the call, and the same with the native half / import split."""
import gc, os, sys, time
sys.path.insert(0, '.')
import torch
import pyarrow as pa
import pyruhvro_amd as P
from pyruhvro_amd import _pyruhvro as nat
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
S = SCHEMAS["full"]
comp = P._get_schema(S)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
data, offsets = fastgen.generate("full", n)
recs = fastgen.split(data, offsets)
for m in (1_000_000, n):
    sub = recs if m == n else recs[:m]
    P.deserialize_array_threaded(sub, S, 8)
    for rep in range(4):
        g0 = [s["collections"] for s in gc.get_stats()]
        t = time.perf_counter()
        res = P.deserialize_array_threaded(sub, S, 8)
        w = time.perf_counter() - t
        g1 = [s["collections"] for s in gc.get_stats()]
        print(f"m={m} rep {rep}: wall {w * 1e3:.3f} ms, boundary total {P.last_decode_profile()['total_ms']:.3f} ms, gc collections {[b - a for a, b in zip(g0, g1)]}", flush=True)
        t = time.perf_counter(); del res; print(f"    del res {1e3 * (time.perf_counter() - t):.3f} ms", flush=True)
    for rep in range(3):
        t0 = time.perf_counter()
        addrs, _ = nat.decode(comp.capsule, sub, 8, -1, 0, False, 0, None)
        t1 = time.perf_counter()
        out = [pa.RecordBatch._import_from_c(a, comp.arrow_schema) for a in addrs]
        for a in addrs: nat.free_struct(a)
        t2 = time.perf_counter()
        print(f"m={m} split rep {rep}: nat.decode {1e3 * (t1 - t0):.3f} ms (boundary total {P.last_decode_profile()['total_ms']:.3f}), import {1e3 * (t2 - t1):.3f} ms", flush=True)
        del out
    gc.disable()
    for rep in range(2):
        t = time.perf_counter()
        res = P.deserialize_array_threaded(sub, S, 8)
        w = time.perf_counter() - t
        print(f"m={m} gc disabled rep {rep}: wall {w * 1e3:.3f} ms, boundary total {P.last_decode_profile()['total_ms']:.3f} ms", flush=True)
        del res
    gc.enable()
