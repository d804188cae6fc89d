"""A/B of host-path knobs inside ONE process on ONE box (boxes differ by 20 % and more): for each env setting a fresh child
process times deserialize_array_threaded at 1M and 10M records (best of 5, results dropped between calls) and rh_decode on slices.
    python scripts/host_ab.py "NAME=VAL,NAME2=VAL2" "..."      ("" = defaults)"""
import json, os, subprocess, sys
CHILD = r'''
import sys, time, json
sys.path.insert(0, '.')
import torch, numpy as np
import pyruhvro_amd as P
from pyruhvro_amd import cabi
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
S = SCHEMAS["full"]
data, offsets = fastgen.generate("full", 10_000_000)
recs = fastgen.split(data, offsets)
out = {}
ptrs = (np.uint64(data.ctypes.data) + offsets[:-1]).astype(np.uint64); lens = np.diff(offsets).astype(np.uint64)
for m in (250_000, 1_000_000, 10_000_000):
    best = 1e9
    for _ in range(5):
        t = time.perf_counter(); r = cabi.decode_slices(ptrs[:m], lens[:m], S, 8); w = time.perf_counter() - t; del r
        best = min(best, w)
    out["slices_%d" % m] = round(best * 1e3, 3)
    best = 1e9
    om = offsets[: m + 1]
    for _ in range(5):
        t = time.perf_counter(); r = cabi.decode_packed(data[: int(om[-1])], om, S, 8); w = time.perf_counter() - t; del r
        best = min(best, w)
    out["packed_%d" % m] = round(best * 1e3, 3)
for m in (10_000, 1_000_000, 10_000_000):
    sub = recs[:m] if m < len(recs) else recs
    for _ in range(3): P.deserialize_array_threaded(sub, S, 8)
    best = 1e9
    for _ in range(7 if m > 10_000 else 200):
        t = time.perf_counter(); r = P.deserialize_array_threaded(sub, S, 8); w = time.perf_counter() - t; del r
        if w < best:
            best = w; pr = P.last_decode_profile()
    out["py_%d" % m] = round(best * 1e3, 3)
    out["py_%d_extract/tail/gil" % m] = [round(pr["extract_ms"], 2), round(pr["engine_tail_release_ms"], 2), round(pr["gil_held_ms"], 2)]
print("RESULT " + json.dumps(out))
'''
for setting in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("=", 1); env[k] = v
    for rep in range(2):
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        print(f"{setting or 'defaults':40s} run {rep}: {line[0][7:] if line else 'FAILED ' + p.stderr[-400:]}", flush=True)
