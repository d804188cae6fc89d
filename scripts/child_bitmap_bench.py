"""A/B of the child-domain bitmap path: a schema whose list items / map values are nullable (every item sets a validity
bit in a child row domain), 2M records, device-resident decode, specialised kernels.  (The A/B recorded in profiles/r02f_child_bitmap_ab.jsonl ran it against a
build with one global atomic per bit, variant BM_GLOBAL, since deleted.)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from arrow_compare import assert_batches_identical
from avrogen.encoder import to_datum
from oracle import c_walker
from oracle.avro_schema import parse_schema
from pyruhvro_amd import cabi

SCHEMA = json.dumps({"type": "record", "name": "CB", "fields": [
    {"name": "id", "type": "long"},
    {"name": "a", "type": {"type": "array", "items": ["null", "string"]}},
    {"name": "m", "type": {"type": "map", "values": ["null", "boolean"]}},
    {"name": "b", "type": {"type": "array", "items": ["null", "int"]}}]})
sc = parse_schema(SCHEMA)
rng = np.random.default_rng(7)
base = []
for i in range(20000):
    base.append(to_datum(sc, {"id": i, "a": [None if rng.random() < .3 else "s%d" % j * (1 + j) for j in range(int(rng.integers(0, 4)))],
                              "m": [("k%d" % j, None if rng.random() < .3 else bool(j & 1)) for j in range(int(rng.integers(0, 4)))],
                              "b": [None if rng.random() < .3 else int(j * 1000) for j in range(int(rng.integers(0, 6)))]}))
data1, off1 = c_walker.pack(base)
data1 = data1[: int(off1[-1])]          # pack() pads the payload
REP = 100
data = np.tile(data1, REP)
offsets = np.concatenate([off1[:-1] + np.uint64(r * int(off1[-1])) for r in range(REP)] + [np.array([REP * int(off1[-1])], dtype=np.uint64)])
n = len(offsets) - 1
exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMA), data1, off1, 1, threaded=False)[0]
got = cabi.decode_packed(data1, off1, SCHEMA, 1, kernel=2)[0]
assert_batches_identical(got, exp)
d_data = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda"); d_data[:len(data)].copy_(torch.from_numpy(data))
d_off = torch.from_numpy(offsets.view(np.int64)).to("cuda"); torch.cuda.synchronize()
best = None
for _ in range(8):
    r = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), n, SCHEMA, 8, device=0, kernel=2)
    st = r.stats; r.free()
    if best is None or st["emit_kernel_ms"] < best["emit_kernel_ms"]:
        best = st
print(json.dumps({"variant": os.environ.get("RUHVRO_HIP_VARIANT", ""), "records": n, "k_size_ms": best["size_kernel_ms"], "k_emit_ms": best["emit_kernel_ms"]}))
