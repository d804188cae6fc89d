"""k < g: is it worth splitting ONE chunk (deserialize_array's single batch) over two GPUs and concatenating the halves on
the host?  Measured on one GPU with two logical shards of device 0 (the same partition / reassembly code as two devices;
the kernels of the halves run one after the other here, so the split leg is charged up to 2x its kernel time -- 1-2 ms of a
40+ ms call): host-in -> host-out of the single batch vs of two half batches + the host concatenation a caller would need
to get ONE batch back (pyarrow concat_batches: offsets rebased, bitmaps re-packed, buffers copied).
    python scripts/k_lt_g_split.py [records]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import pyarrow as pa  # noqa: E402
from avrogen import fastgen  # noqa: E402
from avrogen.schemas import SCHEMAS  # noqa: E402
from pyruhvro_amd import cabi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
data, offsets = fastgen.generate("full", n)
S = SCHEMAS["full"]


def best(f, reps=3):
    b = None
    for _ in range(reps):
        t = time.perf_counter()
        r = f()
        w = time.perf_counter() - t
        b = w if b is None or w < b else b
        del r
    return b * 1e3


one = best(lambda: cabi.decode_packed(data, offsets, S, 1))
two = best(lambda: cabi.decode_packed(data, offsets, S, 2, devices=[0, 0]))
halves = cabi.decode_packed(data, offsets, S, 2, devices=[0, 0])
cat = best(lambda: pa.concat_batches(halves))
whole = cabi.decode_packed(data, offsets, S, 1)[0]
assert pa.concat_batches(halves).equals(whole)
print(json.dumps({"records": n, "single_batch_one_gpu_ms": round(one, 2), "two_half_batches_two_shards_ms": round(two, 2),
                  "host_concat_of_the_halves_ms": round(cat, 2), "split_total_ms": round(two + cat, 2),
                  "what": "k < g: one chunk split over two shards + the host concatenation that makes it one batch again"}))
