"""Quick Arrow -> Avro byte identity on the full schema: n records of the generator decoded on the GPU, re-encoded in 8
chunks by the specialised kernels, compared byte for byte with the generator's datums -- used by gpu_ab_encode.sh to
reject a variant that changes results before its timing is looked at."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import pyruhvro_amd as P
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
from pyruhvro_amd import cabi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
schema = SCHEMAS["full"]
data, offsets = fastgen.generate("full", n)
batch = cabi.decode_packed(data, offsets, schema, 1)[0]
P.set_kernel_mode("specialized")
out = P.serialize_record_batch(batch, schema, 8)
row = 0
for a in out:
    m = len(a)
    off = np.frombuffer(a.buffers()[1], dtype=np.int32, count=m + 1)
    want_off = (offsets[row:row + m + 1] - offsets[row]).astype(np.int64)
    assert np.array_equal(off.astype(np.int64), want_off), "offsets differ"
    got = np.frombuffer(a.buffers()[2], dtype=np.uint8, count=int(off[-1]))
    want = data[int(offsets[row]):int(offsets[row + m])]
    assert np.array_equal(got, want), "bytes differ in chunk starting at row %d" % row
    row += m
assert row == n
print("parity ok")
