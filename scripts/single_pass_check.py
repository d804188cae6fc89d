"""Single-pass form (RUHVRO_HIP_SINGLE_PASS=1): device-resident decode of n records, repeated so that the second call has the
schema's history, buffer identity against the oracle, engine counters.  python scripts/single_pass_check.py [n] [k]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from arrow_compare import assert_batches_identical
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
from oracle import c_walker
from pyruhvro_amd import cabi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for cfg in ("full", "cfg3", "array_and_map") if len(sys.argv) < 4 else (sys.argv[3],):
    if cfg in ("full", "cfg3", "flat4"):
        data, offsets = fastgen.generate(cfg, n)
    else:
        from avrogen import synth
        data, offsets = c_walker.pack(synth.records(cfg, min(n, 50_000), seed=5))
    m = len(offsets) - 1
    exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS[cfg]), data, offsets, k, threaded=True)
    d_data = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda:0")
    d_data[: len(data)].copy_(torch.from_numpy(np.ascontiguousarray(data)))
    d_off = torch.from_numpy(offsets.view(np.int64).copy()).to("cuda:0")
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    c0 = cabi.engine_counters()
    for rep in range(3):
        r = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), m, SCHEMAS[cfg], k, device=0, stream=stream,
                               kernel=cabi.KERNEL_SPECIALIZED, asynchronous=(rep == 2))
        got = r.to_host()
        for g, e in zip(got, exp):
            g.validate(full=True)
            assert_batches_identical(g, e)
        r.free()
    c1 = cabi.engine_counters()
    print(cfg, m, "records: identical x3;", {key: c1[key] - c0[key] for key in c1 if c1[key] != c0[key]})
print("single-pass check ok")
