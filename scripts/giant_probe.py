"""Kernel times of a decode with a few GIANT records (an `emails` array of `items` strings) among ordinary ones: what a record
larger than the LDS window costs the size and the emit pass (spec_body.h ranged_tile / item_scan).
    python scripts/giant_probe.py [small=20000] [giants=4] [items=9000]      (env knobs as for workload_probe.py)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from avrogen import fastgen
    from avrogen.encoder import zigzag
    from avrogen.schemas import SCHEMAS
    from pyruhvro_amd import cabi
    small = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    giants = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    items = int(sys.argv[3]) if len(sys.argv) > 3 else 9000
    data, offsets = fastgen.generate("full", small)
    recs = fastgen.split(data, offsets)
    body = bytes(bytearray(b"\x00\x00") + zigzag(items) + (zigzag(23) + b"y" * 23) * items + b"\x00" + b"\x00\x00\x00\x00" + zigzag(1_750_000_000) + zigzag(1))
    step = max(small // max(giants, 1), 1)
    for g in range(giants):
        recs[min(g * step + 7, small - 1)] = body
    from oracle import c_walker
    d, o = c_walker.pack(recs)
    d_data = torch.zeros(len(d) + 64, dtype=torch.uint8, device="cuda:0")
    d_data[: len(d)].copy_(torch.from_numpy(d.copy()))
    d_off = torch.from_numpy(o.view(np.int64).copy()).to("cuda:0")
    schema = SCHEMAS["full"]
    cabi.prebuild(schema)
    call = cabi.PreparedDeviceDecode(d_data.data_ptr(), d_off.data_ptr(), int(o[-1]), len(recs), schema, 4, device=0,
                                     stream=torch.cuda.current_stream().cuda_stream, kernel=2)
    for _ in range(3):
        call.free(call.run(False))
    acc = {"size_kernel_ms": 0.0, "emit_kernel_ms": 0.0}
    reps = 5
    for _ in range(reps):
        h = call.run(True)
        for k in acc:
            acc[k] += getattr(call.stats, k)
        call.free(h)
    print(json.dumps({"small": small, "giants": giants, "items": items, "bytes": int(o[-1]),
                      "k_size_ms": round(acc["size_kernel_ms"] / reps, 4), "k_emit_ms": round(acc["emit_kernel_ms"] / reps, 4),
                      "env": {e: os.environ[e] for e in sorted(os.environ) if e.startswith("RUHVRO_HIP_") and e != "RUHVRO_HIP_SKIP_WARM"}}))


if __name__ == "__main__":
    import torch  # noqa: F401
    main()
