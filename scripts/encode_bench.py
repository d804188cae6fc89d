#!/usr/bin/env python3
"""Timing of the Arrow -> Avro direction (rh_encode) on the BASELINE config-4 shape: host batch in, host
BinaryArrays out (PCIe inclusive), plus the two kernels' own durations.  Not the round's headline metric."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pyruhvro_amd as P  # noqa: E402
from avrogen import fastgen  # noqa: E402
from avrogen.schemas import SCHEMAS  # noqa: E402
from pyruhvro_amd import cabi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
data, offsets = fastgen.generate("full", n)
batch = cabi.decode_packed(data, offsets, SCHEMAS["full"], 1)[0]
for form in ("specialized", "generic"):
    P.set_kernel_mode(form)
    best = None
    for _ in range(4):
        t = time.perf_counter()
        out, st = P.serialize_record_batch_with_stats(batch, SCHEMAS["full"], 8)
        dt = time.perf_counter() - t
        if best is None or dt < best[0]:
            best = (dt, st)
    total = sum(int(np.frombuffer(a.buffers()[1], dtype=np.int32, count=len(a) + 1)[-1]) for a in out)
    assert total == int(offsets[-1])
    dt, st = best
    kern_ms = st["size_kernel_ms"] + st["scan_kernel_ms"] + st["emit_kernel_ms"]
    alg = st["input_bytes"] + total + 4 * (n + 8)          # Arrow bytes in + Avro bytes out + i32 offsets out
    print(json.dumps({"direction": "arrow->avro", "kernel_form": form, "rows": n, "arrow_bytes_in": st["input_bytes"],
                      "avro_bytes": total, "end_to_end_ms": dt * 1e3, "rows_per_s_end_to_end": n / dt,
                      "e_size_ms": st["size_kernel_ms"], "e_emit_ms": st["emit_kernel_ms"],
                      "rows_per_s_kernels": n / (kern_ms * 1e-3),
                      "emit_alg_GBps": alg / (st["emit_kernel_ms"] * 1e-3) / 1e9,
                      "emit_lds_bytes": st["lds_bytes"], "h2d_ms": st["h2d_ms"], "d2h_ms": st["d2h_ms"]}))
