#!/bin/bash
# where the host-side microseconds of a small device-resident call go (RUHVRO_HIP_HOSTPROF=1), full schema 1.25M records
RUHVRO_HIP_HOSTPROF=1 timeout 200 python bench.py --workload full10m --records 1250000 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end 2>&1 | grep -E "hostprof|value" | tail -8 | cut -c1-400
timeout 200 python - <<'PY'
import time, torch, numpy as np
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
from pyruhvro_amd import cabi
n = 1_250_000
data, offsets = fastgen.generate("full", n)
d_data = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda"); d_data[:len(data)].copy_(torch.from_numpy(data))
d_off = torch.from_numpy(offsets.view(np.int64)).to("cuda"); torch.cuda.synchronize()
stream = torch.cuda.current_stream().cuda_stream
for want in (True, False):
    for _ in range(5):
        r = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), n, SCHEMAS["full"], 1, device=0, stream=stream, want_stats=want); r.free()
    t = time.perf_counter()
    for _ in range(50):
        r = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), n, SCHEMAS["full"], 1, device=0, stream=stream, want_stats=want); r.free()
    torch.cuda.synchronize()
    print("want_stats", want, "ms/call", (time.perf_counter() - t) / 50 * 1e3)
PY
