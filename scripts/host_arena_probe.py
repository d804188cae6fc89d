"""EXPERIMENT: emit kernel writing its Arrow buffers straight into pinned host memory (RUHVRO_HIP_HOST_ARENA=1) vs HBM + D2H copy.
Device-resident input, 125k / 1.25M records per call (one chunk group of a 1M / 10M-record host call)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from pyruhvro_amd import cabi
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
S = SCHEMAS["full"]
for n in (125_000, 1_250_000):
    data, offsets = fastgen.generate("full", n)
    d_data = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda"); d_data[: len(data)].copy_(torch.from_numpy(data))
    d_off = torch.from_numpy(offsets.view(np.int64)).to("cuda"); torch.cuda.synchronize()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), n, S, 1, stream=st).free()
    best = None
    for _ in range(10):
        t = time.perf_counter()
        r = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), n, S, 1, stream=st)
        w = time.perf_counter() - t
        if best is None or w < best[0]: best = (w, dict(r.stats))
        r.free()
    w, s = best
    print(f"HOST_ARENA={os.environ.get('RUHVRO_HIP_HOST_ARENA', '0')} n={n}: call {w * 1e3:.3f} ms, k_size {s['size_kernel_ms']:.3f}, k_emit {s['emit_kernel_ms']:.3f} ms, "
          f"output {s['output_bytes'] / 1e6:.1f} MB -> {s['output_bytes'] / max(s['emit_kernel_ms'], 1e-9) / 1e6:.1f} GB/s in the emit kernel", flush=True)
