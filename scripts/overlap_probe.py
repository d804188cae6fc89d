"""Does running the size pass of one call beside the emit pass of another raise throughput?  Two host threads, each
decoding the same device-resident 10M-record batch on its own stream, against one thread doing the same number of
calls back to back.  (k_size is VALU-issue bound, k_emit store-path bound with VALU at 57 %: if the two overlap, a
per-chunk software pipeline inside ONE call would pay.)"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
from pyruhvro_amd import cabi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
data, offsets = fastgen.generate("full", n)
d_data = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda"); d_data[:len(data)].copy_(torch.from_numpy(data))
d_off = torch.from_numpy(offsets.view(np.int64)).to("cuda"); torch.cuda.synchronize()
dl = int(offsets[-1])


def worker(stream, iters, out):
    for _ in range(iters):
        r = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), dl, n, SCHEMAS["full"], 8, device=0, stream=stream.cuda_stream, want_stats=False)
        r.free()
    out.append(1)


s = [torch.cuda.Stream() for _ in range(3)]
worker(s[0], 3, [])
for nthreads in (1, 2, 3, 1, 2):
    iters = 24 // nthreads
    done = []
    th = [threading.Thread(target=worker, args=(s[i], iters, done)) for i in range(nthreads)]
    torch.cuda.synchronize()
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(json.dumps({"threads": nthreads, "calls": iters * nthreads, "ms_per_call": dt * 1e3 / (iters * nthreads), "records_per_s": n * iters * nthreads / dt}))
