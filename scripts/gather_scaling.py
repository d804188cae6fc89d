#!/usr/bin/env python3
"""Host side of BASELINE config 5 without an 8-GPU node: how fast do g shards gather their record
slices into their staging buffers when all g run at once?  rh_bench_gather (C ABI hook) runs the engine's own two gather
passes (engine.cpp gather_into = gather_slices) on g host threads x t helpers each, into pageable memory (no GPU needed)
or pinned memory (GPU box).  The bar: 8 links x ~50 GB/s of H2D would take ~400 GB/s of gathered bytes.

    python scripts/gather_scaling.py [records=10000000] [reps=5] > profiles/r04_gather_scaling.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avrogen import fastgen  # noqa: E402
from pyruhvro_amd import cabi  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    data, offsets = fastgen.generate("full", n)
    ptrs = (np.uint64(data.ctypes.data) + offsets[:-1]).astype(np.uint64)
    lens = np.diff(offsets).astype(np.uint64)
    L = cabi.lib()
    L.rh_bench_gather.restype = C.c_uint64
    L.rh_bench_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_double)]
    ncpu = os.cpu_count() or 1
    have_gpu = L.rh_device_count() > 0
    out = {"records": n, "payload_bytes": int(offsets[-1]), "host_cpus": ncpu, "gpu_box": bool(have_gpu), "reps": reps,
           "what": "rh_bench_gather: g shards gather their contiguous share of the record slices at the same time, t helper threads per shard "
                   "(sum of lengths, then offsets + memcpy per record -- engine.cpp gather_into); GB/s = payload bytes / best wall time",
           "bar": "8 PCIe links x ~50 GB/s H2D = ~400 GB/s of gathered payload for 8 GPUs; one link ~50", "runs": []}
    L.rh_numa_nodes.restype = C.c_uint32
    out["numa_nodes"] = int(L.rh_numa_nodes())
    try:
        out["cgroup_cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        out["cgroup_cpu_max"] = None
    out["effective_cpus"] = int(L.rh_effective_cpus())
    for pinned in ([0, 1, 2] if have_gpu else [0, 2]):
        for g in (1, 2, 4, 8):
            for t in sorted({1, 2, 4, 8, max(1, min(32, ncpu // g))}):
                if g * t > 2 * ncpu:
                    continue
                ms = C.c_double()
                b = L.rh_bench_gather(ptrs.ctypes.data, lens.ctypes.data, n, g, t, pinned, reps, C.byref(ms))
                if not b:
                    continue
                out["runs"].append({"shards": g, "threads_per_shard": t, "pinned": pinned == 1, "numa_placed": pinned == 2, "ms": round(ms.value, 3),
                                    "GBps": round(b / ms.value / 1e6, 2), "GBps_per_shard": round(b / ms.value / 1e6 / g, 2)})
    best = {}
    for r in out["runs"]:
        key = f"g={r['shards']}{' pinned' if r['pinned'] else ' pageable, numa-placed' if r['numa_placed'] else ' pageable'}"
        if key not in best or r["GBps"] > best[key]["GBps"]:
            best[key] = r
    out["best"] = {k: {"threads_per_shard": v["threads_per_shard"], "GBps": v["GBps"], "ms": v["ms"]} for k, v in best.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
