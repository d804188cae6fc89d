"""Device-resident decode of one synthetic workload: kernel times (HIP timestamps of the launches, rh_stats), the tile
statistics of rh_engine_counters, and buffer identity with the oracle -- what bench.py's `other_configs` lines carry, as a
stand-alone command for A/B runs (env knobs: RUHVRO_HIP_VARIANT, RUHVRO_HIP_WIN_BYTES, RUHVRO_HIP_NO_TRUST ...).

    python scripts/workload_probe.py full_realistic 1000000 [--kernel generic] [--chunks 8] [--reps 20] [--no-parity]

Prints ONE JSON line.  Needs an MI355X."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def probe(workload: str, n: int, chunks: int = 8, kernel: str = "auto", reps: int = 20, parity: bool = True, parity_max: int = 0):
    import numpy as np
    import torch
    from avrogen import fastgen
    from avrogen.schemas import SCHEMAS
    from pyruhvro_amd import cabi
    kern = {"auto": 0, "generic": 1, "specialized": 2}[kernel]
    schema = SCHEMAS[workload]
    t0 = time.perf_counter()
    data, offsets = fastgen.generate(workload, n)
    gen_s = time.perf_counter() - t0
    dev = torch.device("cuda", 0)
    d_data = torch.empty(len(data) + 64, dtype=torch.uint8, device=dev)
    d_data[: len(data)].copy_(torch.from_numpy(data))
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if kern != 1:
        cabi.prebuild(schema)
    prebuild_s = time.perf_counter() - t0
    stream = torch.cuda.current_stream().cuda_stream
    call = cabi.PreparedDeviceDecode(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), n, schema, chunks, device=0,
                                     stream=stream, kernel=kern)
    for _ in range(3):
        call.free(call.run(False))
    torch.cuda.synchronize()
    c0 = cabi.engine_counters()
    acc = {"size_kernel_ms": 0.0, "scan_kernel_ms": 0.0, "emit_kernel_ms": 0.0}
    out_bytes = 0
    t0 = time.perf_counter()
    for _ in range(reps):
        h = call.run(True)
        for key in acc:
            acc[key] += getattr(call.stats, key)
        out_bytes = int(call.stats.output_bytes)
        spec = int(call.stats.specialized)
        lds = int(call.stats.lds_bytes)
        call.free(h)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / reps
    c1 = cabi.engine_counters()
    ctr = {k: (c1[k] - c0[k]) / reps for k in ("tiles", "careful_tiles", "over_window_tiles", "rewalked_waves", "subtiled_tiles")}
    k = {key.replace("_kernel_ms", ""): v / reps for key, v in acc.items()}
    alg = int(offsets[-1]) + 8 * n + out_bytes
    path = k["size"] + k["scan"] + k["emit"]
    out = {"workload": workload, "records": n, "chunks": chunks, "kernel_form": "specialised" if spec else "generic",
           "input_bytes": int(offsets[-1]), "arrow_bytes": out_bytes, "bytes_per_record": alg / n,
           "kernel_ms": {"k_size": round(k["size"], 4), "k_scan": round(k["scan"], 4), "k_emit": round(k["emit"], 4), "path": round(path, 4)},
           "sync_call_ms": round(wall, 4), "records_per_s": n / (path * 1e-3) if path else 0.0,
           "path_frac": alg / (path * 1e-3) / 8e12 if path else 0.0, "emit_frac": alg / (k["emit"] * 1e-3) / 8e12 if k["emit"] else 0.0,
           "lds_bytes": lds, "per_call": ctr, "gen_s": round(gen_s, 2), "prebuild_s": round(prebuild_s, 2),
           "env": {e: os.environ[e] for e in sorted(os.environ) if e.startswith("RUHVRO_HIP_")}}
    if parity:
        from arrow_compare import assert_batches_identical
        from oracle import c_walker
        m = n if not parity_max else min(n, parity_max)
        dl = int(offsets[m])
        t0 = time.perf_counter()
        r = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), dl, m, schema, chunks, device=0, stream=stream, kernel=kern)
        got = r.to_host()
        r.free()
        exp = c_walker.decode_packed(c_walker.CompiledSchema(schema), data[:dl], offsets[: m + 1], chunks, threaded=True)
        res = "identical"
        try:
            assert len(got) == len(exp)
            for g, e in zip(got, exp):
                assert_batches_identical(g, e)
        except AssertionError as e:
            res = "DIFFERENT: " + str(e)[:300]
        out["parity_check"] = {"records": m, "result": res, "check_s": round(time.perf_counter() - t0, 2)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("records", type=int)
    ap.add_argument("--chunks", type=int, default=8)
    ap.add_argument("--kernel", default="auto", choices=("auto", "generic", "specialized"))
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--parity-max", type=int, default=0)
    a = ap.parse_args()
    import torch  # noqa: F401  (first: the library binds to torch's HIP runtime)
    print(json.dumps(probe(a.workload, a.records, a.chunks, a.kernel, a.reps, not a.no_parity, a.parity_max)))


if __name__ == "__main__":
    main()
