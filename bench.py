#!/usr/bin/env python3
"""bench.py -- Avro -> Arrow direct decode on MI355X: records/s + achieved HBM GB/s.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (k_size -> k_scan+layout -> k_emit -> k_publish, C ABI rh_decode_device) over one batch
of synthetic Avro records that is ALREADY resident in HBM; the Arrow buffers are produced in HBM.  Workload
(BASELINE.json config 4, the one the metric's target is quoted on): 10,000,000 records of the
scripts/generate_avro.py schema, num_chunks = 8.

The K timed steps are made with RH_ASYNC (ABI 4): each call returns once it is on the stream and is settled
(rh_device_result_wait: error check, row / null counts) and freed PIPELINE_DEPTH steps later, so the host's turn-around
overlaps the next step; every step's kernels, control-word publish and settle are inside the timed region, which is
closed by draining the pipeline + the barrier + torch.cuda.synchronize().  `config.sync_call_ms` is the same call made
synchronously.  The arenas / workspaces the in-flight calls own are taken from the allocator (and touched) in set-up.

N > 1 (BASELINE.json config 5): the SAME seeded 10M-record list, its 8 reference chunks
(ruhvro/src/deserialize.rs:57-68) dealt to the ranks in contiguous runs (rh_shard_chunks: rank r of N decodes chunks
[r*8/N, (r+1)*8/N)), so every batch is produced by one GPU -- "scaling": "strong", total work fixed.  No data-path
collective: records are independent; RCCL carries the barrier, the MAX over rank times and the stats all-gather.
`--scaling weak` keeps round 1's mode (every rank decodes its own 10M records of the stream).

Rank 0 prints ONE JSON line (contract in the task statement) with these extra objects:
  roofline      k_emit (the dominant kernel): algorithmic bytes per launch / its mean launch duration,
                measured with HIP events on the launch stream inside the timed steps, vs 8 TB/s HBM.  Also the whole
                path against the same bytes (`path_frac`), the north star's own figure -- HBM READ bandwidth of the two
                passes together, rocprofv3 FETCH_SIZE of both (stamped file) / their time (`read_GBps`, `read_frac`) --
                and one sub-object per kernel (`kernels`).
  overlapped    a SECOND timed region: the same K steps dealt round-robin to --overlap-streams (3) HIP streams.  Independent
                batches overlap well (the size pass is VALU-bound, the emit pass store-path-bound), but kernels that share
                the chip have no per-kernel duration to price against the roofline, so this is reported beside `value`,
                never as it.  (Profiler passes run with --overlap-streams 0: their per-kernel averages are those of the
                single-stream region the `roofline` object describes.)
  config5_projection  (N=1) the step ONE rank of BASELINE config 5 would run when the list is dealt to 2 / 4 / 8 GPUs,
                measured on this GPU, and the strong-scaling efficiency it implies, single-stream and overlapped.
                A projection, labelled as one.
  cpu_baseline  the oracle's C restatement of the reference walker ("port"), reference threading shape
                (serial pack + one thread per chunk), timed on this box's host cores on the SAME records, best of 5:
                with the workload's 8 chunks = 8 threads, and (`wide`) with min(64, cores) chunks.  Reported, not targeted.
  other_configs (N=1 only) compact lines of BASELINE configs 2 and 3, the full schema at 1M records and the Arrow -> Avro
                direction (2M rows, device-resident), so that the driver's run of this default command carries them; and
                (round 6, `workload_line`) the walks OFF the friendly value distribution, each with kernel times, the tile
                statistics of rh_engine_counters and a parity_check against the oracle: `full_realistic_10m` (microsecond
                timestamps, epoch-second ints, snowflake ids, 1 % notes of 8 KiB and more, arrays of > 8,191 items),
                `full_realistic_nogiant_10m` (the same without those arrays), `full_skewed_10m` (record sizes log-normal in
                runs: every tile past the 40 KB window),
                `wide200_1m` (200 nullable string columns + 12 arrays: 224 scanned counters), and the GENERIC kernels --
                what every new schema runs on while its specialised kernels compile -- on config 4 and config 3.
  end_to_end    (N=1 only) the same workload through the HOST entry points of the C ABI -- host records in, host
                Arrow batches out, PCIe both ways included -- so the CPU baseline has a like-for-like neighbour:
                rh_decode_packed (one packed payload), rh_decode (one slice per record, what the CPython boundary
                extracts from list[bytes]) and the Python surface itself on a bounded sample.  Never `value`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KERNEL = 0                      # rh_opts.flags: 0 auto, 1 generic interpreter, 2 schema-specialised
STATS_EVERY = 2                 # every 2nd timed step carries the kernel timestamps (see run(); --stats-every)
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# This process decodes a dozen workloads one after the other, each leaving its arenas (up to 5 GB) in the engine's device pool;
# past the pool's default of 24 GB of idle blocks every release is a hipFree and every call a hipMalloc (the skewed workload's
# synchronous call: 144 ms instead of 6.3).  288 GB of HBM: let the pool keep them (read when the library creates its pools).
os.environ.setdefault("RUHVRO_HIP_DEVICE_CACHE_MB", str(96 * 1024))
WORKLOADS = {
    # name: (generator config, records per GPU, num_chunks, description)
    "full10m": ("full", 10_000_000, 8, "10M records of the generate_avro.py schema (BASELINE.json config 4), num_chunks=8"),
    "full1m": ("full", 1_000_000, 8, "1M records of the generate_avro.py schema, num_chunks=8"),
    "cfg3_1m": ("cfg3", 1_000_000, 8, "1M string + nullable-union + enum records (BASELINE.json config 3)"),
    "flat4_1m": ("flat4", 1_000_000, 8, "1M flat-primitive records (BASELINE.json config 2)"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="full10m", choices=sorted(WORKLOADS))
    ap.add_argument("--records", type=int, default=0, help="override records per GPU (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stats-every", type=int, default=STATS_EVERY,
                    help="every Nth timed step carries the kernels' HIP-event timestamps (costs such a step ~30 us)")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the asynchronous steps are dealt to round-robin")
    ap.add_argument("--overlap-streams", type=int, default=OVERLAP_STREAMS,
                    help="streams of the second timed region reported as `overlapped` (0 = skip)")
    ap.add_argument("--sync-calls", action="store_true", help="every step waits for its own call (no RH_ASYNC pipelining)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the compact lines of BASELINE configs 2-3, full1m and encode")
    ap.add_argument("--no-cold-start", action="store_true", help="skip cold_start (a schema's first call against an empty kernel cache)")
    ap.add_argument("--no-projection", action="store_true", help="skip config5_projection (profiler passes: only full-size launches)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "specialized"])
    ap.add_argument("--cpu-sample", type=int, default=0, help="records of the CPU baseline (0 = the whole workload)")
    ap.add_argument("--direction", default="decode", choices=["decode", "encode"],
                    help="encode = the other direction (SURVEY 8f N1, rh_encode: Arrow -> Avro), a secondary line")
    ap.add_argument("--rows", type=int, default=2_000_000, help="--direction encode: rows of the full schema")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N>1: strong = one list, whole chunks dealt to the ranks (BASELINE config 5); weak = the list per rank")
    return ap.parse_args(argv)


def stamped_traffic(schema_json: str, encode: bool = False):
    """Per-kernel HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/hbm_traffic.json / profiles/encode_hbm_traffic.json, written by scripts/rocpd_summary.py --traffic-json
    from separate FETCH_SIZE / WRITE_SIZE runs, gfx950 FETCH_SIZE x2 correction applied).  The file is stamped with the
    content hash of the kernels it was measured on: a file from another kernel revision is REFUSED ({}), never reported."""
    try:
        from pyruhvro_amd import cabi
        name = "encode_hbm_traffic.json" if encode else "hbm_traffic.json"
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        if d.get("_kernel_key") != cabi.kernel_key(schema_json, encode):
            return {}
        return {k: v for k, v in d.items() if isinstance(v, dict)}
    except Exception:
        return {}


def count_buffers(batch) -> int:
    """Arrow buffers (present ones) of every node of a RecordBatch."""
    import pyarrow as pa

    def walk(a):
        t = a.type
        c = sum(1 for b in a.buffers()[: 3 if (pa.types.is_string(t) or pa.types.is_binary(t)) else 2 if not pa.types.is_struct(t) else 1] if b is not None)
        if pa.types.is_struct(t) or pa.types.is_union(t):
            c += sum(walk(a.field(i)) for i in range(t.num_fields))
        elif pa.types.is_map(t):
            c += walk(a.keys) + walk(a.items)
        elif pa.types.is_list(t):
            c += walk(a.values)
        return c
    return sum(walk(batch.column(i)) for i in range(batch.num_columns))


def parity_check(cs, data, offsets, num_chunks: int, got, config: str):
    """The batches ONE call of the timed configuration produced (settled and copied to the host after the timed regions)
    against the oracle's decode of the same records -- every buffer of every node of every chunk, bit-exact
    (tests/arrow_compare.py, the bar of the GPU parity suite; the reference's own check is assert_round_trip,
    ruhvro/src/fast_decode.rs:945-953).  The oracle is the checker here, never the thing measured."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from arrow_compare import assert_batches_identical
    from oracle import c_walker
    t = time.perf_counter()
    exp = c_walker.decode_packed(cs, data, offsets, num_chunks, threaded=True)
    out = {"config": config, "records": int(len(offsets) - 1), "chunks": len(exp), "buffers": 0, "result": "identical"}
    try:
        assert len(got) == len(exp), f"{len(got)} chunks, the oracle has {len(exp)}"
        for g, e in zip(got, exp):
            assert_batches_identical(g, e)
            out["buffers"] += count_buffers(g)
    except AssertionError as e:
        out["result"] = "DIFFERENT: " + str(e)[:300]
    out["check_s"] = round(time.perf_counter() - t, 2)
    return out


def workload_line(workload: str, n: int, chunks: int = 8, kernel: str = "auto", reps: int = 20, parity: bool = True, parity_max: int = 0):
    """One synthetic workload, device-resident, synchronous calls that carry the kernels' timestamps: kernel times, the tile
    statistics rh_k_publish sums (rh_engine_counters: careful / over-window / sub-tiled tiles, re-walked wavefronts -- all 0 on
    input that stays inside the fast wire forms and the LDS window), algorithmic-byte fractions of the 8 TB/s peak, and the
    buffers of one more call against the oracle.  What `other_configs` reports for the workloads off the benchmark generator's
    distribution and for the generic kernels; scripts/workload_probe.py is the same as a command (A/B runs)."""
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
    # warm-up: the first settled calls tell the schema whether its ranged kernels are needed; a call that meets tiles past the
    # window before that pair is loaded is repeated on the generic kernels (RH_CTR_RANGED_RETRIES) -- not what is measured here
    quiet, t_w = 0, time.perf_counter()
    while quiet < 5 and time.perf_counter() - t_w < 120.0:
        cw = cabi.engine_counters()
        call.free(call.run(False))
        cx = cabi.engine_counters()
        quiet = quiet + 1 if all(cx[key] == cw[key] for key in ("ranged_retries", "background_compiles", "capacity_retries")) else 0
    torch.cuda.synchronize()
    c0 = cabi.engine_counters()
    acc = {"size_kernel_ms": 0.0, "scan_kernel_ms": 0.0, "emit_kernel_ms": 0.0}
    out_bytes = spec = lds = 0
    t0 = time.perf_counter()
    for _ in range(reps):
        h = call.run(True)
        for key in acc:
            acc[key] += getattr(call.stats, key)
        out_bytes, spec, lds = int(call.stats.output_bytes), int(call.stats.specialized), int(call.stats.lds_bytes)
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
           "path_frac": alg / (path * 1e-3) / 1e9 / HBM_PEAK_GBPS if path else 0.0,
           "emit_frac": alg / (k["emit"] * 1e-3) / 1e9 / HBM_PEAK_GBPS if k["emit"] else 0.0,
           "lds_bytes": lds, "per_call": ctr, "gen_s": round(gen_s, 2), "prebuild_s": round(prebuild_s, 2),
           "env": {e: os.environ[e] for e in sorted(os.environ) if e.startswith("RUHVRO_HIP_")}}
    if parity:
        m = n if not parity_max else min(n, parity_max)
        dl = int(offsets[m])
        r = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), dl, m, schema, chunks, device=0, stream=stream, kernel=kern)
        got = r.to_host()
        r.free()
        from oracle import c_walker
        out["parity_check"] = parity_check(c_walker.CompiledSchema(schema), data[:dl], offsets[: m + 1], chunks, got,
                                           f"rh_decode_device, {m} records of {workload}, num_chunks={chunks}, {out['kernel_form']} kernels")
    del d_data, d_off
    return out


def cpu_baseline(gen_cfg: str, schema_json: str, n_sample: int, num_chunks: int, parity_of=None):
    """Oracle C walker ("port" of ruhvro/src/fast_decode.rs) with the reference's threading shape (serial pack + one task
    per chunk, ruhvro/src/deserialize.rs:76-121), on the SAME inputs as the GPU leg, best of 5 after a warm-up: once
    with the workload's own num_chunks (= threads), once with as many chunks as this host has cores (capped at 64) so
    that the host-in -> host-out neighbour of `end_to_end` is a CPU that was given its cores."""
    from avrogen import fastgen
    from oracle import c_walker
    data, offsets = fastgen.generate(gen_cfg, n_sample)
    cs = c_walker.CompiledSchema(schema_json)

    def timed(threads):
        c_walker.decode_packed(cs, data, offsets, threads, threaded=True, materialize=False)   # warm-up
        best = float("inf")
        for _ in range(5):
            t = time.perf_counter()
            c_walker.decode_packed(cs, data, offsets, threads, threaded=True, materialize=False)
            best = min(best, time.perf_counter() - t)
        return n_sample / best

    ncpu = os.cpu_count() or 1
    wide = max(num_chunks, min(64, ncpu))
    out = {
        "value": timed(num_chunks), "unit": "records/s", "cores": num_chunks, "kind": "port",
        "sample": f"all {n_sample} records, {num_chunks} chunks = {num_chunks} threads (serial pack + 1 task per chunk), best of 5; {ncpu} cpus",
    }
    if wide != num_chunks:
        out["wide"] = {"value": timed(wide), "unit": "records/s", "cores": wide,
                       "sample": f"same records, {wide} chunks = {wide} threads (what the reference would use with num_chunks={wide}), best of 5"}
    out["reference"] = reference_probe()
    # the same port at the metric's other sizes (BASELINE.json: 10k / 1M / 10M), the neighbours of end_to_end's Python-surface lines
    out["at_metric_sizes"] = {}
    for m in (10_000, 1_000_000):
        if m < n_sample:
            off_m = offsets[: m + 1]
            best = float("inf")
            for _ in range(6):
                t = time.perf_counter()
                c_walker.decode_packed(cs, data[: int(off_m[-1])], off_m, num_chunks, threaded=True, materialize=False)
                best = min(best, time.perf_counter() - t)
            out["at_metric_sizes"][str(m)] = {"value": m / best, "wall_ms": best * 1e3, "unit": "records/s", "cores": num_chunks}
    parity = None
    if parity_of is not None:       # (got batches, description) of one call of the timed GPU configuration on these same records
        got, config = parity_of
        parity = parity_check(cs, data, offsets, num_chunks, got, config)
    return out, parity


def cold_start(gen_cfg: str, schema_json: str, n: int = 100_000, num_chunks: int = 8):
    """What a NEW schema costs (the reference: a JSON parse, src/lib.rs:39-54).  An empty kernel cache and the compiler's own
    cache off: the first n-record device-resident call of a fresh schema handle is served by the generic kernels while the
    specialised pair compiles in helper processes (kernel_jobs.cpp); `compile_s` later the next call runs on them."""
    import shutil
    import tempfile
    import numpy as np
    import torch
    from avrogen import fastgen
    from pyruhvro_amd import cabi
    data, offsets = fastgen.generate(gen_cfg, n)
    d_data = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
    d_data[: len(data)].copy_(torch.from_numpy(data))
    d_off = torch.from_numpy(offsets.view(np.int64)).to("cuda")
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    args = (d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), n)
    tmp = tempfile.mkdtemp(prefix="rh_cold_")
    saved = {k: os.environ.get(k) for k in ("RUHVRO_HIP_KERNEL_CACHE", "AMD_COMGR_CACHE")}
    os.environ["RUHVRO_HIP_KERNEL_CACHE"] = tmp
    os.environ["AMD_COMGR_CACHE"] = "0"          # (inherited by the compile helpers)
    try:
        fresh = schema_json + "\n\n"             # a schema handle of its own: nothing loaded, nothing remembered
        t = time.perf_counter()
        r = cabi.decode_device(*args, fresh, num_chunks, stream=stream)
        first_ms = (time.perf_counter() - t) * 1e3
        spec_first = int(r.stats["specialized"])
        r.free()
        t = time.perf_counter()
        r = cabi.decode_device(*args, fresh, num_chunks, stream=stream)
        second_ms = (time.perf_counter() - t) * 1e3
        r.free()
        t = time.perf_counter()
        ready = cabi.kernels_ready(fresh, timeout_ms=300_000)
        compile_s = time.perf_counter() - t
        r = cabi.decode_device(*args, fresh, num_chunks, stream=stream)
        spec_after = int(r.stats["specialized"])
        r.free()
        t = time.perf_counter()
        r = cabi.decode_device(*args, fresh, num_chunks, stream=stream)
        warm_ms = (time.perf_counter() - t) * 1e3
        r.free()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        shutil.rmtree(tmp, ignore_errors=True)
    return {"records": n, "cold_first_call_ms": first_ms, "specialized_first_call": spec_first, "second_call_ms": second_ms,
            "compile_s": compile_s, "kernels_ready": bool(ready), "specialized_after": spec_after, "warm_call_ms": warm_ms,
            "what": (f"fresh schema handle, empty kernel cache, AMD_COMGR_CACHE=0: first synchronous rh_decode_device of {n} records (generic "
                     "kernels; rh_spec_size / rh_spec_emit start compiling in two rh_kcompile helper processes), the call after it, the "
                     "wait until rh_schema_kernels_ready, and a call on the specialised kernels")}


def reference_probe():
    """the reference's own numbers need a Rust toolchain on the bench host (cargo bench -p ruhvro --bench
    deserialize, ruhvro/benches/deserialize.rs:55-82; scripts/run_benchmarks.sh).  Probe for one; run the bench when it and the
    reference tree are both there (never the case on the GPU box: /root/reference does not travel)."""
    import shutil
    import subprocess
    cargo = shutil.which("cargo")
    if not cargo:
        return "unavailable (no cargo on this host)"
    ref = os.environ.get("RUHVRO_REFERENCE_DIR", "/root/reference")
    if not os.path.exists(os.path.join(ref, "Cargo.toml")):
        return f"unavailable (cargo at {cargo}, but no reference tree at {ref})"
    try:
        p = subprocess.run([cargo, "bench", "--offline", "-p", "ruhvro", "--bench", "deserialize"], cwd=ref, capture_output=True, text=True, timeout=900)
        tail = [ln for ln in p.stdout.splitlines() if "time:" in ln or "thrpt:" in ln][-12:]
        return {"rc": p.returncode, "lines": tail} if p.returncode == 0 else f"cargo bench failed (rc {p.returncode}): {p.stderr[-300:]}"
    except Exception as e:      # noqa: BLE001 - reported in the line
        return f"cargo bench did not run: {e}"


def end_to_end(gen_cfg: str, schema_json: str, n: int, num_chunks: int):
    """Host in -> host out through the C ABI's host entry points (PCIe-inclusive), best of 3 each, with the engine's
    stage timings (rh_stats).  The payload is generated once; slices point into it."""
    import numpy as np
    import pyruhvro_amd as P
    from avrogen import fastgen
    from pyruhvro_amd import cabi

    data, offsets = fastgen.generate(gen_cfg, n)
    out = {"records": n, "input_bytes": int(offsets[-1]), "num_chunks": num_chunks, "unit": "records/s"}

    def best_of(f, reps=3):
        best = None
        for _ in range(reps):
            t = time.perf_counter()
            res, st = f()
            wall = time.perf_counter() - t
            del res
            if best is None or wall < best[0]:
                best = (wall, st)
        wall, st = best
        keys = ("pack_ms", "h2d_ms", "size_kernel_ms", "scan_kernel_ms", "emit_kernel_ms", "d2h_ms", "total_ms")
        return {"value": n / wall, "wall_ms": wall * 1e3, "stage_ms": {k: round(float(st[k]), 3) for k in keys}}

    out["packed_pageable"] = best_of(lambda: cabi.decode_packed(data, offsets, schema_json, num_chunks, want_stats=True))
    out["packed_pageable"]["what"] = "rh_decode_packed: one pageable payload + u64 offsets -> host RecordBatches"
    ptrs = (np.uint64(data.ctypes.data) + offsets[:-1]).astype(np.uint64)
    lens = np.diff(offsets).astype(np.uint64)
    out["record_slices"] = best_of(lambda: cabi.decode_slices(ptrs, lens, schema_json, num_chunks, want_stats=True))
    out["record_slices"]["what"] = ("rh_decode: one (pointer, length) per record, gathered into pinned memory per chunk group "
                                    "while earlier groups are on the wire -> host RecordBatches")
    out["packed_8_logical_shards"] = best_of(lambda: cabi.decode_packed(data, offsets, schema_json, num_chunks, want_stats=True,
                                                                        devices=[0] * num_chunks))
    out["packed_8_logical_shards"]["what"] = (f"rh_decode_packed with rh_opts.devices = [0] * {num_chunks}: the in-process multi-GPU driver "
                                              "(one host thread + stream per shard) on this one GPU")
    # The Python surface itself at the metric's own sizes (BASELINE.json quotes 10k / 1M / 10M through
    # deserialize_array_threaded, src/lib.rs:73-89): list[bytes] in, list[RecordBatch] out, best of 3, with the boundary's
    # phase split (pyruhvro_amd.last_decode_profile(): set-up, extraction, rest of the engine call, GIL-held time).
    recs = fastgen.split(data, offsets)
    out["python_list_bytes"] = {}
    for m in sorted({min(x, n) for x in (1_000_000, 10_000_000)}):
        sub = recs if m == n else recs[:m]
        P.deserialize_array_threaded(sub, schema_json, num_chunks)
        best, prof = float("inf"), None
        for _ in range(3):
            t = time.perf_counter()
            res = P.deserialize_array_threaded(sub, schema_json, num_chunks)
            w = time.perf_counter() - t
            if w < best:
                best, prof = w, P.last_decode_profile()
            del res
        out["python_list_bytes"][str(m)] = {
            "value": m / best, "wall_ms": best * 1e3, "records": m,
            "gil_held_ms": round(prof["gil_held_ms"], 3),
            "phase_ms": {k: round(prof[k], 3) for k in ("alloc_setup_ms", "extract_ms", "engine_tail_release_ms", "total_ms")},
            "streaming_handover": bool(prof["streaming"]),
            "vs_record_slices": (m / best) / out["record_slices"]["value"] if out["record_slices"]["value"] else None,
            "what": f"pyruhvro_amd.deserialize_array_threaded(list[bytes] of {m} records, schema, {num_chunks}), host in -> host out, best of 3"}
        del sub
    # BASELINE config 1: the reference's own CPU-runnable case, 10,000 records through the Python surface
    small = recs[:10_000]
    del recs
    for _ in range(20):
        P.deserialize_array_threaded(small, schema_json, num_chunks)
    t = time.perf_counter()
    for _ in range(200):
        res = P.deserialize_array_threaded(small, schema_json, num_chunks)
    per = (time.perf_counter() - t) / 200
    del res
    out["config1_python_10k"] = {"value": len(small) / per, "wall_ms": per * 1e3, "records": len(small),
                                 "what": "BASELINE config 1: deserialize_array_threaded(10,000 records, schema, 8), host in -> host out, mean of 200 calls"}
    return out


def run(args, make_step=None, backend="nccl"):
    """Timed loop shared by the real bench and the gloo CPU test (which injects `make_step`)."""
    import torch
    import torch.distributed as dist
    from pyruhvro_amd import dist as rdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = backend == "nccl"
    if use_cuda:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):      # (BENCH_FORCE_DIST: a one-rank RCCL group, to test the plumbing on one GPU)
        rdist.init_process_group(backend, dev if use_cuda else None)
    # Multi-GPU readiness: the launcher's world, the process group's world and --gpus must be ONE
    # number, and every rank must see a device of its own -- a mis-launched run fails here, loudly, instead of printing a line
    # whose n_gpus says one thing while a different number of GPUs worked.
    if getattr(args, "gpus", world) != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: launch N ranks with torch.distributed.run --nproc-per-node N")
    if dist.is_initialized() and dist.get_world_size() != world:
        raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks, WORLD_SIZE says {world}")
    if use_cuda:
        visible = torch.cuda.device_count()
        if visible < 1 or local_rank >= visible:
            raise SystemExit(f"bench.py: rank {rank} (LOCAL_RANK {local_rank}) sees {visible} GPU(s): one process per GPU needs LOCAL_RANK < visible devices")

    gen_cfg, n, num_chunks, desc = WORKLOADS[args.workload]
    if args.records:
        n = args.records
    if getattr(args, "scaling", "weak") == "strong":
        # one list of n records, k reference chunks; this rank owns whole chunks [c0, c1) = rows [lo, hi)
        shard = rdist.strong_shard(n, num_chunks, world, rank)
    else:
        lo, hi = rdist.shard_rows(n, rank)
        shard = {"row_lo": lo, "rows": n, "chunks": min(num_chunks, max(n, 1)), "chunk_rows": 0}
    step, info = make_step(gen_cfg, shard, dev, local_rank)

    def sync():
        if world > 1 or dist.is_initialized():      # (BENCH_FORCE_DIST: the one-rank group runs the same collectives)
            dist.barrier(device_ids=[local_rank]) if use_cuda else dist.barrier()
        if use_cuda:
            torch.cuda.synchronize()

    drain = getattr(step, "drain", lambda: [])
    for _ in range(args.warmup):
        step()
    drain()
    sync()
    t0 = time.perf_counter()
    # Kernel durations come from the kernels' own start / stop timestamps (HIP events handed to the launches by the
    # engine).  Collecting them costs a call ~30 us (measured: 0.231 vs 0.200 ms per 1.25M-record call), so inside the
    # timed region every STATS_EVERY-th step is a timed launch and the others run exactly as a product call does.
    acc = {"size_kernel_ms": 0.0, "scan_kernel_ms": 0.0, "emit_kernel_ms": 0.0}
    sampled = 0
    # A step = one rh_decode_device call, made with RH_ASYNC: the call is on the stream when the C entry point returns and
    # is settled (rh_device_result_wait: error check, row / null counts; then freed) PIPELINE_DEPTH steps later, so the GPU
    # has the next call queued while this one drains -- every step's full work, its control-word read-back and its settle
    # are inside the timed region; only the host's waiting is overlapped.  `sync_call_ms` in the line is the same call made
    # synchronously (one call, one wait), back to back.
    def take(stats_list):
        nonlocal sampled
        for st in stats_list:
            sampled += 1
            for k in acc:
                acc[k] += st.get(k, 0.0)
    for i in range(args.steps):
        se = max(getattr(args, "stats_every", STATS_EVERY), 1)
        want = i % se == se - 1      # (not the first step after the barrier: its launches run ~15 % slow, clocks / TLBs after the idle sync)
        st = step(want) if use_cuda else step()
        if isinstance(st, dict):
            take([st])
        elif st:
            take(st)
    take(drain())
    sync()
    wall = time.perf_counter() - t0
    wall = rdist.max_over_ranks(wall, dev)

    # Second timed region (same K steps, same barriers): the steps dealt round-robin to OVERLAP_STREAMS HIP streams.
    # Consecutive steps are independent batches, so the VALU-bound size pass of one runs beside the store-path-bound emit
    # pass of another and small launches fill each other's tails.  Kernels that share the chip have no meaningful
    # per-kernel duration, so `value` / `roofline` stay with the single-stream region above; this one is reported beside
    # them (`overlapped`).
    run.overlapped = None
    if use_cuda and hasattr(step, "overlapped") and OVERLAP_STREAMS > 1 and not SYNC_CALLS:
        ostep = step.overlapped(OVERLAP_STREAMS)
        for _ in range(max(args.warmup, 2 * OVERLAP_STREAMS)):
            ostep()
        ostep.drain()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ostep()
        ostep.drain()
        sync()
        owall = rdist.max_over_ranks(time.perf_counter() - t1, dev)
        run.overlapped = {"streams": OVERLAP_STREAMS, "wall_s": owall, "ms_per_step": owall * 1e3 / args.steps}

    run.info = info
    run.step = step
    run.dist = ({"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                 "collectives": ["barrier(device_ids)" if use_cuda else "barrier", "all_reduce(MAX) of the per-rank wall time",
                                 f"all_gather of a {len(rdist.STAT_KEYS)}-double stats vector"]}
                if dist.is_available() and dist.is_initialized() else None)
    local = {"records": shard["rows"], "input_bytes": info["input_bytes"], "output_bytes": info["output_bytes"],
             "step_ms": wall * 1e3 / args.steps}
    for k in acc:
        local[k] = acc[k] / max(sampled, 1)
    per_rank = rdist.gather_stats(local, dev)
    agg = rdist.aggregate(per_rank, args.steps, wall)
    return rank, world, wall, per_rank, agg, (gen_cfg, n, num_chunks, desc)


PIPELINE_DEPTH = int(os.environ.get("BENCH_PIPELINE_DEPTH", "3"))      # asynchronous calls in flight before the oldest is settled and freed
STREAMS = 1             # --streams: HIP streams the steps of the MAIN timed region are dealt to round-robin (default: one)
OVERLAP_STREAMS = 3     # --overlap-streams: streams of the second, `overlapped` timed region (0 / 1 = skip it)
SYNC_CALLS = False      # --sync-calls: every step waits for its own call (the pre-RH_ASYNC behaviour)


class Pipeline:
    """Asynchronous rh_decode_device calls, settled PIPELINE_DEPTH submissions later (bounded device memory: each
    unsettled call owns its arena and workspace).  `calls`: one prepared call per stream, used round-robin (--streams:
    consecutive steps are independent batches, so the size pass of one may run beside the tail of the previous emit)."""

    def __init__(self, calls, info):
        import collections
        self.calls = calls if isinstance(calls, (list, tuple)) else [calls]
        self.info, self.ring, self.i = info, collections.deque(), 0

    def _retire(self):
        call, h, want = self.ring.popleft()
        try:
            call.wait(h, want)          # raises on a malformed record, like the synchronous call
            st = None
            if want:
                st = call.stats.as_dict()
                self.info["output_bytes"] = call.output_bytes(h)
        finally:
            call.free(h)
        return st

    def submit(self, want_stats):
        call = self.calls[self.i % len(self.calls)]
        self.i += 1
        self.ring.append((call, call.run(want_stats), want_stats))
        out = []
        while len(self.ring) > PIPELINE_DEPTH:
            st = self._retire()
            if st:
                out.append(st)
        return out

    def drain(self):
        out = []
        while self.ring:
            st = self._retire()
            if st:
                out.append(st)
        return out


def gpu_step_factory(gen_cfg, shard, dev, local_rank):
    import numpy as np
    import torch
    from avrogen import fastgen
    from avrogen.schemas import SCHEMAS
    from pyruhvro_amd import cabi

    n, num_chunks = shard["rows"], shard["chunks"]
    data, offsets = fastgen.generate(gen_cfg, max(n, 1), start=shard["row_lo"])
    if n == 0:
        data, offsets = data[:0], offsets[:1]
    d_data = torch.empty(len(data) + 64, dtype=torch.uint8, device=dev)
    d_data[: len(data)].copy_(torch.from_numpy(data))
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    schema = SCHEMAS[gen_cfg]
    # what a service does at start-up when it wants its first batch at full speed (rh_schema_prebuild): the specialised kernels of
    # this schema, all of them, compiled now if the kernel cache does not hold them -- a call that misses the cache would otherwise
    # run on the generic kernels while they compile in the background (measured separately: `cold_start`)
    t_pre = time.perf_counter()
    info_prebuild_cached = cabi.prebuild(schema)
    stream = torch.cuda.current_stream().cuda_stream
    data_len = int(offsets[-1])
    info = {"input_bytes": data_len, "output_bytes": 0, "prebuild_s": round(time.perf_counter() - t_pre, 3), "prebuild_cached": bool(info_prebuild_cached)}

    # every ctypes argument is built once (cabi.PreparedDeviceDecode): a timed step is the C entry point, the wait and the free
    extra = [torch.cuda.Stream(device=dev) for _ in range(max(STREAMS, 1) - 1)]
    streams = [stream] + [x.cuda_stream for x in extra]

    def prepared(dl, rows, chunks, chunk_rows, asynchronous):
        return [cabi.PreparedDeviceDecode(d_data.data_ptr(), d_off.data_ptr(), dl, rows, schema, chunks, device=local_rank,
                                          stream=sx, kernel=KERNEL, chunk_rows=chunk_rows, asynchronous=asynchronous)
                for sx in (streams if asynchronous else streams[:1])]
    pipe = Pipeline(prepared(data_len, n, num_chunks, shard["chunk_rows"], not SYNC_CALLS), info)

    def step(want_stats=True):
        return pipe.submit(want_stats)
    step.drain = pipe.drain

    def rank_step(world, rank=0, nstreams=1):
        """The step rank `rank` of `world` would run on BASELINE config 5 (one list, whole reference chunks per GPU):
        its rows of THIS list, the list's chunk geometry (rh_opts.chunk_rows).  Rank 0's rows are a prefix of the
        buffers already in HBM.  nstreams > 1: the steps dealt round-robin to that many streams (`overlapped`)."""
        from pyruhvro_amd import dist as rdist
        sh = rdist.strong_shard(n, num_chunks, world, rank)
        assert sh["row_lo"] == 0
        dl = int(offsets[sh["rows"]])
        xs = [torch.cuda.Stream(device=dev) for _ in range(nstreams - 1)]
        calls = [cabi.PreparedDeviceDecode(d_data.data_ptr(), d_off.data_ptr(), dl, sh["rows"], schema, sh["chunks"], device=local_rank,
                                           stream=sx, kernel=KERNEL, chunk_rows=sh["chunk_rows"], asynchronous=not SYNC_CALLS)
                 for sx in [stream] + [x.cuda_stream for x in xs]]
        p = Pipeline(calls, {})

        def f():
            p.submit(False)
        f.drain = p.drain
        f.keepalive = xs
        return f, sh

    def overlapped(nstreams):
        xs = [torch.cuda.Stream(device=dev) for _ in range(nstreams - 1)]
        calls = [cabi.PreparedDeviceDecode(d_data.data_ptr(), d_off.data_ptr(), data_len, n, schema, num_chunks, device=local_rank,
                                           stream=sx, kernel=KERNEL, chunk_rows=shard["chunk_rows"], asynchronous=True)
                 for sx in [stream] + [x.cuda_stream for x in xs]]
        p = Pipeline(calls, {})

        def f():
            p.submit(False)
        f.drain = p.drain
        f.keepalive = xs
        return f
    step.overlapped = overlapped

    def sync_call_ms(reps=30):
        """The same call made synchronously (what a caller that needs the row counts before its next step pays)."""
        c = cabi.PreparedDeviceDecode(d_data.data_ptr(), d_off.data_ptr(), data_len, n, schema, num_chunks,
                                      device=local_rank, stream=stream, kernel=KERNEL, chunk_rows=shard["chunk_rows"])
        for _ in range(3):
            c.free(c.run(False))
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            c.free(c.run(False))
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 1e3 / reps
    step.sync_call_ms = sync_call_ms

    def other_form_ms(single_pass, reps=10):
        """The same call on the OTHER form of the launch sequence -- RH_SINGLE_PASS (one kernel sizes, scans across tiles and
        emits) when the timed steps were two-pass, RH_TWO_PASS when RUHVRO_HIP_SINGLE_PASS=1 made them single-pass -- every call
        carrying the kernels' timestamps, in the same process on the same buffers."""
        c = cabi.PreparedDeviceDecode(d_data.data_ptr(), d_off.data_ptr(), data_len, n, schema, num_chunks, device=local_rank,
                                      stream=stream, kernel=KERNEL, chunk_rows=shard["chunk_rows"], two_pass=not single_pass,
                                      single_pass=single_pass)
        for _ in range(3):
            c.free(c.run(False))
        acc = {"size_kernel_ms": 0.0, "scan_kernel_ms": 0.0, "emit_kernel_ms": 0.0}
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            c.free(c.run(True))
            for key in acc:
                acc[key] += getattr(c.stats, key)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) * 1e3 / reps
        return wall, {key: v / reps for key, v in acc.items()}
    step.other_form_ms = other_form_ms

    def parity_batches():
        """One more call of the TIMED configuration (rh_decode_device + RH_ASYNC on the timed stream, the shard's chunk
        geometry), settled and copied to the host: what `parity_check` compares with the oracle."""
        c = pipe.calls[0]
        h = c.run(False)
        try:
            c.wait(h, False)
            return c.to_host(h)
        finally:
            c.free(h)
    step.parity_batches = parity_batches

    step.rank_step = rank_step
    step.keepalive = (d_data, d_off, extra)
    # set-up, outside every timed region: PIPELINE_DEPTH + 1 calls are in flight at once, each owning an arena (1.9 GB at
    # 10M records) and a workspace from the engine's pools -- take them from the allocator and touch them once here, or
    # the first timed steps pay hipMalloc and first-touch page mapping (measured: 4.07 vs 1.12 ms per step, and 0.79 vs
    # 0.74 ms for the emit launches that write into fresh pages)
    for _ in range(PIPELINE_DEPTH + 2):
        step(False)
    step.drain()
    step()
    first = step.drain()[0]      # also fills output_bytes
    assert first["records"] == n
    info["specialized"] = int(first.get("specialized", 0))
    info["lds_bytes"] = int(first.get("lds_bytes", 0))
    return step, info


def encode_main(args):
    print(json.dumps(encode_line(args)))


def encode_line(args):
    """Secondary line: Arrow -> Avro on the GPU, DEVICE-RESIDENT (rh_encode_device): the Arrow buffers of a batch that
    rh_decode_device left in HBM are read in place and the BinaryArrays of Avro datums are produced in HBM -- `value`
    is rows/s of that, like the decode line.  The host entry point (rh_encode: host batch in, host arrays out, PCIe
    both ways) is reported beside it as `end_to_end`, never as `value`."""
    import ctypes as C
    import numpy as np
    import torch
    import pyruhvro_amd as P
    from avrogen import fastgen
    from avrogen.schemas import SCHEMAS
    from pyruhvro_amd import cabi
    n, k = args.rows, 8
    schema = SCHEMAS["full"]
    kernel = {"auto": 0, "generic": 1, "specialized": 2}[args.kernel]
    P.set_kernel_mode(args.kernel)
    torch.cuda.set_device(0)
    data, offsets = fastgen.generate("full", n)
    d_data = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda:0")
    d_data[: len(data)].copy_(torch.from_numpy(data))
    d_off = torch.from_numpy(offsets.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    src = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), n, schema, 1, device=0, stream=stream)
    arrow_bytes = src.output_bytes
    view = src.export(0)                       # ArrowDeviceArray: device pointers into the decode result's arena
    sch = cabi.schema_struct(schema)

    def step(want_stats):
        enc = cabi.encode_device(C.addressof(view.array), C.addressof(sch), schema, k, device=0, stream=stream,
                                 want_stats=want_stats, kernel=kernel)
        st, ob = enc.stats, enc.output_bytes
        enc.free()
        return st, ob

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    acc = {"size_kernel_ms": 0.0, "scan_kernel_ms": 0.0, "emit_kernel_ms": 0.0}
    sampled = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        want = i % max(args.stats_every, 1) == 0
        st, ob = step(want)
        if want:
            sampled += 1
            for key in acc:
                acc[key] += st[key]
            spec = st.get("specialized")
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    for key in acc:
        acc[key] /= max(sampled, 1)
    total = ob - 4 * (n + k)                                   # Avro bytes (the result's exact bytes = i32 offsets + datums)
    if not os.environ.get("RUHVRO_HIP_VARIANT"):               # (a timing-only kernel variant of an A/B script may produce other bytes: its line is not a result)
        assert total == int(offsets[-1]), "re-encoded bytes differ in size from the generator's datums"
    # parity evidence for this line's own configuration: one more call, its BinaryArrays copied to the host, datums and offsets
    # against the generator's payload (the generator writes the reference's single-block form: encode(decode(x)) == x)
    parity = None
    if not os.environ.get("RUHVRO_HIP_VARIANT"):
        tp = time.perf_counter()
        enc = cabi.encode_device(C.addressof(view.array), C.addressof(sch), schema, k, device=0, stream=stream, kernel=kernel)
        arrays = enc.to_host()
        pos, same = 0, len(arrays) == k
        for a in arrays:
            o = np.frombuffer(a.buffers()[1], dtype=np.int32, count=len(a) + 1)
            dd = np.frombuffer(a.buffers()[2], dtype=np.uint8, count=int(o[-1]))
            same = same and np.array_equal(o.astype(np.uint64), offsets[pos: pos + len(a) + 1] - offsets[pos]) \
                and np.array_equal(dd, data[int(offsets[pos]): int(offsets[pos + len(a)])])
            pos += len(a)
        same = bool(same and pos == n)
        del arrays
        enc.free()
        parity = {"config": f"rh_encode_device on the timed stream, {n} rows, num_chunks={k}, every BinaryArray copied to the host after the timed region",
                  "rows": n, "chunks": k, "result": "identical" if same else "DIFFERENT", "check_s": round(time.perf_counter() - tp, 2)}
        assert same, "re-encoded datums differ from the generator's"
    alg = arrow_bytes + total + 4 * (n + k)                 # Arrow bytes in + Avro bytes out + i32 offsets out
    emit_ms = acc["emit_kernel_ms"]
    kern_ms = acc["size_kernel_ms"] + acc["scan_kernel_ms"] + emit_ms
    emit_name = "rh_espec_emit" if spec else "rh_e_emit"
    et = stamped_traffic(schema, encode=True).get(emit_name)
    enc_traffic = et["hbm_bytes"] if et and n == 2_000_000 else None      # measured on the 2M-row launch only
    # the host entry point on the same batch (PCIe-inclusive), best of 3
    batch = src.to_host()[0]
    P.serialize_record_batch(batch, schema, k)
    best, hst = None, None
    for _ in range(3):
        t = time.perf_counter()
        out, st = P.serialize_record_batch_with_stats(batch, schema, k)
        w = time.perf_counter() - t
        del out
        if best is None or w < best:
            best, hst = w, st
    return ({
        "metric": "Arrow rows/sec -> Avro (rh_encode_device: Arrow buffers and Avro datums resident in HBM)",
        "value": n * args.steps / wall, "unit": "rows/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic (the decode workload's records, decoded to Arrow on the GPU and left in HBM, then re-encoded)",
        "config": {"workload": f"{n} rows of the generate_avro.py schema, num_chunks={k}, Arrow -> Avro (SURVEY 8f N1; not the headline metric)",
                   "arrow_bytes_in": int(arrow_bytes), "avro_bytes_out": int(total),
                   "kernel_ms": {"e_size": acc["size_kernel_ms"], "k_scan": acc["scan_kernel_ms"], "e_emit": emit_ms},
                   "rows_per_s_kernels_only": n / (kern_ms * 1e-3) if kern_ms else 0.0,
                   "kernel_form": "schema-specialised" if spec else "generic interpreter"},
        "roofline": {"bound": "hbm", "kernel": emit_name, "achieved": alg / (emit_ms * 1e-3) / 1e9 if emit_ms else 0.0,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg / (emit_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if emit_ms else 0.0,
                     "traffic": enc_traffic, "traffic_rows": 2_000_000 if enc_traffic else None,
                     "algorithmic_bytes_per_launch": int(alg), "bytes_per_record": alg / n, "avg_launch_ms": emit_ms,
                     "timed_launches": sampled,
                     "path_frac": alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if kern_ms else 0.0},
        "parity_check": parity,
        "end_to_end": {"value": n / best, "unit": "rows/s", "wall_ms": best * 1e3,
                       "what": "rh_encode: host RecordBatch in -> host BinaryArrays out (PCIe both ways), best of 3",
                       "stage_ms": {key: round(float(hst[key]), 3) for key in ("h2d_ms", "size_kernel_ms", "scan_kernel_ms", "emit_kernel_ms", "d2h_ms", "total_ms")}}})


def other_configs(local_rank: int = 0, steps: int = 60):
    """BASELINE configs 2 and 3 (and the full schema at 1M records) and the Arrow -> Avro direction, in the DEFAULT line
    so that the driver's run carries them: for each, the pipelined step, the synchronous call, the kernels' own times
    (every 4th step carries timestamps) and the same steps dealt to OVERLAP_STREAMS streams.  At 1M records the input
    (16-120 MB) lives in the 256 MB Infinity Cache and a call is 3-4 generations of resident waves: these are
    launch / latency statements, not HBM-bandwidth ones."""
    import argparse
    import torch
    from pyruhvro_amd import dist as rdist
    out = {}
    dev = torch.device("cuda", local_rank)
    for name in ("flat4_1m", "cfg3_1m", "full1m"):
        gen_cfg, n, k, desc = WORKLOADS[name]
        shard = rdist.strong_shard(n, k, 1, 0)
        step, info = gpu_step_factory(gen_cfg, shard, dev, local_rank)

        def timed(f, nsteps, stats_every=0):
            acc, cnt = {"size_kernel_ms": 0.0, "scan_kernel_ms": 0.0, "emit_kernel_ms": 0.0}, 0
            for _ in range(6):
                f(False)
            f.drain()
            torch.cuda.synchronize()
            t = time.perf_counter()
            sts = []
            for i in range(nsteps):
                sts += f(bool(stats_every) and i % stats_every == stats_every - 1) or []
            sts += f.drain()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t) * 1e3 / nsteps
            for st in sts:
                cnt += 1
                for key in acc:
                    acc[key] += st.get(key, 0.0)
            return wall, {key: v / max(cnt, 1) for key, v in acc.items()}
        ms, kern = timed(step, steps, 4)
        ov = step.overlapped(max(OVERLAP_STREAMS, 2))
        ov_step = lambda w, _f=ov: _f()          # noqa: E731
        ov_step.drain = ov.drain
        oms, _ = timed(ov_step, steps)
        alg = info["input_bytes"] + 8 * n + info["output_bytes"]
        out[name] = {"workload": desc, "records": n, "ms_per_step": ms, "records_per_s": n / (ms * 1e-3),
                     "sync_call_ms": step.sync_call_ms(), "overlapped_ms_per_step": oms,
                     "kernel_ms": {"k_size": kern["size_kernel_ms"], "k_scan": kern["scan_kernel_ms"], "k_emit": kern["emit_kernel_ms"]},
                     "emit_frac": alg / (kern["emit_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS if kern["emit_kernel_ms"] > 0 else 0.0}
        del step, ov, ov_step
    # round 6: the walks off the benchmark generator's distribution, wide schemas, and the generic kernels (`workload_line`)
    for name, (wl, n, kw) in {
        "full_realistic_10m": ("full_realistic", 10_000_000, {"parity_max": 2_000_000}),
        "full_realistic_nogiant_10m": ("full_realistic_nogiant", 10_000_000, {"parity": False}),      # (the same without the > 8,191-item arrays: what one serial item chase per giant record costs the line above)
        "full_skewed_10m": ("full_skewed", 10_000_000, {"parity_max": 2_000_000}),
        "wide200_1m": ("wide200", 1_000_000, {}),
        "full10m_generic": ("full", 10_000_000, {"kernel": "generic", "reps": 8, "parity_max": 2_000_000}),
        "cfg3_1m_generic": ("cfg3", 1_000_000, {"kernel": "generic"}),
    }.items():
        try:
            out[name] = workload_line(wl, n, **kw)
        except Exception as e:      # (one line failing must not take the contract line with it)
            out[name] = {"workload": wl, "failed": repr(e)[:300]}
        torch.cuda.empty_cache()
    enc = encode_line(argparse.Namespace(rows=2_000_000, steps=5, warmup=2, kernel="auto", stats_every=STATS_EVERY))
    out["encode_2m_rows"] = {"workload": enc["config"]["workload"], "ms_per_step": enc["ms_per_step"], "rows_per_s": enc["value"],
                             "kernel_ms": enc["config"]["kernel_ms"], "emit_frac": enc["roofline"]["frac"],
                             "traffic": enc["roofline"]["traffic"], "end_to_end_rows_per_s": enc["end_to_end"]["value"],
                             "parity_check": enc.get("parity_check")}
    return out


def config5_projection(step, ms_per_step_1gpu: float, num_chunks: int, reps: int = 40, overlapped_1gpu=None):
    """BASELINE config 5 without an 8-GPU node: the step ONE rank would run when the same list is dealt to g GPUs
    (rank 0's chunks of it, the list's chunk geometry via rh_opts.chunk_rows), timed on this GPU like the main loop
    (asynchronous calls, settled PIPELINE_DEPTH steps later, one sync at the end).  Ranks are independent (no data-path
    collective), so the job's step at g GPUs is the slowest rank's step: implied strong-scaling efficiency =
    (ms_per_step at 1 GPU / g) / that.  A PROJECTION from one GPU -- the RCCL barrier and the MAX over ranks of the
    real run are not in it.  `overlapped`: the same with the steps dealt to OVERLAP_STREAMS streams, against the
    1-GPU figure measured the same way."""
    import torch
    out = {"what": "per-rank step of the 10M-record list over g GPUs, measured on this one GPU; a projection, not an N-GPU run",
           "ms_per_step_1gpu": ms_per_step_1gpu, "g": {}}
    if overlapped_1gpu:
        out["overlapped_streams"] = overlapped_1gpu["streams"]
        out["overlapped_ms_per_step_1gpu"] = overlapped_1gpu["ms_per_step"]

    def timed(f):
        for _ in range(6):
            f()
        f.drain()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            f()
        f.drain()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 1e3 / reps

    for g in (2, 4, 8):
        if g > num_chunks:
            continue
        f, sh = step.rank_step(g)
        ms = timed(f)
        e = {"records_per_rank": sh["rows"], "chunks_per_rank": sh["chunks"], "ms_per_step": ms,
             "ideal_ms": ms_per_step_1gpu / g, "implied_efficiency": (ms_per_step_1gpu / g) / ms if ms > 0 else 0.0,
             "implied_records_per_s": sh["rows"] * g / (ms * 1e-3) if ms > 0 else 0.0}
        if overlapped_1gpu:
            f2, _ = step.rank_step(g, 0, overlapped_1gpu["streams"])
            ms2 = timed(f2)
            base = overlapped_1gpu["ms_per_step"]
            e["overlapped"] = {"ms_per_step": ms2, "ideal_ms": base / g, "implied_efficiency": (base / g) / ms2 if ms2 > 0 else 0.0,
                               "implied_records_per_s": sh["rows"] * g / (ms2 * 1e-3) if ms2 > 0 else 0.0}
        out["g"][str(g)] = e
    return out


def main(argv=None):
    args = parse_args(argv)
    if args.direction == "encode":
        return encode_main(args)
    global KERNEL, SYNC_CALLS, STREAMS, OVERLAP_STREAMS
    KERNEL = {"auto": 0, "generic": 1, "specialized": 2}[args.kernel]
    SYNC_CALLS = bool(args.sync_calls)
    STREAMS = max(1, int(args.streams))
    OVERLAP_STREAMS = int(args.overlap_streams)
    import torch  # noqa: F401  (first: our library must share torch's HIP runtime)
    from avrogen.schemas import SCHEMAS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    rank, world, wall, per_rank, agg, (gen_cfg, n, num_chunks, desc) = run(args, gpu_step_factory, "nccl")
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if world > 1:
            dist.barrier(device_ids=[int(os.environ.get("LOCAL_RANK", "0"))])
        dist.destroy_process_group()          # clean teardown of the RCCL communicator on every rank
    if rank != 0:
        return
    # roofline of the dominant kernel, on the slowest rank's launch: SURVEY 8(d) B_in + 8 (u64 offset) + B_out per record
    rs = max(per_rank, key=lambda r: r["emit_kernel_ms"])
    r0 = rs
    b_in, b_out = rs["input_bytes"], rs["output_bytes"]
    alg_bytes = b_in + 8 * rs["records"] + b_out
    emit_ms = rs["emit_kernel_ms"]
    achieved = alg_bytes / (emit_ms * 1e-3) / 1e9 if emit_ms > 0 else 0.0
    path_ms = r0["size_kernel_ms"] + r0["scan_kernel_ms"] + r0["emit_kernel_ms"]
    strong = args.scaling == "strong"
    per_gpu = [{"rank": i, "records": int(r["records"]), "records_per_s": r["records"] / (r["step_ms"] * 1e-3) if r["step_ms"] else 0.0,
                "kernel_ms": {"k_size": r["size_kernel_ms"], "k_scan": r["scan_kernel_ms"], "k_emit": r["emit_kernel_ms"]},
                "emit_alg_GBps": (r["input_bytes"] + 8 * r["records"] + r["output_bytes"]) / (r["emit_kernel_ms"] * 1e-3) / 1e9
                if r["emit_kernel_ms"] > 0 else 0.0} for i, r in enumerate(per_rank)]
    spec = bool(getattr(run, "info", {}).get("specialized"))
    # the single-pass form (one kernel sizes, scans across tiles and emits): no size / scan launches in the timed steps
    single = spec and rs["size_kernel_ms"] == 0.0 and rs["emit_kernel_ms"] > 0 and gen_cfg != "flat4"
    emit_kernel = "rh_spec_fused" if single else "rh_spec_emit" if spec else "rh_k_emit"
    size_kernel = "rh_spec_size" if spec else "rh_k_size"
    shard_whole = args.workload == "full10m" and not args.records
    # HBM bytes per launch from the stamped PMC passes: only for the launch they were measured on (1 GPU, 10M records)
    traffic = stamped_traffic(SCHEMAS[gen_cfg]) if world == 1 and shard_whole else {}
    size_ms, scan_ms = r0["size_kernel_ms"], r0["scan_kernel_ms"]
    kernels = {}
    for name, ms, alg in (((emit_kernel, emit_ms, alg_bytes),) if single else
                          ((size_kernel, size_ms, b_in + 8 * rs["records"]), ("rh_k_scan", scan_ms, 0), (emit_kernel, emit_ms, alg_bytes))):
        t = traffic.get(name)
        k = {"avg_launch_ms": ms, "algorithmic_bytes_per_launch": int(alg),
             "alg_GBps": alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0}
        if t:
            k.update({"hbm_read_bytes": t["hbm_read_bytes"], "hbm_write_bytes": t["hbm_write_bytes"],
                      "hbm_read_GBps": t["hbm_read_bytes"] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                      # the north star's "40 % of HBM read bandwidth", stated per kernel
                      "read_frac": t["hbm_read_bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if ms > 0 else 0.0,
                      "hbm_GBps": t["hbm_bytes"] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0})
        kernels[name] = k
    both_ms = size_ms + emit_ms
    path_kernels = (emit_kernel,) if single else (size_kernel, emit_kernel)
    read_bytes = sum(traffic[k]["hbm_read_bytes"] for k in path_kernels) if all(k in traffic for k in path_kernels) else None
    all_bytes = sum(traffic[k]["hbm_bytes"] for k in path_kernels) if all(k in traffic for k in path_kernels) else None
    roofline = {"bound": "hbm", "kernel": emit_kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic[emit_kernel]["hbm_bytes"] if emit_kernel in traffic else None,
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "bytes_per_record": alg_bytes / max(rs["records"], 1), "avg_launch_ms": emit_ms,
                "timed_launches": args.steps // max(args.stats_every, 1),
                # the whole path (k_size + k_scan + k_emit) against the same algorithmic bytes, and the north star's own
                # figure: HBM READ bandwidth of the two passes together (rocprofv3 FETCH_SIZE of both / their time)
                "path_achieved": alg_bytes / (path_ms * 1e-3) / 1e9 if path_ms > 0 else 0.0,
                "path_frac": alg_bytes / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if path_ms > 0 else 0.0,
                "read_GBps": read_bytes / (both_ms * 1e-3) / 1e9 if read_bytes and both_ms > 0 else None,
                "read_frac": read_bytes / (both_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if read_bytes and both_ms > 0 else None,
                # every HBM byte the path's kernels move (PMC, read + write) / their time: the single-pass form reads each
                # record ONCE, so its READ rate is low by construction -- this is the figure that says how busy HBM is
                "hbm_GBps": all_bytes / (both_ms * 1e-3) / 1e9 if all_bytes and both_ms > 0 else None,
                "hbm_frac": all_bytes / (both_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if all_bytes and both_ms > 0 else None,
                "form": ("single pass: rh_spec_fused sizes a tile, scans across the tiles of its chunk (look-back) and emits out of the same LDS window; "
                         "`frac` is the whole path's" if single else "two passes: size pass, scan + layout, emit pass; `frac` is the emit pass's"),
                "path_traffic": (sum(traffic[n_]["hbm_bytes"] for n_ in (path_kernels + (("rh_k_layout", "rh_k_publish") if single else
                                                                                      ("rh_k_scan_layout", "rh_k_publish"))) if n_ in traffic)
                                 if all(k in traffic for k in path_kernels) else None),
                "kernels": kernels}
    out = {
        "metric": "Avro records/sec -> Arrow (direct decode, input and output resident in HBM)",
        "value": agg["records_per_s"],
        "unit": "records/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic (seeded generator with scripts/generate_avro.py's distributions; Faker unavailable)",
        "config": {"workload": desc + (f", list-partitioned over {world} GPUs (BASELINE.json config 5 shape)" if strong and world > 1 else ""),
                   "records_total": int(sum(r["records"] for r in per_rank)), "records_per_gpu": [int(r["records"]) for r in per_rank],
                   "num_chunks": num_chunks, "schema": gen_cfg,
                   "input_bytes_total": int(sum(r["input_bytes"] for r in per_rank)),
                   "arrow_bytes_total": int(sum(r["output_bytes"] for r in per_rank)),
                   "parallelism": (f"{world} GPUs x whole reference chunks of ONE list (rh_shard_chunks), no data-path collective" if strong
                                   else f"{world} x independent 10M-record shard (weak), no data-path collective"),
                   "per_gpu": per_gpu,
                   "kernel_ms": dict({"k_size": r0["size_kernel_ms"], "k_scan": r0["scan_kernel_ms"], "k_emit": r0["emit_kernel_ms"]},
                                     **({"k_fused": r0["emit_kernel_ms"]} if single else {})),
                   "kernel_form": ("schema-specialised, single pass (rh_spec_fused; `k_emit` is that kernel)" if single else
                                   "schema-specialised" if getattr(run, "info", {}).get("specialized") else "generic interpreter"),
                   "emit_lds_bytes_per_workgroup": getattr(run, "info", {}).get("lds_bytes", 0),
                   "calls": ("synchronous: every step waits for its own call" if SYNC_CALLS else
                             f"RH_ASYNC on {STREAMS} stream(s), settled {PIPELINE_DEPTH} steps later (rh_device_result_wait), everything inside the timed region"),
                   "sync_call_ms": run.step.sync_call_ms() if hasattr(run.step, "sync_call_ms") else None,
                   "path_kernel_ms": path_ms,
                   "path_alg_GBps": alg_bytes / (path_ms * 1e-3) / 1e9 if path_ms > 0 else 0.0},
        "roofline": roofline,
    }
    if spec and world == 1 and shard_whole and hasattr(run.step, "other_form_ms"):
        wall2, k2 = run.step.other_form_ms(single_pass=not single)
        p2 = k2["size_kernel_ms"] + k2["scan_kernel_ms"] + k2["emit_kernel_ms"]
        if single:
            out["two_pass"] = {
                "what": "the same call forced onto the two-pass form (RH_TWO_PASS), 10 synchronous calls with kernel timestamps, same process and buffers",
                "sync_call_ms": wall2, "kernel_ms": {"k_size": k2["size_kernel_ms"], "k_scan": k2["scan_kernel_ms"], "k_emit": k2["emit_kernel_ms"]},
                "emit_frac": alg_bytes / (k2["emit_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS if k2["emit_kernel_ms"] > 0 else None,
                "path_frac": alg_bytes / (p2 * 1e-3) / 1e9 / HBM_PEAK_GBPS if p2 > 0 else None}
        elif k2["size_kernel_ms"] == 0.0 and k2["emit_kernel_ms"] > 0:
            tf = traffic.get("rh_spec_fused")
            out["single_pass"] = {
                "what": ("the same call on the opt-in single-pass form (RH_SINGLE_PASS: rh_spec_fused sizes a tile, scans across the tiles of its "
                         "chunk by look-back and emits out of the same LDS window -- every record read from HBM once), 10 synchronous calls with "
                         "kernel timestamps, same process and buffers"),
                "sync_call_ms": wall2, "kernel_ms": {"k_fused": k2["emit_kernel_ms"]},
                "path_frac": alg_bytes / (k2["emit_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "traffic": tf["hbm_bytes"] if tf else None, "traffic_vs_algorithmic": tf["hbm_bytes"] / alg_bytes if tf else None,
                "hbm_frac": tf["hbm_bytes"] / (k2["emit_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS if tf else None}
    if getattr(run, "dist", None):
        out["dist"] = run.dist
    if getattr(run, "overlapped", None):
        ov = run.overlapped
        total = sum(r["records"] for r in per_rank) * args.steps
        out["overlapped"] = {
            "what": (f"the same {args.steps} steps dealt round-robin to {ov['streams']} HIP streams (independent batches: one step's size pass "
                     "runs beside another's emit pass); second timed region, same barriers; not the headline because kernels that "
                     "share the chip have no per-kernel duration to price against the roofline"),
            "streams": ov["streams"], "ms_per_step": ov["ms_per_step"], "value": total / ov["wall_s"] if ov["wall_s"] > 0 else 0.0,
            "unit": "records/s",
            "path_alg_GBps": alg_bytes / (ov["ms_per_step"] * 1e-3) / 1e9 if ov["ms_per_step"] > 0 and world == 1 else None,
            "path_frac": alg_bytes / (ov["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS if ov["ms_per_step"] > 0 and world == 1 else None}
    if world == 1 and shard_whole and hasattr(run.step, "rank_step") and not args.no_projection:
        out["config5_projection"] = config5_projection(run.step, wall * 1e3 / args.steps, num_chunks,
                                                       overlapped_1gpu=getattr(run, "overlapped", None))
    if not args.no_cpu_baseline and world == 1:   # the CPU port is timed at N=1 only (rank 0's host cores)
        n_cpu = min(args.cpu_sample, n) if args.cpu_sample else n
        got = None
        if n_cpu == n and hasattr(run.step, "parity_batches"):       # the oracle decodes the whole workload: check the timed configuration against it
            got = (run.step.parity_batches(),
                   f"rh_decode_device + RH_ASYNC on the timed stream, {n} records, num_chunks={num_chunks}, kernel form of the timed steps, "
                   "every chunk copied to the host after the timed regions")
        out["cpu_baseline"], parity = cpu_baseline(gen_cfg, SCHEMAS[gen_cfg], n_cpu, num_chunks, got)
        del got
        if parity:
            out["parity_check"] = parity
    if not args.no_end_to_end and world == 1:
        out["end_to_end"] = end_to_end(gen_cfg, SCHEMAS[gen_cfg], n, num_chunks)
    if not args.no_cold_start and world == 1:
        out["cold_start"] = cold_start(gen_cfg, SCHEMAS[gen_cfg])
    if not args.no_other_configs and not args.no_end_to_end and world == 1 and shard_whole:     # (profiler passes give --no-end-to-end)
        run.step = None            # (the 10M-record buffers of the main workload are not needed any more)
        out["other_configs"] = other_configs(0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
