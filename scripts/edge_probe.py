"""Edges of the reference's contract, timed: num_chunks = n (deserialize.rs:53-55 allows it), one 64 MB array record among
small ones (fast_decode.rs:703-719), against the CPU port on the same input."""
import json, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
from avrogen.encoder import zigzag
from oracle import c_walker
from pyruhvro_amd import cabi
from arrow_compare import assert_batches_identical

which = sys.argv[1:] or ["chunks", "giant"]
if "chunks" in which:
    for name, n in (("full", 10_000), ("full", 100_000)):
        data, offsets = fastgen.generate(name, n)
        S = SCHEMAS[name]
        for k in (n,):
            t = time.perf_counter(); got, st = cabi.decode_packed(data, offsets, S, k, want_stats=True); w = time.perf_counter() - t
            t = time.perf_counter(); got2 = cabi.decode_packed(data, offsets, S, k); w2 = time.perf_counter() - t
            t = time.perf_counter(); exp = c_walker.decode_packed(c_walker.CompiledSchema(S), data, offsets, k, threaded=False); wc = time.perf_counter() - t
            assert len(got) == len(exp) == k
            for i in list(range(0, k, max(1, k // 200))) + [k - 1]:
                assert_batches_identical(got[i], exp[i])
            print(f"{name} n={n} num_chunks={k}: GPU call {w * 1e3:.1f} ms (again {w2 * 1e3:.1f} ms), engine total {st['total_ms']:.1f} ms, output {st['output_bytes']} B; "
                  f"CPU port (1 thread, materialised) {wc * 1e3:.1f} ms", flush=True)
            del got, got2, exp
if "giant" in which:
    # one record whose `emails` array holds 2M strings (~64 MB of payload) in the middle of 100k ordinary records
    n = 100_000
    data, offsets = fastgen.generate("full", n)
    recs = fastgen.split(data, offsets)
    items = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2_000_000
    body = bytearray()
    body += b"\x00"                     # name: null
    body += b"\x00"                     # age: null
    body += zigzag(items)               # emails: one block of `items` strings
    item = zigzag(29) + b"x" * 29
    body += item * items
    body += b"\x00"                     # end of array
    body += b"\x00"                     # address: null
    body += b"\x00"                     # phone_numbers: empty map
    body += b"\x00"                     # preferences: null
    body += b"\x00"                     # status: null
    body += zigzag(1_750_000_000) + zigzag(1)
    recs[n // 2] = bytes(body)
    S = SCHEMAS["full"]
    print(f"giant record: {len(body) / 1e6:.1f} MB among {n} records", flush=True)
    import pyruhvro_amd as P
    t = time.perf_counter(); exp = c_walker.decode_threaded(recs, S, 8); wc = time.perf_counter() - t
    for mode in ("generic", "specialized"):
        old = P.set_kernel_mode(mode)
        t = time.perf_counter(); got = P.deserialize_array_threaded(recs, S, 8); w = time.perf_counter() - t
        t = time.perf_counter(); got, st = P.deserialize_array_threaded_with_stats(recs, S, 8); w2 = time.perf_counter() - t
        P.set_kernel_mode(old)
        print(f"    k_size {st['size_kernel_ms']:.1f} ms, k_emit {st['emit_kernel_ms']:.1f} ms", flush=True)
        for g, e in zip(got, exp):
            assert_batches_identical(g, e)
        print(f"  {mode}: GPU call {w * 1e3:.1f} ms, again {w2 * 1e3:.1f} ms; CPU port 8 threads {wc * 1e3:.1f} ms", flush=True)
