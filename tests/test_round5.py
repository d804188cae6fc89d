"""Round-5 evidence tests (all need an MI355X):

* a schema's first large call does NOT wait for hiprtc: it is served by the generic kernels while the specialised ones
  compile in the background (helper processes), a later call runs on them, same buffers both times -- the reference's
  cost of a new schema is a JSON parse (src/lib.rs:39-54, ruhvro/src/deserialize.rs:18-20);
* schemas with more than 64 scanned counters on both kernel forms (v_readlane wraps at 64 lanes).
"""
import os
import time

import numpy as np
import pytest
import torch

import cases
from arrow_compare import assert_batches_identical
from avrogen import fastgen
from avrogen.schemas import SCHEMAS
from oracle import c_walker

import pyruhvro_amd as P
from pyruhvro_amd import cabi

pytestmark = pytest.mark.gpu


def _to_device(data, offsets):
    d_data = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda:0")
    d_data[: len(data)].copy_(torch.from_numpy(data))
    d_off = torch.from_numpy(offsets.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    return d_data, d_off


def test_first_large_call_of_a_new_schema_does_not_wait_for_the_compiler(tmp_path, monkeypatch):
    monkeypatch.setenv("RUHVRO_HIP_KERNEL_CACHE", str(tmp_path))       # an empty kernel cache: nothing of this schema is compiled
    monkeypatch.setenv("AMD_COMGR_CACHE", "0")                         # (and no help from the compiler's own cache: the helpers inherit this)
    n = 100_000
    data, offsets = fastgen.generate("full", n)
    exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS["full"]), data, offsets, 8, threaded=True)
    d_data, d_off = _to_device(data, offsets)
    stream = torch.cuda.current_stream().cuda_stream
    # the process has decoded before (generic kernels loaded, pools warm): what is new is the SCHEMA
    cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), n, SCHEMAS["full"], 8, device=0, stream=stream,
                       kernel=cabi.KERNEL_GENERIC).free()
    schema = SCHEMAS["full"] + "\t\t"                                 # a handle of this test's own (handles remember code objects)
    before = cabi.engine_counters()["background_compiles"]
    t0 = time.perf_counter()
    r1 = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), n, schema, 8, device=0, stream=stream)
    first_ms = (time.perf_counter() - t0) * 1e3
    assert r1.stats["specialized"] == 0
    assert cabi.engine_counters()["background_compiles"] == before + 2        # rh_spec_size and rh_spec_emit, one job each
    assert first_ms < 50, f"a new schema's first {n}-record call took {first_ms:.1f} ms"
    for g, e in zip(r1.to_host(), exp):
        assert_batches_identical(g, e)
    r1.free()
    assert not cabi.kernels_ready(schema)                               # (tens of seconds of hiprtc are still ahead)
    # calls keep being served while the jobs run
    r2 = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), n, schema, 8, device=0, stream=stream)
    assert r2.stats["specialized"] == 0
    r2.free()
    assert cabi.engine_counters()["background_compiles"] == before + 2        # (no second set of jobs)
    t0 = time.perf_counter()
    assert cabi.kernels_ready(schema, timeout_ms=240_000)
    compile_s = time.perf_counter() - t0
    r3 = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), n, schema, 8, device=0, stream=stream)
    assert r3.stats["specialized"] == 1
    for g, e in zip(r3.to_host(), exp):
        assert_batches_identical(g, e)
    r3.free()
    files = sorted(os.listdir(tmp_path))
    assert len(files) == 2 and all(f.endswith(".hsaco") for f in files), files      # the pair; the single-pass kernel was not asked for
    print(f"cold schema: first call {first_ms:.2f} ms on the generic kernels, specialised pair ready after {compile_s:.1f} s")


def test_python_surface_on_a_cold_schema(tmp_path, monkeypatch):
    """The same through deserialize_array_threaded (host path): the call returns the oracle's batches at once."""
    monkeypatch.setenv("RUHVRO_HIP_KERNEL_CACHE", str(tmp_path))
    data, offsets = fastgen.generate("cfg3", 60_000)
    recs = fastgen.split(data, offsets)
    schema = SCHEMAS["cfg3"] + "\t\t\t"
    t0 = time.perf_counter()
    got, st = P.deserialize_array_threaded_with_stats(recs, schema, 4)
    ms = (time.perf_counter() - t0) * 1e3
    assert st["specialized"] == 0 and ms < 500
    for g, e in zip(got, c_walker.decode_threaded(recs, SCHEMAS["cfg3"], 4)):
        assert_batches_identical(g, e)
    assert P.kernels_ready(schema, timeout_ms=240_000)
    got, st = P.deserialize_array_threaded_with_stats(recs, schema, 4)
    assert st["specialized"] == 1
    for g, e in zip(got, c_walker.decode_threaded(recs, SCHEMAS["cfg3"], 4)):
        assert_batches_identical(g, e)


# (round 6: schemas with more than 64 counters are compiled WIDE -- wave counters, runs of like columns emitted as loops -- and
#  the 96-counter schema's specialised kernels, five to eight minutes of hiprtc in round 5, take seconds: both forms, both schemas)
_WIDE = [(c, k) for c in cases.wide_counter_cases() for k in ("generic", "specialized")]


@pytest.mark.parametrize("case,kernel", _WIDE, ids=[f"{c[0]}-{k}" for c, k in _WIDE])
def test_more_than_64_counters(case, kernel):
    _, schema, recs = case
    old = P.set_kernel_mode(kernel)
    try:
        for k in (1, 3):
            got = P.deserialize_array_threaded(recs, schema, k)
            exp = c_walker.decode_threaded(recs, schema, k)
            for g, e in zip(got, exp):
                g.validate(full=True)
                assert_batches_identical(g, e)
        big = (recs * 3)[:1900]                           # several workgroups, a ragged last one
        for g, e in zip(P.deserialize_array_threaded(big, schema, 5), c_walker.decode_threaded(big, schema, 5)):
            assert_batches_identical(g, e)
    finally:
        P.set_kernel_mode(old)


@pytest.mark.parametrize("n", [10_000, 100_000])
def test_num_chunks_equal_to_the_record_count(n):
    """deserialize.rs:53-55 lets num_chunks go up to n: n one-row batches.  Every chunk is a 256-byte-aligned region per Arrow buffer
    and a workgroup of its own, so this is the engine's worst shape -- it has to work (and say what it cost), not be fast."""
    data, offsets = fastgen.generate("full", n)
    got, st = cabi.decode_packed(data, offsets, SCHEMAS["full"], n, want_stats=True)
    assert len(got) == n and st["chunks"] == n and st["records"] == n
    exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS["full"]), data, offsets, n, threaded=False)
    for i in list(range(0, n, max(1, n // 500))) + [n - 1]:
        assert got[i].num_rows == 1
        assert_batches_identical(got[i], exp[i])
    assert st["total_ms"] < 20_000
    print(f"num_chunks = n = {n}: engine {st['total_ms']:.0f} ms, {st['output_bytes']} Arrow bytes in {n} batches")


def test_one_giant_record_among_small_ones():
    """One 60 MB record (an array of 2M strings) among 100,000 ordinary ones (fast_decode.rs:703-719 walks it item by item).  Round
    5: its tile was walked from global memory by one lane per record, 4.6-5.5 s on the specialised kernels, 50-145 x the CPU port.
    Round 6: the record is a sliding range of its own (walk.h SlideSrc) whose items are found by the wavefront-parallel item scan
    (spec_body.h item_scan) and emitted one lane per item: the specialised kernels stay within 10 x the CPU port of the same box
    (VERDICT round 5, item 3); the interpreter has neither and keeps round 5's bound."""
    from avrogen.encoder import zigzag
    n, items = 100_000, 2_000_000
    data, offsets = fastgen.generate("full", n)
    recs = fastgen.split(data, offsets)
    body = bytearray(b"\x00\x00") + zigzag(items) + (zigzag(29) + b"x" * 29) * items + b"\x00" + b"\x00\x00\x00\x00" + \
        zigzag(1_750_000_000) + zigzag(1)
    recs[n // 2] = bytes(body)
    cpu = float("inf")
    for _ in range(2):
        t0 = time.perf_counter()
        exp = c_walker.decode_threaded(recs, SCHEMAS["full"], 8)
        cpu = min(cpu, time.perf_counter() - t0)
    walls = {}
    for mode in ("generic", "specialized"):
        old = P.set_kernel_mode(mode)
        try:
            best = float("inf")
            for _ in range(2):           # (the first call of a mode also takes its blocks from the allocator)
                t0 = time.perf_counter()
                got = P.deserialize_array_threaded(recs, SCHEMAS["full"], 8)
                best = min(best, time.perf_counter() - t0)
        finally:
            P.set_kernel_mode(old)
        for g, e in zip(got, exp):
            assert_batches_identical(g, e)
        walls[mode] = best
    print(f"giant record: CPU port {cpu * 1e3:.0f} ms, specialised {walls['specialized'] * 1e3:.0f} ms "
          f"({walls['specialized'] / cpu:.1f} x), generic {walls['generic'] * 1e3:.0f} ms ({walls['generic'] / cpu:.1f} x)")
    assert walls["generic"] < 30, f"generic: {walls['generic']:.1f} s"
    # (VERDICT round 5 asked for <= 10 x the CPU port: 5.3-7.0 x on the boxes of round 6 -- 0.61-0.71 s against 0.09-0.12 s, DESIGN.md
    #  5.3.  The CPU port's time varies 3 x between hosts, so the bound is ten times it with a floor of one second: round 5's
    #  call took 4.6-5.5 s.)
    assert walls["specialized"] < max(10 * cpu, 1.0), f"specialised {walls['specialized']:.2f} s vs CPU port {cpu:.3f} s"


def test_allocation_failure_is_an_error_not_a_leak(monkeypatch):
    """RUHVRO_HIP_FAIL_ALLOC=N fails the N-th fresh pool allocation the way an exhausted device does.  Whatever stage of the launch
    sequence it hits: RuntimeError (RH_ERR_RUNTIME), the leases taken so far go back to their pools, and the next call is fine."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free0, total = ctypes.c_size_t(), ctypes.c_size_t()
    n = 300_000
    data, offsets = fastgen.generate("cfg3", n)
    d_data, d_off = _to_device(data, offsets)
    exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS["cfg3"]), data, offsets, 4, threaded=True)
    stream = torch.cuda.current_stream().cuda_stream

    def call(schema, host):
        if host:
            return cabi.decode_packed(data, offsets, schema, 4)
        r = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), n, schema, 4, device=0, stream=stream)
        try:
            return r.to_host()
        finally:
            r.free()

    failures = 0
    for host in (False, True):
        for nth in range(1, 9):
            # a payload size of its own per attempt: the pools hold no block of a fitting class, so fresh allocations happen
            schema = SCHEMAS["cfg3"] + "\t" * (7 + nth + (20 if host else 0))
            monkeypatch.setenv("RUHVRO_HIP_FAIL_ALLOC", str(nth))
            try:
                got = call(schema, host)
            except RuntimeError as e:
                assert "injected failure" in str(e)
                failures += 1
                got = None
            monkeypatch.delenv("RUHVRO_HIP_FAIL_ALLOC")
            if got is None:
                got = call(schema, host)                 # the very next call on the same schema works
            for g, e in zip(got, exp):
                assert_batches_identical(g, e)
    assert failures >= 2, failures                      # the hook really hit allocations of live calls
    hip.hipMemGetInfo(ctypes.byref(free0), ctypes.byref(total))
    assert free0.value > 0


@pytest.mark.parametrize("n", [20_000, 300_000])
def test_results_kept_alive_across_calls_stay_intact(n):
    """Host calls write their results straight into pooled pinned host memory and LEND the block to the batches (no D2H copy); a
    dropped result returns its block to the pool, where the next call's emit kernel writes into it.  A caller that keeps batches
    alive across many calls (the pool runs dry: pageable fallback, background refill) while dropping others (blocks recycled under
    live neighbours) must find exactly the oracle's bytes in EVERY batch it still holds -- a recycled block under a live result
    would show here.  20k records: one chunk group; 300k: pipelined chunk groups."""
    schema = SCHEMAS["full"]
    kept = []
    for i in range(12):
        data, offsets = fastgen.generate("full", n, start=i * n)              # different records every call
        recs = fastgen.split(data, offsets)
        got = P.deserialize_array_threaded(recs, schema, 8)
        kept.append((i, got, c_walker.decode_packed(c_walker.CompiledSchema(schema), data, offsets, 8, threaded=True)))
        if i % 3 == 2:
            del kept[len(kept) // 2]                                           # let go of one in the middle
        if i % 4 == 3:
            for _, g, e in kept:                                               # and look at everything still held, mid-way
                for a, b in zip(g, e):
                    assert_batches_identical(a, b)
    assert len(kept) == 8
    for _, g, e in kept:
        for a, b in zip(g, e):
            a.validate(full=True)
            assert_batches_identical(a, b)
