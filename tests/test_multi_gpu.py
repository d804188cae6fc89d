"""The multi-GPU form of the chunk driver (BASELINE.json config 5): one list, the reference's chunks
(ruhvro/src/deserialize.rs:57-68) dealt to g shards in contiguous runs, each shard on its own device / host thread /
stream / arena.  Tier C of SURVEY 4.4: the list decoded over g in {1,2,4,8} shards must be identical to the 1-shard
result and to the oracle.  On a one-GPU box the g shards are logical shards of device 0 (rh_opts.devices = [0]*g):
the partition / reassembly logic is the same code that runs with g distinct devices."""
import os

import numpy as np
import pytest

import cases
from arrow_compare import assert_batches_identical
from avrogen import fastgen, synth
from avrogen.schemas import SCHEMAS
from oracle import c_walker

import pyruhvro_amd as P
from pyruhvro_amd import cabi
from pyruhvro_amd.dist import partition_chunks, strong_shard


# ------------------------------------------------------------------------------------------------ CPU (no GPU needed)
def test_shard_deal_keeps_the_reference_chunk_bounds():
    for n in (0, 1, 5, 103, 1000, 10_000_000):
        for k in (0, 1, 2, 3, 8, 16, 500):
            kk = max(1, min(max(k, 1), max(n, 1)))
            sz = n // kk
            bounds = [(i * sz, n if i == kk - 1 else (i + 1) * sz) for i in range(kk)]        # deserialize.rs:57-68
            for g in (1, 2, 3, 4, 8, 11):
                deal = [cabi.shard_chunks(n, k, g, j) for j in range(g)]
                assert deal[0][0] == 0 and deal[-1][1] == kk
                for (c0, c1, r0, r1), nxt in zip(deal, deal[1:] + [None]):
                    assert c0 <= c1
                    if nxt:
                        assert nxt[0] == c1                                                   # contiguous runs, in order
                    if c1 > c0:
                        assert (r0, r1) == (bounds[c0][0], bounds[c1 - 1][1])                 # whole reference chunks
                    else:
                        assert r0 == r1
                assert [b for part in partition_chunks(n, k, g) for b in part] == bounds
                assert [len(p) for p in partition_chunks(n, k, g)] == [c1 - c0 for c0, c1, _, _ in deal]


def test_strong_shard_geometry():
    s = [strong_shard(10_000_000, 8, 8, r) for r in range(8)]
    assert [x["rows"] for x in s] == [1_250_000] * 8 and [x["chunks"] for x in s] == [1] * 8
    assert [x["row_lo"] for x in s] == [r * 1_250_000 for r in range(8)]
    s = [strong_shard(1003, 8, 2, r) for r in range(2)]
    assert [(x["row_lo"], x["rows"], x["chunks"], x["chunk_rows"]) for x in s] == [(0, 500, 4, 125), (500, 503, 4, 125)]
    s = [strong_shard(5, 2, 4, r) for r in range(4)]
    assert [x["chunks"] for x in s] == [0, 1, 0, 1] and sum(x["rows"] for x in s) == 5


def test_multi_device_call_without_a_gpu_fails_loudly():
    if P.device_count() > 0:
        pytest.skip("a GPU is present")
    data, offsets = c_walker.pack(synth.records("full", 10))
    with pytest.raises(RuntimeError):
        cabi.decode_packed(data, offsets, SCHEMAS["full"], 2, devices=[0, 0])


# ------------------------------------------------------------------------------------------------ GPU
KERNELS = {"generic": cabi.KERNEL_GENERIC, "specialized": cabi.KERNEL_SPECIALIZED}


@pytest.fixture(params=sorted(KERNELS))
def kernel(request):
    old = P.set_kernel_mode(request.param)
    yield KERNELS[request.param]
    P.set_kernel_mode(old)


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,k", [("full", 1003, 8), ("full", 20000, 8), ("cfg3", 5003, 7), ("flat4", 4096, 16),
                                      ("array_and_map", 777, 3), ("full", 5, 8), ("full", 300, 1)])
def test_tier_c_g_shards_equal_one_shard_and_the_oracle(name, n, k, kernel):
    recs = synth.records(name, n, seed=5)
    data, offsets = c_walker.pack(recs)
    exp = c_walker.decode_threaded(recs, SCHEMAS[name], k)
    one = cabi.decode_packed(data, offsets, SCHEMAS[name], k, kernel=kernel)
    assert len(one) == len(exp)
    for a, e in zip(one, exp):
        assert_batches_identical(a, e)
    for g in (1, 2, 3, 4, 8, 16):
        got, st = cabi.decode_packed(data, offsets, SCHEMAS[name], k, kernel=kernel, devices=[0] * g, want_stats=True)
        assert len(got) == len(exp), g
        for a, e in zip(got, exp):
            a.validate(full=True)
            assert_batches_identical(a, e)
        assert st["records"] == n and st["chunks"] == len(exp)
        per = st["device_stats"]
        assert len(per) == g and sum(x["records"] for x in per) == n
        assert [x["chunks"] for x in per] == [c1 - c0 for c0, c1, _, _ in (cabi.shard_chunks(n, k, g, j) for j in range(g))]


@pytest.mark.gpu
def test_tier_c_1m_records_over_8_shards(kernel):
    """config-5 shape at 1M: 8 chunks over 8 (logical) shards, full buffer identity with the 1-shard call and the oracle."""
    data, offsets = fastgen.generate("full", 1_000_000)
    exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS["full"]), data, offsets, 8, threaded=True)
    for g in (2, 8):
        got = cabi.decode_packed(data, offsets, SCHEMAS["full"], 8, kernel=kernel, devices=[0] * g)
        assert [b.num_rows for b in got] == [125_000] * 8
        for a, e in zip(got, exp):
            assert_batches_identical(a, e)


@pytest.mark.gpu
@pytest.mark.parametrize("case", cases.error_cases()[:6], ids=lambda c: c[0])
def test_multi_shard_errors_report_the_lowest_failing_record(case):
    _, schema, good, bad, msg = case
    recs = good * 200 + [bad] + good * 100 + [b"\x80" * 11]         # a malformed record in an early AND in the last shard
    for g in (2, 4, 8):
        data, offsets = c_walker.pack(recs)
        with pytest.raises(ValueError) as ei:
            cabi.decode_packed(data, offsets, schema, 8, devices=[0] * g)
        assert str(ei.value) == msg
    ok = good * 50                                                    # and nothing is left waiting at a gate
    data, offsets = c_walker.pack(ok)
    for a, e in zip(cabi.decode_packed(data, offsets, schema, 4, devices=[0, 0]), c_walker.decode_threaded(ok, schema, 4)):
        assert_batches_identical(a, e)


@pytest.mark.gpu
def test_python_surface_device_list(monkeypatch):
    recs = synth.records("full", 2000, seed=9)
    exp = c_walker.decode_threaded(recs, SCHEMAS["full"], 8)
    old = P.set_devices([0, 0, 0])
    try:
        for a, e in zip(P.deserialize_array_threaded(recs, SCHEMAS["full"], 8), exp):
            assert_batches_identical(a, e)
        assert_batches_identical(P.deserialize_array(recs, SCHEMAS["full"]), c_walker.decode(recs, SCHEMAS["full"]))
    finally:
        P.set_devices(old)
    monkeypatch.setenv("PYRUHVRO_DEVICES", "0,0")
    for a, e in zip(P.deserialize_array_threaded(recs, SCHEMAS["full"], 8), exp):
        assert_batches_identical(a, e)
    monkeypatch.setenv("PYRUHVRO_DEVICES", "0,x")
    with pytest.raises(ValueError):
        P.deserialize_array_threaded(recs, SCHEMAS["full"], 8)
    monkeypatch.delenv("PYRUHVRO_DEVICES")
    data, offsets = c_walker.pack(recs)
    with pytest.raises(ValueError, match="out of range"):
        cabi.decode_packed(data, offsets, SCHEMAS["full"], 8, devices=[0, 99])


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(1003, 8), (40000, 8), (9, 4)])
def test_process_per_gpu_ranges_with_explicit_chunk_geometry(n, k, kernel):
    """What bench.py's ranks do (one process per GPU): every rank decodes only ITS rows through rh_decode_device with
    rh_opts.chunk_rows; the ranks' batches in rank order are the one-call result."""
    import hipmem
    recs = synth.records("full", n, seed=11)
    exp = c_walker.decode_threaded(recs, SCHEMAS["full"], k)
    for world in (1, 2, 4, 8):
        out = []
        for rank in range(world):
            sh = strong_shard(n, k, world, rank)
            part = recs[sh["row_lo"]: sh["row_lo"] + sh["rows"]]
            if sh["chunks"] == 0:
                continue
            data, offsets = c_walker.pack(part)
            d_data, d_off = hipmem.upload_packed(data, offsets)
            r = cabi.decode_device(d_data.ptr, d_off.ptr, int(offsets[-1]), len(part), SCHEMAS["full"],
                                   sh["chunks"], device=0, kernel=kernel, chunk_rows=sh["chunk_rows"])
            out += r.to_host()
            r.free()
        assert len(out) == len(exp)
        for a, e in zip(out, exp):
            assert_batches_identical(a, e)
    with pytest.raises(ValueError, match="chunk_rows"):
        cabi.decode_device(0, 0, 0, 10, SCHEMAS["full"], 3, device=0, chunk_rows=7)      # 2 chunks of 7 > 10 rows
