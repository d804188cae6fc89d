"""The engine's rarely taken branches (engine_device_call.cpp decode_device_impl1), each forced with its test hook and PROVEN taken
through rh_engine_counters: the arena-capacity retry (LF_CAPACITY), the wide-index fallback to the generic kernels
(NeedWideIndex / LF_NEED_WIDE), the 32-bit Arrow offset overflow (LF_OFFSET32) on a real > 2 GiB column, and the
two-submission path (RUHVRO_HIP_TWO_SYNC) over the parity matrix.  Needs an MI355X."""
import json

import numpy as np
import pytest

import cases
from arrow_compare import assert_batches_identical
from avrogen import synth
from avrogen.encoder import to_datum
from avrogen.schemas import SCHEMAS
from oracle import c_walker
from oracle.avro_schema import parse_schema

import pyruhvro_amd as P
from pyruhvro_amd import cabi

pytestmark = pytest.mark.gpu


def _check(recs, schema, k):
    got = P.deserialize_array_threaded(recs, schema, k)
    exp = c_walker.decode_threaded(recs, schema, k)
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        g.validate(full=True)
        assert_batches_identical(g, e)


def _delta(before):
    now = cabi.engine_counters()
    return {k: now[k] - before[k] for k in now}


@pytest.mark.parametrize("mode", ["generic", "specialized"])
def test_arena_capacity_retry(mode, monkeypatch):
    """The arena is reserved from the schema's size history; when that turns out too small the layout kernel raises
    LF_CAPACITY, init / emit return at once and the host re-runs the tail with an exact arena."""
    old = P.set_kernel_mode(mode)
    try:
        recs = synth.records("full", 40000, seed=5)             # ~7 MB of Arrow buffers: beyond the reservation's fixed 1 MiB slack
        _check(recs, SCHEMAS["full"], 3)                        # gives the schema a history (first call: two submissions)
        monkeypatch.setenv("RUHVRO_HIP_ARENA_PERMILLE", "1")   # "history": 0.001 output bytes per input byte
        c0 = cabi.engine_counters()
        _check(recs, SCHEMAS["full"], 3)
        _check(synth.records("array_and_map", 40000, seed=1), SCHEMAS["array_and_map"], 4)
        d = _delta(c0)
        assert d["capacity_retries"] == 2 and d["fused_calls"] == 2, d
        monkeypatch.delenv("RUHVRO_HIP_ARENA_PERMILLE")
        c0 = cabi.engine_counters()
        _check(recs, SCHEMAS["full"], 3)                        # and the real history still fits
        d = _delta(c0)
        assert d["capacity_retries"] == 0 and d["fused_calls"] == 1, d
    finally:
        P.set_kernel_mode(old)


def test_wide_index_fallback_to_the_generic_kernels(monkeypatch):
    """The specialised kernels index every chunk buffer with 32-bit byte offsets and child row domains below
    narrow_rows; a call whose child domain reaches the bound is re-run on the generic (64-bit) kernels.  The hook lowers
    the bound so that 2000-row chunks with ~3 list items per row cross it."""
    old = P.set_kernel_mode("specialized")
    try:
        recs = synth.records("full", 6000, seed=2)
        _check(recs, SCHEMAS["full"], 3)
        monkeypatch.setenv("RUHVRO_HIP_NARROW_ROWS", "2500")     # rows per chunk 2000 < 2500 <= child rows (~3000 per chunk)
        c0 = cabi.engine_counters()
        _, st = P.deserialize_array_threaded_with_stats(recs, SCHEMAS["full"], 3)
        d = _delta(c0)
        assert d["wide_fallbacks"] == 1, d
        assert st["specialized"] == 0                            # the batches came from the generic kernels
        _check(recs, SCHEMAS["full"], 3)
        monkeypatch.setenv("RUHVRO_HIP_NARROW_ROWS", "1500")     # below the chunk's own rows: generic from the start, no re-run
        c0 = cabi.engine_counters()
        _check(recs, SCHEMAS["full"], 3)
        assert _delta(c0)["wide_fallbacks"] == 0
    finally:
        P.set_kernel_mode(old)


def test_offset32_overflow_on_a_real_column():
    """A chunk whose string column holds more than 2^31 - 1 bytes cannot have 32-bit Arrow offsets (the reference's
    StringBuilder would overflow its i32 offsets): the layout pass says so and the call fails -- on a REAL 2.2 GB column,
    2200 records of one 1 MiB string each, in one chunk."""
    schema = json.dumps({"type": "record", "name": "Big", "fields": [{"name": "i", "type": "int"}, {"name": "s", "type": "string"}]})
    tree = parse_schema(schema)
    rec = np.frombuffer(to_datum(tree, {"i": 7, "s": "x" * (1 << 20)}), dtype=np.uint8)
    n = 2200
    data = np.tile(rec, n)
    offsets = (np.arange(n + 1, dtype=np.uint64) * np.uint64(len(rec)))
    c0 = cabi.engine_counters()
    with pytest.raises(ValueError) as ei:
        cabi.decode_packed(data, offsets, schema, 1)
    assert "offset overflow" in str(ei.value) and "32-bit Arrow offsets" in str(ei.value)
    assert _delta(c0)["offset32_errors"] >= 1
    # the same records in two chunks fit (1100 MiB per chunk) and decode
    got = cabi.decode_packed(data, offsets, schema, 2)
    assert [b.num_rows for b in got] == [1100, 1100]
    col = got[1].column("s")
    o = np.frombuffer(col.buffers()[1], dtype=np.int32, count=1101)
    assert o[0] == 0 and o[-1] == 1100 * (1 << 20) and got[0].column("i").to_pylist()[:3] == [7, 7, 7]
    assert bytes(col.buffers()[2][:4]) == b"xxxx" and bytes(col.buffers()[2][o[-1] - 4:o[-1]]) == b"xxxx"
    del got, col
    # the schema has a size history now, so the one-chunk call takes the single-submission path: this time it is the
    # layout KERNEL that refuses (LF_OFFSET32), emit returns at once, and the host reports the same error
    c0 = cabi.engine_counters()
    with pytest.raises(ValueError) as ei:
        cabi.decode_packed(data, offsets, schema, 1)
    assert "offset overflow" in str(ei.value)
    d = _delta(c0)
    assert d["offset32_errors"] == 1 and d["two_sync_calls"] == 0, d


@pytest.mark.parametrize("mode", ["generic", "specialized"])
def test_two_submission_path_parity(mode, monkeypatch):
    """RUHVRO_HIP_TWO_SYNC=1: totals to the host between scan and emit, arena laid out there (round 1's path, and what
    the first call of every schema still does) -- same buffers and same errors as the single-submission path."""
    old = P.set_kernel_mode(mode)
    try:
        monkeypatch.setenv("RUHVRO_HIP_TWO_SYNC", "1")
        c0 = cabi.engine_counters()
        calls = 0
        for name, n, k in (("full", 5003, 7), ("cfg3", 3000, 2), ("array_and_map", 2500, 3), ("flat4", 4000, 5), ("full", 65, 1)):
            _check(synth.records(name, n, seed=3), SCHEMAS[name], k)
            calls += 1
        for _, schema, recs in cases.wire_cases() + cases.nesting_cases():
            _check((recs * 40)[:1000], schema, 3)
            calls += 1
        d = _delta(c0)
        assert d["two_sync_calls"] == calls and d["fused_calls"] == 0, d
        for _, schema, good, bad, msg in cases.error_cases()[:10]:
            with pytest.raises(ValueError) as ei:
                P.deserialize_array_threaded(good * 200 + [bad] + good * 100, schema, 5)
            assert str(ei.value) == msg
    finally:
        P.set_kernel_mode(old)
