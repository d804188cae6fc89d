"""The single-pass form of the device-resident decode (spec_body.h spec_fused; engine_device_call.cpp rh_decode_call::try_single): ONE kernel
sizes a tile, scans across the tiles of its chunk (decoupled look-back) and emits out of the same LDS window.  Opt-in
(RH_SINGLE_PASS in rh_opts.flags).  A schema's first call has no size history and takes the two-pass form; every later one the single pass -- proven through
rh_engine_counters -- and produces the same buffers (oracle: ruhvro/src/fast_decode.rs:570-922 restated), the same error
texts, the same results when a column outgrows its capacity and the call fails over to the two-pass form.  Needs an MI355X.
"""
import numpy as np
import pytest
import torch

import cases
import random_cases
from arrow_compare import assert_batches_identical
from avrogen import fastgen, synth
from avrogen.schemas import SCHEMAS
from oracle import c_walker

from pyruhvro_amd import cabi

pytestmark = pytest.mark.gpu


def _resident(data, offsets):
    d_data = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda:0")
    d_data[: len(data)].copy_(torch.from_numpy(np.ascontiguousarray(data).copy()))
    d_off = torch.from_numpy(offsets.view(np.int64).copy()).to("cuda:0")
    torch.cuda.synchronize()
    return d_data, d_off, int(offsets[-1])


def _call(res, n, schema, k, **kw):
    d_data, d_off, dl = res
    kw.setdefault("single_pass", not kw.get("two_pass", False))        # RH_SINGLE_PASS: the form under test is opt-in
    return cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), dl, n, schema, k, device=0,
                              stream=torch.cuda.current_stream().cuda_stream, kernel=cabi.KERNEL_SPECIALIZED, **kw)


def _delta(before):
    now = cabi.engine_counters()
    return {key: now[key] - before[key] for key in now}


def _check(r, exp):
    got = r.to_host()
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        g.validate(full=True)
        assert_batches_identical(g, e)
    r.free()


@pytest.mark.parametrize("k", [1, 3, 8, 103])
@pytest.mark.parametrize("name,n", [("full", 60_011), ("cfg3", 40_000), ("array_and_map", 9_000), ("nullable_primitives", 5_000)])
def test_single_pass_is_identical_to_the_oracle(name, n, k):
    schema = SCHEMAS[name] + " " * k                 # (its own cached schema object per case: every case starts without history)
    recs = synth.records(name, n, seed=7 + k)
    data, offsets = c_walker.pack(recs)
    exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS[name]), data, offsets, k, threaded=True)
    res = _resident(data, offsets)
    c0 = cabi.engine_counters()
    _check(_call(res, n, schema, k), exp)                               # no history yet: the two-pass form
    assert _delta(c0)["single_pass_calls"] == 0
    _check(_call(res, n, schema, k), exp)                               # the single pass
    _check(_call(res, n, schema, k, asynchronous=True), exp)            # ... settled by its first accessor
    d = _delta(c0)
    # (a call whose chunks are so small that the k x K capacities do not fit the workspace's blocksum area -- fewer than two tiles
    #  per chunk on average -- stays on the two-pass form: since round 5 that is decided before anything is counted or leased)
    sz, last = n // k, n - (k - 1) * (n // k)
    nblocks = (k - 1) * ((sz + 255) // 256) + (last + 255) // 256
    want = 2 if 2 * k <= nblocks else 0
    assert d["single_pass_calls"] == want and d["single_pass_failovers"] == 0
    _check(_call(res, n, schema, k, two_pass=True), exp)                # RH_TWO_PASS: the caller's choice
    _check(_call(res, n, schema, k, single_pass=False), exp)            # neither flag: the default is the two-pass form
    assert _delta(c0)["single_pass_calls"] == want


def test_single_pass_at_one_million_records_full_buffer_identity():
    n, k = 1_000_000, 8
    data, offsets = fastgen.generate("full", n)
    exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS["full"]), data, offsets, k, threaded=True)
    res = _resident(data, offsets)
    schema = SCHEMAS["full"] + "\r"                 # (a schema object of its own: no history or back-off from other tests)
    _call(res, n, schema, k).free()
    c0 = cabi.engine_counters()
    inflight = [_call(res, n, schema, k, asynchronous=True) for _ in range(3)]
    for r in inflight:
        _check(r, exp)
    d = _delta(c0)
    assert d["single_pass_calls"] == 3 and d["single_pass_failovers"] == 0


def test_single_pass_at_ten_million_records_full_buffer_identity():
    """The size bench.py times (BASELINE config 4): every buffer of every chunk of a single-pass call against the oracle."""
    n, k = 10_000_000, 8
    data, offsets = fastgen.generate("full", n)
    exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS["full"]), data, offsets, k, threaded=True)
    res = _resident(data, offsets)
    schema = SCHEMAS["full"] + "\r\n"             # (a schema object of its own)
    _call(res, n, schema, k).free()                  # history
    c0 = cabi.engine_counters()
    _check(_call(res, n, schema, k, asynchronous=True), exp)
    d = _delta(c0)
    assert d["single_pass_calls"] == 1 and d["single_pass_failovers"] == 0


def test_a_column_that_outgrows_its_capacity_fails_over_to_the_two_pass_form(monkeypatch):
    """History from short strings, then the same schema with strings several times as long: the single pass raises LF_CAPACITY
    (it writes nothing beyond a capacity), the call is repeated on the two-pass form, and the schema sits the next calls out
    (backing off: 8 calls after the first fail-over)."""
    schema = SCHEMAS["cfg3"] + "\n"
    short = synth.records("cfg3", 30_000, seed=1)
    import avrogen.encoder as enc
    from oracle.avro_schema import parse_schema
    sch = parse_schema(SCHEMAS["cfg3"])
    long_recs = [enc.to_datum(sch, {"id": i, "name": "n" * (40 + i % 50), "age": None, "s": "x" * (30 + i % 7), "class": "ABC"[i % 3]})
                 for i in range(30_000)]
    for recs in (short, long_recs):
        data, offsets = c_walker.pack(recs)
        exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS["cfg3"]), data, offsets, 4, threaded=True)
        c0 = cabi.engine_counters()
        _check(_call(_resident(data, offsets), len(recs), schema, 4), exp)
        d = _delta(c0)
        if recs is long_recs:
            assert d["single_pass_calls"] == 1 and d["single_pass_failovers"] == 1
    # cooling down: this schema object does not try the single pass on its next call
    c0 = cabi.engine_counters()
    data, offsets = c_walker.pack(short)
    _check(_call(_resident(data, offsets), len(short), schema, 4),
           c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS["cfg3"]), data, offsets, 4, threaded=True))
    assert _delta(c0)["single_pass_calls"] == 0


def test_failover_hook_every_call(monkeypatch):
    """RUHVRO_HIP_SINGLE_SLACK_PERMILLE below 1000 shrinks every capacity below what the last call needed: every single-pass
    call fails over (not latched under the hook), results identical."""
    n, k = 50_000, 8
    data, offsets = fastgen.generate("full", n)
    exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS["full"]), data, offsets, k, threaded=True)
    res = _resident(data, offsets)
    schema = SCHEMAS["full"] + "\t"
    _call(res, n, schema, k).free()
    monkeypatch.setenv("RUHVRO_HIP_SINGLE_SLACK_PERMILLE", "700")
    c0 = cabi.engine_counters()
    _check(_call(res, n, schema, k), exp)
    _check(_call(res, n, schema, k, asynchronous=True), exp)
    d = _delta(c0)
    assert d["single_pass_calls"] == 2 and d["single_pass_failovers"] == 2
    monkeypatch.delenv("RUHVRO_HIP_SINGLE_SLACK_PERMILLE")
    c0 = cabi.engine_counters()
    _check(_call(res, n, schema, k), exp)
    assert _delta(c0) ["single_pass_failovers"] == 0 and _delta(c0)["single_pass_calls"] == 1


def test_error_texts_in_the_single_pass():
    """A malformed record met by the single pass reports the reference's message for the LOWEST failing record
    (fast_decode.rs:575,591,646,849,866,874,884,898,906,910), like the two-pass form."""
    for i, (_, schema, good, bad, msg) in enumerate(cases.error_cases()):
        schema = schema + " " * (i + 1)
        warm = good * 400
        data, offsets = c_walker.pack(warm)
        _call(_resident(data, offsets), len(warm), schema, 3).free()                      # history
        recs = good * 300 + [bad] + good * 200 + [bad]
        data, offsets = c_walker.pack(recs)
        c0 = cabi.engine_counters()
        with pytest.raises(ValueError) as ei:
            _call(_resident(data, offsets), len(recs), schema, 3)
        assert str(ei.value) == msg, (schema, str(ei.value), msg)
        assert _delta(c0)["single_pass_calls"] == 1
        r = _call(_resident(data, offsets), len(recs), schema, 3, asynchronous=True)
        with pytest.raises(ValueError) as ei:
            r.wait()
        assert str(ei.value) == msg
        r.free()


@pytest.mark.parametrize("seed", range(12))
def test_random_schemas_single_pass(seed):
    schema, recs = random_cases.random_case(1000 + seed, 900)
    exp = c_walker.decode_threaded(recs, schema, 5)
    data, offsets = c_walker.pack(recs)
    res = _resident(data, offsets)
    schema_sp = schema + " "
    _check(_call(res, len(recs), schema_sp, 5), exp)
    c0 = cabi.engine_counters()
    _check(_call(res, len(recs), schema_sp, 5), exp)
    _check(_call(res, len(recs), schema_sp, 5), exp)
    assert _delta(c0)["single_pass_calls"] in (0, 2)       # (0: a schema without variable-length output has no size pass to fuse)


def test_nesting_and_wire_cases_single_pass():
    for name, schema, recs, *_ in list(cases.nesting_cases()) + list(cases.wire_cases()) + list(cases.dense_list_cases()):
        if "n4" in name:               # (the C oracle has no N4 leaves; tests/test_n4_types.py covers them through the host path)
            continue
        recs = (list(recs) * 40)[:3000]
        exp = c_walker.decode_threaded(recs, schema, 3)
        data, offsets = c_walker.pack(recs)
        res = _resident(data, offsets)
        s2 = schema + "  "
        for _ in range(3):
            _check(_call(res, len(recs), s2, 3), exp)
