"""RH_ASYNC (rh_opts.flags, ABI 4): rh_decode_device returns with the call on its stream and rh_device_result_wait (or
the first accessor) settles it.  Same buffers, same errors, same retry / fallback branches as the synchronous call --
each branch proven taken through rh_engine_counters.  Needs an MI355X."""
import numpy as np
import pytest
import torch

import cases
from arrow_compare import assert_batches_identical
from avrogen import synth
from avrogen.schemas import SCHEMAS
from oracle import c_walker

from pyruhvro_amd import cabi

pytestmark = pytest.mark.gpu


def _resident(recs):
    data, offsets = c_walker.pack(recs)
    d_data = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda:0")
    d_data[: len(data)].copy_(torch.from_numpy(data.copy()))
    d_off = torch.from_numpy(offsets.view(np.int64).copy()).to("cuda:0")
    torch.cuda.synchronize()
    return d_data, d_off, int(offsets[-1])


def _call(res, n, schema, k, asynchronous=True, kernel=0, want_stats=False):
    d_data, d_off, dl = res
    return cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), dl, n, schema, k, device=0,
                              stream=torch.cuda.current_stream().cuda_stream, want_stats=want_stats, kernel=kernel,
                              asynchronous=asynchronous)


def _delta(before):
    now = cabi.engine_counters()
    return {k: now[k] - before[k] for k in now}


@pytest.mark.parametrize("kernel", [cabi.KERNEL_GENERIC, cabi.KERNEL_SPECIALIZED])
@pytest.mark.parametrize("name,n,k", [("full", 30011, 8), ("cfg3", 20000, 3), ("flat4", 50000, 4), ("array_and_map", 9000, 2)])
def test_async_calls_in_flight_are_identical_to_the_oracle(kernel, name, n, k):
    recs = synth.records(name, n, seed=11)
    exp = c_walker.decode_threaded(recs, SCHEMAS[name], k)
    res = _resident(recs)
    _call(res, n, SCHEMAS[name], k, asynchronous=False, kernel=kernel).free()      # the schema's size history
    c0 = cabi.engine_counters()
    inflight = [_call(res, n, SCHEMAS[name], k, kernel=kernel, want_stats=(i == 1)) for i in range(4)]   # four calls on the stream, none settled
    assert _delta(c0)["fused_calls"] == 4
    inflight[3].free()                                                  # freed unsettled: drains the stream first
    inflight[1].wait()
    assert inflight[1].stats["records"] == n and inflight[1].stats["emit_kernel_ms"] > 0
    for r in inflight[:3]:                                              # [0] and [2] settle inside their first accessor
        assert r.chunks == len(exp)
        for g, e in zip(r.to_host(), exp):
            g.validate(full=True)
            assert_batches_identical(g, e)
        r.wait()                                                        # settled: a no-op
        r.free()


@pytest.mark.parametrize("kernel", [cabi.KERNEL_GENERIC, cabi.KERNEL_SPECIALIZED])
def test_async_error_is_reported_by_wait_and_by_every_accessor(kernel):
    for _, schema, good, bad, msg in cases.error_cases()[:8]:
        recs = good * 300 + [bad] + good * 200 + [bad]
        res = _resident(recs)
        _call(_resident(good * 400), len(good) * 400, schema, 3, asynchronous=False, kernel=kernel).free()
        r = _call(res, len(recs), schema, 3, kernel=kernel)            # returns: the error is not known yet
        with pytest.raises(ValueError) as ei:
            r.wait()
        assert str(ei.value) == msg                                      # the LOWEST malformed record's message
        with pytest.raises(ValueError) as ei:
            r.to_host()
        assert str(ei.value) == msg
        assert r.output_bytes == 0
        r.free()
        r = _call(res, len(recs), schema, 3, kernel=kernel)
        with pytest.raises(ValueError) as ei:
            r.to_host()                                                  # first accessor settles
        assert str(ei.value) == msg
        r.free()


def test_async_arena_retry_and_wide_index_fallback(monkeypatch):
    recs = synth.records("full", 40000, seed=5)
    exp = c_walker.decode_threaded(recs, SCHEMAS["full"], 3)
    res = _resident(recs)
    _call(res, len(recs), SCHEMAS["full"], 3, asynchronous=False, kernel=cabi.KERNEL_SPECIALIZED).free()
    monkeypatch.setenv("RUHVRO_HIP_ARENA_PERMILLE", "1")
    c0 = cabi.engine_counters()
    r = _call(res, len(recs), SCHEMAS["full"], 3, kernel=cabi.KERNEL_SPECIALIZED)
    for g, e in zip(r.to_host(), exp):
        assert_batches_identical(g, e)
    r.free()
    d = _delta(c0)
    assert d["capacity_retries"] == 1 and d["fused_calls"] == 1, d
    monkeypatch.delenv("RUHVRO_HIP_ARENA_PERMILLE")
    monkeypatch.setenv("RUHVRO_HIP_NARROW_ROWS", "15000")               # 13333 rows per chunk < 15000 <= ~20000 child rows
    c0 = cabi.engine_counters()
    r = _call(res, len(recs), SCHEMAS["full"], 3, kernel=cabi.KERNEL_SPECIALIZED, want_stats=True)
    r.wait()
    assert _delta(c0)["wide_fallbacks"] == 1
    assert r.stats["specialized"] == 0
    for g, e in zip(r.to_host(), exp):
        assert_batches_identical(g, e)
    r.free()


def test_first_call_of_a_schema_completes_synchronously():
    import json
    schema = json.dumps({"type": "record", "name": "FreshAsync", "fields": [{"name": "a", "type": "long"}, {"name": "s", "type": "string"}]})
    from avrogen.encoder import to_datum
    from oracle.avro_schema import parse_schema
    tree = parse_schema(schema)
    recs = [to_datum(tree, {"a": i, "s": "v%d" % i}) for i in range(3000)]
    res = _resident(recs)
    c0 = cabi.engine_counters()
    r = _call(res, len(recs), schema, 2)
    d = _delta(c0)
    assert d["two_sync_calls"] == 1 and d["fused_calls"] == 0           # no history: nothing to reserve the arena from
    for g, e in zip(r.to_host(), c_walker.decode_threaded(recs, schema, 2)):
        assert_batches_identical(g, e)
    r.free()


def test_async_calls_from_several_threads_on_their_own_streams():
    """Four host threads, each with its own HIP stream, keep three RH_ASYNC calls in flight and settle them in order --
    the pools (control blocks, workspaces, arenas, pinned blocks), the token spin and the publish kernel under
    concurrency; every result buffer-identical to the oracle."""
    import threading
    name, n, k = "full", 20011, 4
    recs = synth.records(name, n, seed=23)
    exp = c_walker.decode_threaded(recs, SCHEMAS[name], k)
    res = _resident(recs)
    d_data, d_off, dl = res
    _call(res, n, SCHEMAS[name], k, asynchronous=False).free()
    errors = []

    def work(tid):
        try:
            st = torch.cuda.Stream(device="cuda:0")
            ring = []
            for i in range(12):
                ring.append(cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), dl, n, SCHEMAS[name], k, device=0,
                                               stream=st.cuda_stream, want_stats=False, asynchronous=True))
                if len(ring) > 3:
                    r = ring.pop(0)
                    r.wait()
                    if i % 4 == tid % 4:
                        for g, e in zip(r.to_host(), exp):
                            assert_batches_identical(g, e)
                    r.free()
            for r in ring:
                for g, e in zip(r.to_host(), exp):          # settles inside the accessor
                    assert_batches_identical(g, e)
                r.free()
        except Exception as e:                               # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
