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
    schema = SCHEMAS["full"] + "   "                                   # a handle of this test's own (handles remember code objects)
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
    schema = SCHEMAS["cfg3"] + "    "
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


@pytest.mark.parametrize("kernel", ["generic", "specialized"])
@pytest.mark.parametrize("case", cases.wide_counter_cases(), ids=lambda c: c[0])
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
