"""GPU parity: the HIP path (through the C ABI) vs the oracle, buffer for buffer.  Needs an MI355X."""
import json
import os

import numpy as np
import pyarrow as pa
import pytest

import cases
from arrow_compare import assert_batches_identical
from avrogen import fastgen, synth
from avrogen.schemas import SCHEMAS
from oracle import c_walker

import pyruhvro_amd as P
from pyruhvro_amd import cabi

pytestmark = pytest.mark.gpu

KERNELS = {"generic": cabi.KERNEL_GENERIC, "specialized": cabi.KERNEL_SPECIALIZED}


@pytest.fixture(params=sorted(KERNELS), autouse=True)
def kernel(request):
    """Every parity test runs twice: the generic schema-program interpreter kernels and the
    schema-specialised kernels (same field handlers, walk.h) must both match the oracle."""
    old = P.set_kernel_mode(request.param)
    yield KERNELS[request.param]
    P.set_kernel_mode(old)

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")


def _check(recs, schema, k):
    got = P.deserialize_array_threaded(recs, schema, k)
    exp = c_walker.decode_threaded(recs, schema, k)
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        g.validate(full=True)
        assert_batches_identical(g, e)
    return got


def test_native_library_is_the_path_that_runs(kernel):
    assert P.device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libruhvro_hip.so" in maps and "_pyruhvro.so" in maps
    recs = synth.records("full", 500)
    _, st = P.deserialize_array_threaded_with_stats(recs, SCHEMAS["full"], 2)
    assert st["specialized"] == (1 if kernel == cabi.KERNEL_SPECIALIZED else 0)
    assert st["records"] == 500 and st["emit_kernel_ms"] > 0


def test_golden_vectors():
    g = json.load(open(GOLDEN))
    for v in g["vectors"]:
        schema = json.dumps(g["schemas"][v["schema"]])
        rec = bytes.fromhex(v["hex"])
        rb = P.deserialize_array([rec], schema)
        assert_batches_identical(rb, c_walker.decode([rec], schema))
        row = rb.to_pylist()[0]
        exp = dict(v["expected"])
        exp.pop("status_type_id", None)
        if "registrationDate_ms" in exp:
            assert rb.column("registrationDate").cast("int64")[0].as_py() == exp.pop("registrationDate_ms")
            row.pop("registrationDate")
        norm = json.loads(json.dumps(row))
        assert norm == exp, v["name"]
    by = {v["name"]: v for v in g["vectors"]}
    v = by["deserialize_rs_244"]
    rb = P.deserialize_array([bytes.fromhex(v["hex"])] * 4, json.dumps(g["schemas"][v["schema"]]))
    assert (rb.num_columns, rb.num_rows) == (8, 4)                       # deserialize.rs:248-249


@pytest.mark.parametrize("case", cases.differential_cases(), ids=lambda c: c[0])
def test_differential_restatements(case):
    _, schema, recs, expected = case
    rb = P.deserialize_array(recs, schema)
    assert_batches_identical(rb, c_walker.decode(recs, schema))
    assert json.loads(json.dumps(rb.to_pylist())) == json.loads(json.dumps(expected))


def test_logical_types():
    schema, recs, vals = cases.logical_case()
    rb = P.deserialize_array(recs, schema)
    assert_batches_identical(rb, c_walker.decode(recs, schema))
    assert rb.column("tu").cast("int64").to_pylist() == [v["tu"] for v in vals]


@pytest.mark.parametrize("case", cases.wire_cases() + cases.nesting_cases(), ids=lambda c: c[0])
def test_wire_forms_and_nesting(case):
    _, schema, recs = case
    for k in (1, 3):
        _check(recs, schema, k)
    # many copies: rows of every kind spread over several workgroups and wave positions
    _check((recs * 40)[:1000], schema, 7)


@pytest.mark.parametrize("case", cases.enum_form_cases(), ids=lambda c: c[0])
def test_enum_symbol_forms(case):
    """Symbols as immediates in the specialised size pass (<= 16 symbols of <= 8 bytes) and the symbol-table path beside
    it (17 symbols, a 9-byte symbol), top level / nullable / list items / union variant: both kernel forms, buffer for
    buffer against the oracle."""
    name, schema, recs = case
    for k in (1, 3, 8):
        _check(recs, schema, k)
    _check((recs * 2)[5:], schema, 4)


@pytest.mark.parametrize("case", cases.dense_list_cases(), ids=lambda c: c[0])
def test_item_dense_lists(case):
    """The specialised emit kernel materialises top-level arrays / maps one lane per ITEM (spec_body.h dense_list):
    every body kind, wavefronts with more items than the position table holds (rounds), one huge list among short ones,
    positive multi-block lists, two-byte block counts, nullable lists, N4 leaves -- buffer for buffer against the oracle,
    on both kernel forms (the generic interpreter keeps the per-record loop: the two must agree with each other too)."""
    name, schema, recs = case
    if "n4" in name:       # types beyond the reference's direct path: the specification oracle (py_walker, extended)
        from oracle import py_walker
        exp = py_walker.decode(recs, schema, extended=True)
        for k in (1, 4):
            got = P.deserialize_array_threaded(recs, schema, k)
            assert sum(b.num_rows for b in got) == exp.num_rows
            off = 0
            for g in got:
                g.validate(full=True)
                assert_batches_identical(g, py_walker.decode(recs[off: off + g.num_rows], schema, extended=True))
                off += g.num_rows
        return
    for k in (1, 3, 8):
        _check(recs, schema, k)
    _check((recs * 3)[7:], schema, 5)          # other tile / wave positions for every row


@pytest.mark.parametrize("case", cases.error_cases(), ids=lambda c: c[0])
def test_error_messages_match_reference(case):
    _, schema, good, bad, msg = case
    for recs, k in ((good + [bad] + good, 1), (good * 200 + [bad] + good * 100 + [b"\x80" * 11], 5), ([bad], 1)):
        with pytest.raises(ValueError) as ei:
            P.deserialize_array_threaded(recs, schema, k)
        assert str(ei.value) == msg
        with pytest.raises(ValueError) as eo:
            c_walker.decode_threaded(recs, schema, k)
        assert str(eo.value) == str(ei.value)


@pytest.mark.parametrize("name", sorted(synth.GENERATORS))
@pytest.mark.parametrize("n,k", [(1, 1), (63, 1), (64, 2), (65, 3), (255, 1), (256, 1), (257, 2), (1000, 8), (5003, 7)])
def test_generated_records(name, n, k):
    _check(synth.records(name, n, seed=3), SCHEMAS[name], k)


@pytest.mark.parametrize("seed", range(40))
def test_random_schemas_and_records(seed, kernel):
    """Seeded random schemas inside the direct-decode subset x random records (tests/random_cases.py)."""
    import random_cases
    assert seed < random_cases.PREBUILT_SEEDS      # build() compiled these schemas' specialised kernels ahead of time
    js, recs = random_cases.random_case(seed, 700)
    _check(recs, js, 1 + seed % 4)


def test_concurrent_calls_from_python_threads(kernel):
    """The reference releases the GIL around the native call and is safe for concurrent callers
    (src/lib.rs:64-68,82-86); so is the engine: shared schema cache, pooled memory, one device."""
    import threading
    jobs = [("full", 3000, 4), ("cfg3", 5000, 3), ("flat4", 7000, 2), ("array_and_map", 2000, 5)]
    inputs = {name: synth.records(name, n, seed=11) for name, n, _ in jobs}
    expect = {name: c_walker.decode_threaded(inputs[name], SCHEMAS[name], k) for name, _, k in jobs}
    errors = []

    def work(name, k):
        try:
            for _ in range(6):
                got = P.deserialize_array_threaded(inputs[name], SCHEMAS[name], k)
                for g, e in zip(got, expect[name]):
                    assert_batches_identical(g, e)
        except Exception as ex:  # noqa: BLE001 - reported below
            errors.append((name, repr(ex)))

    threads = [threading.Thread(target=work, args=(name, k)) for name, _, k in jobs for _ in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("name", ["full", "array_and_map", "cfg3", "nullable_primitives"])
def test_corrupted_records_same_outcome_as_oracle(name, kernel):
    """Random damage (bit flips, truncation, junk insertion) to valid records: the HIP path must end exactly like
    the oracle -- the same ValueError text (lowest failing record wins) or the same buffers.  Exercises the
    fast -> careful re-walk of both kernels on every kind of anomaly."""
    import random
    base = synth.records(name, 1500, seed=9)
    r = random.Random(1234 + len(name))
    outcomes = {"error": 0, "ok": 0}
    for trial in range(60):
        recs = list(base)
        for _ in range(r.choice([1, 1, 2, 5])):
            i = r.randrange(len(recs))
            b = bytearray(recs[i])
            how = r.randrange(4)
            if how == 0 and b:
                b[r.randrange(len(b))] ^= 1 << r.randrange(8)
            elif how == 1 and b:
                del b[r.randrange(len(b)):]
            elif how == 2:
                pos = r.randrange(len(b) + 1)
                b[pos:pos] = bytes(r.randrange(256) for _ in range(r.randint(1, 4)))
            elif b:
                b[r.randrange(len(b))] = r.choice([0x80, 0xFF, 0x7F, 0x01, 0x00])
            recs[i] = bytes(b)
        k = 1 + trial % 5
        try:
            exp = c_walker.decode_threaded(recs, SCHEMAS[name], k)
        except ValueError as e:
            outcomes["error"] += 1
            with pytest.raises(ValueError) as ei:
                P.deserialize_array_threaded(recs, SCHEMAS[name], k)
            assert str(ei.value) == str(e), (trial, name)
            continue
        outcomes["ok"] += 1
        got = P.deserialize_array_threaded(recs, SCHEMAS[name], k)
        for g, e in zip(got, exp):
            assert_batches_identical(g, e)
    assert outcomes["error"] > 5 and outcomes["ok"] > 0, outcomes


def test_chunk_semantics():
    recs = synth.records("full", 103)
    for k, want in ((1, [103]), (8, [12] * 7 + [19]), (0, [103]), (103, [1] * 103), (500, [1] * 103), (2, [51, 52])):
        out = _check(recs, SCHEMAS["full"], k)
        assert [b.num_rows for b in out] == want                        # deserialize.rs:53-68
    out = P.deserialize_array_threaded([], SCHEMAS["full"], 4)          # n = 0 -> one empty batch
    assert [b.num_rows for b in out] == [0]
    assert_batches_identical(out[0], c_walker.decode_threaded([], SCHEMAS["full"], 4)[0])
    one = P.deserialize_array(recs, SCHEMAS["full"])
    assert isinstance(one, pa.RecordBatch) and one.num_rows == 103
    assert P.deserialize_array_threaded_spawn(recs, SCHEMAS["full"], 3)[1].equals(P.deserialize_array_threaded(recs, SCHEMAS["full"], 3)[1])


@pytest.fixture
def pipelined(monkeypatch):
    """Force the pipelined host path (groups of chunks on their own streams, engine_host.cpp decode_host_impl) on small
    inputs; by default it starts at 64 MB per call (the 1M / 10M config tests below run it at its default)."""
    monkeypatch.setenv("RUHVRO_HIP_PIPELINE_MIN_MB", "0")


def test_pipelined_host_path_is_identical_to_the_serial_one(pipelined):
    for name, n, k in (("full", 103, 10), ("full", 5003, 7), ("cfg3", 1000, 8), ("array_and_map", 777, 3), ("full", 2, 2),
                       ("flat4", 40000, 16), ("full", 1200, 500)):
        out = _check(synth.records(name, n, seed=2), SCHEMAS[name], k)
        kk = min(max(k, 1), n)
        assert [b.num_rows for b in out] == [n // kk] * (kk - 1) + [n - (kk - 1) * (n // kk)]      # deserialize.rs:53-68
    recs = synth.records("full", 3000, seed=4)
    out, st = P.deserialize_array_threaded_with_stats(recs, SCHEMAS["full"], 16)
    assert st["chunks"] == 16 and st["records"] == 3000 and st["emit_kernel_ms"] > 0 and st["d2h_ms"] > 0
    data, offsets = c_walker.pack(recs)
    for x, y in zip(out, cabi.decode_packed(data, offsets, SCHEMAS["full"], 16)):
        assert_batches_identical(x, y)


@pytest.mark.parametrize("case", cases.error_cases()[:8], ids=lambda c: c[0])
def test_pipelined_errors_report_the_lowest_failing_group(pipelined, case):
    _, schema, good, bad, msg = case
    recs = good * 200 + [bad] + good * 100 + [b"\x80" * 11]
    for k in (5, 9):
        with pytest.raises(ValueError) as ei:
            P.deserialize_array_threaded(recs, schema, k)
        assert str(ei.value) == msg
    # and the engine is healthy afterwards (no group left waiting at a gate)
    ok = good * 50
    _check(ok, schema, 4)


def test_input_forms(kernel):
    recs = synth.records("cfg3", 300)
    a = P.deserialize_array_threaded(recs, SCHEMAS["cfg3"], 2)
    b = P.deserialize_array_threaded([bytearray(r) for r in recs], SCHEMAS["cfg3"], 2)   # bytearray is copied (PyBackedBytes)
    for x, y in zip(a, b):
        assert_batches_identical(x, y)
    data, offsets = c_walker.pack(recs)
    c = cabi.decode_packed(data, offsets, SCHEMAS["cfg3"], 2, kernel=kernel)
    for x, y in zip(a, c):
        assert_batches_identical(x, y)


def test_arrow_binary_array_input():
    """Records already in an Arrow BinaryArray / LargeBinaryArray (zero-copy ingest), incl. sliced arrays."""
    recs = synth.records("full", 2500)
    exp = c_walker.decode_threaded(recs, SCHEMAS["full"], 4)
    for typ in (pa.binary(), pa.large_binary()):
        arr = pa.array(recs, type=typ)
        got = P.deserialize_binary_array(arr, SCHEMAS["full"], 4)
        assert len(got) == 4
        for g, e in zip(got, exp):
            assert_batches_identical(g, e)
    sl = pa.array([b"junk"] * 7 + recs + [b"x"], type=pa.binary()).slice(7, len(recs))
    for g, e in zip(P.deserialize_binary_array(sl, SCHEMAS["full"], 4), exp):
        assert_batches_identical(g, e)
    ch = pa.chunked_array([pa.array(recs[:1000], type=pa.binary()), pa.array(recs[1000:], type=pa.binary())])
    for g, e in zip(P.deserialize_binary_array(ch, SCHEMAS["full"], 4), exp):
        assert_batches_identical(g, e)
    empty = P.deserialize_binary_array(pa.array([], type=pa.binary()), SCHEMAS["full"], 3)
    assert [b.num_rows for b in empty] == [0]
    with pytest.raises(ValueError):
        P.deserialize_binary_array(pa.array([recs[0], None], type=pa.binary()), SCHEMAS["full"], 1)


def test_records_larger_than_the_lds_window():
    # 256 x 70 KB strings overflow any LDS window -> the global-memory read path of the same kernels
    s = SCHEMAS["flat_primitives"]
    vals = [{"i": i, "l": i, "f": 0.5, "d": 0.25, "b": i % 2 == 0, "s": chr(97 + i % 26) * (70000 + i)} for i in range(300)]
    _check(cases._enc(s, vals), s, 2)
    # mixed: one huge record among small ones
    vals = [{"i": i, "l": i, "f": 0.5, "d": 0.25, "b": False, "s": "q" * (200000 if i == 77 else i % 9)} for i in range(600)]
    _check(cases._enc(s, vals), s, 1)


@pytest.mark.parametrize("name,n", [("flat4", 1_000_000), ("cfg3", 1_000_000), ("full", 1_000_000)])
def test_baseline_configs_1m(name, n, kernel):
    """BASELINE.json configs 2-4 at 1M records: full buffer identity against the oracle."""
    data, offsets = fastgen.generate(name, n)
    got = cabi.decode_packed(data, offsets, SCHEMAS[name], 8, kernel=kernel)
    cs = c_walker.CompiledSchema(SCHEMAS[name])
    exp = c_walker.decode_packed(cs, data, offsets, 8, threaded=True)
    assert [b.num_rows for b in got] == [125_000] * 8
    for g, e in zip(got, exp):
        assert_batches_identical(g, e)


def test_full_schema_10m_properties(kernel):
    """BASELINE.json config 4 at full size (the size bench.py runs): FULL buffer identity against the oracle on every
    buffer of every node of every chunk (vectorised compare, arrow_compare.py) + size-independent properties."""
    n = 10_000_000
    data, offsets = fastgen.generate("full", n)
    got = cabi.decode_packed(data, offsets, SCHEMAS["full"], 8, kernel=kernel)
    assert sum(b.num_rows for b in got) == n and len(got) == 8
    cs = c_walker.CompiledSchema(SCHEMAS["full"])
    exp = c_walker.decode_packed(cs, data, offsets, 8, threaded=True)
    total_str = 0
    for g, e in zip(got, exp):
        assert_batches_identical(g, e)                                   # every buffer, bit-masked bitmaps
        for col in ("name", "class"):
            o = np.frombuffer(g.column(col).buffers()[1], dtype=np.int32, count=g.num_rows + 1)
            assert o[0] == 0 and np.all(np.diff(o) >= 0)
            total_str += int(o[-1])
        lo = np.frombuffer(g.column("emails").buffers()[1], dtype=np.int32, count=g.num_rows + 1)
        assert lo[0] == 0 and lo[-1] == len(g.column("emails").values) and np.all(np.diff(lo) >= 0) and np.diff(lo).max() <= 3
        assert g.column("created_at").null_count == 0 and g.column("created_at").buffers()[0] is None
    assert total_str > 0
    # every input byte of every string column is accounted for: re-decode of a slice equals the slice
    part = cabi.decode_packed(data[: int(offsets[1000])], offsets[:1001], SCHEMAS["full"], 1, kernel=kernel)[0]
    assert part.equals(got[0].slice(0, 1000))


def test_long_and_short_strings_everywhere():
    """Strings of 0..3000 bytes at the top level, inside a nullable record, in arrays, nested arrays and as map keys /
    values (cases.long_string_case): every length class of the per-lane string copy, wave spans from a few bytes to
    tens of KB, on both kernel forms."""
    s, recs = cases.long_string_case()
    for k in (1, 4):
        _check(recs, s, k)


@pytest.mark.parametrize("pct,pad", [(100, 0), (103, 64), (400, 65536)])
def test_any_lds_window_size(pct, pad, monkeypatch):
    """The LDS input window from tighter than the mean tile (most tiles are then walked from global memory) to far
    larger than any tile: same buffers (RUHVRO_HIP_WIN_PCT / RUHVRO_HIP_WIN_PAD, read per call)."""
    monkeypatch.setenv("RUHVRO_HIP_WIN_PCT", str(pct))
    monkeypatch.setenv("RUHVRO_HIP_WIN_PAD", str(pad))
    for name, n, k in (("full", 5003, 7), ("cfg3", 3000, 2), ("array_and_map", 2500, 3)):
        _check(synth.records(name, n, seed=3), SCHEMAS[name], k)


def test_stats_json_line(tmp_path):
    """RUHVRO_HIP_STATS=1: one JSON line per host decode call on stderr (the per-call stats struct of the C ABI, with the
    per-shard entries of a call dealt to several devices) -- SURVEY.md section 5's metrics row."""
    import json
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import torch; import pyruhvro_amd as P\n"
            "from avrogen import synth; from avrogen.schemas import SCHEMAS\n"
            "recs = synth.records('full', 3000)\n"
            "P.deserialize_array_threaded(recs, SCHEMAS['full'], 4)\n"
            "P.set_devices([0, 0]); P.deserialize_array_threaded(recs, SCHEMAS['full'], 4)\n"
            "try:\n    P.deserialize_array_threaded(recs[:10] + [b'\\x80'], SCHEMAS['full'], 2)\nexcept ValueError:\n    pass\n"
            ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RUHVRO_HIP_STATS="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = [json.loads(l) for l in p.stderr.splitlines() if l.startswith('{"ruhvro_hip"')]
    assert len(lines) == 3, p.stderr
    a, b, c = lines
    assert a["ruhvro_hip"] == "rh_decode" and a["rc"] == 0 and a["stats"]["records"] == 3000 and a["stats"]["chunks"] == 4
    assert a["stats"]["emit_kernel_ms"] > 0 and a["stats"]["total_ms"] >= a["stats"]["emit_kernel_ms"] and a["stats"]["output_bytes"] > 0
    assert [d["device"] for d in b["devices"]] == [0, 0] and sum(d["stats"]["records"] for d in b["devices"]) == 3000
    assert c["rc"] == 2                                       # RH_ERR_DECODE: the line is printed for failed calls too


def test_streaming_hand_over_of_the_list(tmp_path):
    """Large lists are handed to the engine WHILE they are extracted (rh_opts.ready / gathered, pymodule.cpp): the same
    batches, the same TypeError for a non-bytes element wherever it sits, bytearray elements copied, the reference's
    message for a malformed record.  PYRUHVRO_STREAM_MIN is read once per process, so the streamed calls run in a child
    (the whole GPU suite also passes with PYRUHVRO_STREAM_MIN=1, scripts/gpu_r03ak.sh)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, pyruhvro_amd as P
from arrow_compare import assert_batches_identical
from avrogen import synth
from avrogen.schemas import SCHEMAS
from oracle import c_walker
S = SCHEMAS["full"]
recs = synth.records("full", 20011)
for k in (2, 8, 100):
    got = P.deserialize_array_threaded(recs, S, k)
    for g, e in zip(got, c_walker.decode_threaded(recs, S, k)):
        assert_batches_identical(g, e)
mixed = [bytearray(r) if i %% 7 == 0 else r for i, r in enumerate(recs[:3000])]
for g, e in zip(P.deserialize_array_threaded(mixed, S, 3), c_walker.decode_threaded(recs[:3000], S, 3)):
    assert_batches_identical(g, e)
for pos in (0, 1, 8191, 8192, 15000, 20010):
    bad = list(recs); bad[pos] = "not bytes"
    try:
        P.deserialize_array_threaded(bad, S, 8); raise SystemExit("no TypeError")
    except TypeError as e:
        assert "list element %%d" %% pos in str(e), str(e)
for pos in (0, 9000, 20010):
    bad = list(recs); bad[pos] = recs[pos][:5]
    try:
        P.deserialize_array_threaded(bad, S, 8); raise SystemExit("no ValueError")
    except ValueError as e:
        try:
            c_walker.decode_threaded(bad, S, 8); raise SystemExit("oracle accepted it")
        except ValueError as eo:
            assert str(e) == str(eo), (str(e), str(eo))
print("streamed ok")
''' % (root, os.path.join(root, "tests"))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYRUHVRO_STREAM_MIN="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "streamed ok" in p.stdout, p.stdout + p.stderr
