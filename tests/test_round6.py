"""Round 6: the walks off their tuned value distribution.  Wider single-read wire forms, tiles past the LDS
window staged in record ranges, records past the window walked through a sliding window, wide schemas (wave counters)."""
import json

import numpy as np
import pytest

import cases
from arrow_compare import assert_batches_identical
from avrogen import fastgen, synth
from avrogen.schemas import SCHEMAS
from oracle import c_walker, py_walker

import pyruhvro_amd as P
from pyruhvro_amd import cabi

KERNELS = {"generic": cabi.KERNEL_GENERIC, "specialized": cabi.KERNEL_SPECIALIZED}


@pytest.fixture(params=sorted(KERNELS))
def kernel(request):
    old = P.set_kernel_mode(request.param)
    yield KERNELS[request.param]
    P.set_kernel_mode(old)


def _check(recs, schema, k):
    got = P.deserialize_array_threaded(recs, schema, k)
    exp = c_walker.decode_threaded(recs, schema, k)
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        g.validate(full=True)
        assert_batches_identical(g, e)
    return got


# ---- CPU: the two oracles agree on the new cases (oracle independence) -------------------------------------
@pytest.mark.parametrize("case", cases.wide_form_cases(), ids=lambda c: c[0])
def test_oracles_agree_on_the_wide_form_cases(case):
    name, schema, recs = case
    recs = recs[:650] if name == "wide_forms" else recs          # (the pure-Python walker and megabyte strings)
    assert_batches_identical(c_walker.decode(recs, schema), py_walker.decode(recs, schema))


@pytest.mark.parametrize("name", ["full_realistic", "full_skewed", "wide97"])
def test_oracles_agree_on_the_round6_generators(name):
    recs = synth.records(name, 120, seed=3)
    assert_batches_identical(c_walker.decode(recs, SCHEMAS[name]), py_walker.decode(recs, SCHEMAS[name]))


# ---- GPU ------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", cases.wide_form_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("k", [1, 3])
def test_wide_wire_forms(case, k, kernel):
    name, schema, recs = case
    _check(recs, schema, k)


@pytest.mark.gpu
def test_fast_forms_cover_production_varints(kernel):
    """Microsecond timestamps behind a branch byte, epoch seconds as int, snowflake ids, 3-byte lengths and block counts: no
    wavefront is walked twice, no tile takes the careful emit walk (rh_engine_counters, summed by rh_k_publish)."""
    n = 200_000
    data, offsets = fastgen.generate("full_realistic", n)
    # keep the records whose every length / count fits the forms AND whose tile fits a window: drop the 8 KiB notes and big arrays
    lens = np.diff(offsets.astype(np.int64))
    keep = np.flatnonzero(lens < 2000)
    recs = [bytes(data[int(offsets[i]): int(offsets[i + 1])]) for i in keep[:150_000]]
    _check(recs[:5000], SCHEMAS["full_realistic"], 3)
    import torch
    d, o = c_walker.pack(recs)
    d_data = torch.zeros(len(d) + 64, dtype=torch.uint8, device="cuda:0")
    d_data[: len(d)].copy_(torch.from_numpy(d.copy()))
    d_off = torch.from_numpy(o.view(np.int64).copy()).to("cuda:0")
    for _ in range(2):      # (the second call is a single-submission call: the one rh_k_publish counts)
        c0 = cabi.engine_counters()
        r = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(o[-1]), len(recs), SCHEMAS["full_realistic"], 4, device=0,
                               stream=torch.cuda.current_stream().cuda_stream, kernel=kernel)
        got = r.to_host()
        r.free()
        c1 = cabi.engine_counters()
    for g, e in zip(got, c_walker.decode_threaded(recs, SCHEMAS["full_realistic"], 4)):
        assert_batches_identical(g, e)
    assert c1["tiles"] > c0["tiles"]
    if kernel == cabi.KERNEL_SPECIALIZED:
        assert c1["rewalked_waves"] == c0["rewalked_waves"] and c1["careful_tiles"] == c0["careful_tiles"]


# ---- tiles and records past the LDS window -----------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("win", [8192, 16384, 40960])
def test_tiles_past_the_window_are_walked_in_ranges(win, kernel, monkeypatch):
    """Record sizes roughly log-normal and correlated in runs (avrogen full_skewed): with a window of `win` bytes most tiles hold
    two to a dozen ranges, some records are ranges of their own.  Same buffers; the specialised kernels stage every such tile
    through the window (RH_CTR_SUBTILED_TILES == RH_CTR_OVER_WINDOW_TILES > 0)."""
    monkeypatch.setenv("RUHVRO_HIP_WIN_BYTES", str(win))
    data, offsets = fastgen.generate("full_skewed", 30_011)
    for k in (1, 5):
        got = cabi.decode_packed(data, offsets, SCHEMAS["full_skewed"], k, kernel=kernel)
        exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS["full_skewed"]), data, offsets, k, threaded=True)
        for g, e in zip(got, exp):
            assert_batches_identical(g, e)
    import torch
    d_data = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda:0")
    d_data[: len(data)].copy_(torch.from_numpy(data.copy()))
    d_off = torch.from_numpy(offsets.view(np.int64).copy()).to("cuda:0")
    for _ in range(2):
        c0 = cabi.engine_counters()
        r = cabi.decode_device(d_data.data_ptr(), d_off.data_ptr(), int(offsets[-1]), len(offsets) - 1, SCHEMAS["full_skewed"], 3, device=0,
                               stream=torch.cuda.current_stream().cuda_stream, kernel=kernel)
        r.free()
        c1 = cabi.engine_counters()
    assert c1["over_window_tiles"] > c0["over_window_tiles"]
    if kernel == cabi.KERNEL_SPECIALIZED:
        assert c1["subtiled_tiles"] - c0["subtiled_tiles"] == c1["over_window_tiles"] - c0["over_window_tiles"]
        assert c1["careful_tiles"] == c0["careful_tiles"]           # past the window is not an anomaly: the fast walks, trust and dense lists stay


@pytest.mark.gpu
@pytest.mark.parametrize("case", cases.giant_record_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("win", [0, 8192, 40960])
def test_records_past_the_window_slide(case, win, kernel, monkeypatch):
    """Arrays of tens of thousands of items, megabyte strings: a record larger than the window is a range of its own whose window
    follows the cursor (walk.h SlideSrc).  win = 0: nothing is ever staged (every read is served from global memory)."""
    if win:
        monkeypatch.setenv("RUHVRO_HIP_WIN_BYTES", str(win))
    name, schema, recs = case
    for k in (1, 3):
        _check(recs, schema, k)


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["in_a_range", "in_a_sliding_record", "behind_a_sliding_record"])
def test_errors_past_the_window(where, kernel, monkeypatch):
    """A malformed record inside a tile that is walked in ranges: the reference's message, the lowest failing record."""
    monkeypatch.setenv("RUHVRO_HIP_WIN_BYTES", "8192")
    name, schema, recs = cases.giant_record_cases()[0]
    recs = list(recs)
    if where == "in_a_range":
        recs[30] = recs[30][:-3]
        recs[100] = recs[100][:5]
    elif where == "in_a_sliding_record":
        recs[64] = recs[64][: len(recs[64]) // 2]              # the 9,000-item record, cut inside an array
    else:
        recs[66] = recs[66][:4]
    with pytest.raises(ValueError) as g:
        P.deserialize_array_threaded(recs, schema, 2)
    with pytest.raises(ValueError) as e:
        c_walker.decode_threaded(recs, schema, 2)
    assert str(g.value) == str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("name,n", [("full_realistic", 400_000), ("full_realistic_heavy", 20_000), ("full_skewed", 400_000)])
def test_round6_workloads_buffer_identity(name, n, kernel):
    data, offsets = fastgen.generate(name, n)
    got = cabi.decode_packed(data, offsets, SCHEMAS[name], 8, kernel=kernel)
    exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS[name]), data, offsets, 8, threaded=True)
    for g, e in zip(got, exp):
        assert_batches_identical(g, e)


# ---- wide schemas (a 97-column record was refused; the reference has no width limit) ---------------------
def test_schemas_of_any_width_compile():
    """More than 64 scanned counters -> the schema is compiled wide (wave counters numbered behind the per-lane ones); 97, 200,
    400 nullable string columns + 12 arrays, and a 2,000-column record, all translate and take a schema program."""
    from avrogen.schemas import wide_schema
    for n in (97, 200, 400, 2000):
        sj = json.dumps(wide_schema(n))
        assert len(P.arrow_schema(sj)) == n + 12
    recs = synth.records("wide97", 40, seed=9)
    assert_batches_identical(c_walker.decode(recs, SCHEMAS["wide97"]), py_walker.decode(recs, SCHEMAS["wide97"]))


def test_wide_schema_kernels_compile_in_seconds(tmp_path, monkeypatch):
    """The specialised decode kernels of the 200-column schema from an EMPTY kernel cache (hiprtc, gfx950, no GPU needed): the
    size / emit pair and the ranged pair, each kernel a compile job of its own, side by side -- ready in about a minute on 8 vCPUs
    (round 5: the unrolled 96-counter emit kernel alone took hiprtc 5-8 minutes).  The size / emit pair and rh_spec_size_r take
    5-20 s; rh_spec_emit_r is the long one: 42 s while its walk was a function of its own, ~60 s since that walk is inlined
    (its context lived in scratch memory: 1M records 12.8 -> 6.0 ms).  The bound leaves room for a slower host."""
    import subprocess
    import sys
    import time
    import os
    env = dict(os.environ, RUHVRO_HIP_KERNEL_CACHE=str(tmp_path), AMD_COMGR_CACHE="0", RUHVRO_HIP_PREBUILD_FUSED="0")
    code = ("import time; from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS; t = time.time(); "
            "assert cabi.prebuild(SCHEMAS['wide200']) is False; print('secs', time.time() - t)")
    t = time.time()
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    secs = float(out.stdout.split("secs")[1])
    ncpu = len(os.sched_getaffinity(0))
    print(f"wide200: four specialised kernels from an empty cache in {secs:.1f} s on {ncpu} cpus")
    # (60-85 s by host and load on 8 vCPUs; the bound only has to tell seconds from round 5's minutes, and a loaded CI host must
    #  not turn the suite red over it)
    assert secs < (200 if ncpu >= 8 else 200 * 8 / max(ncpu, 1)), f"{secs:.1f} s on {ncpu} cpus"
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]) == 4       # size, emit, size_r, emit_r
    # ... and none of them keeps anything in scratch memory: the ranged kernels reach their walk through ONE inlined call site
    # (as a function of its own the walk kept its captured context -- counters, lane state -- in scratch: 38 scratch accesses per
    # column and wavefront, wide200 12.8 instead of 6.0 ms per 1M records)
    from test_specialize import _kernel_notes
    notes = {}
    for f in os.listdir(tmp_path):
        if f.endswith(".hsaco"):
            notes.update(_kernel_notes(os.path.join(tmp_path, f)))
    assert set(notes) == {"rh_spec_size", "rh_spec_emit", "rh_spec_size_r", "rh_spec_emit_r"}
    for name, md in notes.items():
        assert md["private_segment_fixed_size"] == 0 and md["vgpr_spill_count"] == 0, (name, md)


def test_generated_source_of_wide_and_giant_friendly_schemas():
    """What the generator adds for round 6's input classes: top-up hooks between the columns of a WIDE schema only (walk.h
    h_topup: lane windows / the sliding window), and the counter of a list whose body is one plain string (Spec::dense_str: the
    item scan decodes such candidates by hand) -- not for a list of nullable strings or of records."""
    wide = cabi.kernel_source(SCHEMAS["wide200"])
    full = cabi.kernel_source(SCHEMAS["full"])
    assert wide.count("h_topup(c, src, L);") >= 7 and "h_topup(" not in full
    assert "#define RH_WIDE_SCHEMA" in wide and "#define RH_WIDE_SCHEMA" not in full
    name, schema, recs = cases.giant_record_cases()[0]          # tags: array<string>, nums: array<["null","int"]>, recs: array<record>, m: map<int>
    src = cabi.kernel_source(schema)
    body = src[src.index("static constexpr int dense_str(int list)"):]
    body = body[: body.index("default: return -1;")]
    assert body.count("case ") == 1 and "case 0: return" in body, body


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["wide97", "wide200", "wide400"])
@pytest.mark.parametrize("k", [1, 3, 5])
def test_wide_schemas(name, k, kernel):
    """97 / 200 / 400 nullable string columns + 12 arrays of strings (121 / 224 / 424 scanned counters), both kernel forms, a
    ragged last tile; the records are 0.8 - 2.8 KB, so most 64-record tiles are walked in ranges."""
    n = {"wide97": 3001, "wide200": 1537, "wide400": 1037}[name]
    data, offsets = fastgen.generate(name, n)
    got = cabi.decode_packed(data, offsets, SCHEMAS[name], k, kernel=kernel)
    exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS[name]), data, offsets, k, threaded=True)
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        g.validate(full=True)
        assert_batches_identical(g, e)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 3])
def test_wide_schema_lane_windows(k, kernel):
    """Wide records that do not look alike: strings longer than a lane's 192-byte slice of the window and longer than the window,
    arrays of hundreds of items (dense lists whose item lanes read another lane's record), all-null rows between 3 KB ones, a
    record larger than the window in the middle of a direct tile, and a malformed record -- the lane windows of walk.h SlideSrc."""
    from avrogen import synth
    from avrogen.encoder import to_datum
    from oracle.avro_schema import parse_schema
    sc = parse_schema(SCHEMAS["wide97"])
    gen = synth.GENERATORS["wide97"]
    recs = []
    for r in range(700):
        v = gen(77, r)
        if r % 7 == 0:
            v = {key: ([] if key.startswith("a") else None) for key in v}             # a row of nulls and empty arrays
        if r % 11 == 3:
            v[f"c{r % 97}"] = "L" * (200 + 37 * (r % 40))                               # 200 .. 1,643 bytes: around and beyond a slice
        if r % 53 == 5:
            v["c50"] = "W" * 30_000                                                     # beyond the whole window
        if r % 17 == 4:
            v[f"a{r % 12}"] = [f"item-{r}-{j}" * (1 + j % 4) for j in range(150 + r % 200)]
        if r == 333:
            v["a5"] = [f"giant-{j}" for j in range(40_000)]                             # a record of its own sliding range
        recs.append(to_datum(sc, v))
    _check(recs, SCHEMAS["wide97"], k)
    bad = list(recs)
    bad[400] = bad[400][: len(bad[400]) // 2]
    with pytest.raises(ValueError) as g:
        P.deserialize_array_threaded(bad, SCHEMAS["wide97"], k)
    with pytest.raises(ValueError) as e:
        c_walker.decode_threaded(bad, SCHEMAS["wide97"], k)
    assert str(g.value) == str(e.value)


@pytest.mark.gpu
def test_wide_schema_mostly_null_fits_the_window(kernel):
    """A wide schema whose columns are almost all null: small records, the tiles fit the window (the staged fast walk + wave
    counters, no ranges); and with malformed records: the reference's message, the lowest failing record."""
    from avrogen.encoder import to_datum
    from oracle.avro_schema import parse_schema
    sc = parse_schema(SCHEMAS["wide97"])
    vals = []
    for r in range(2000):
        v = {f"c{i}": (f"v{r}.{i}" if (r * 31 + i) % 23 == 0 else None) for i in range(97)}
        v.update({f"a{j}": ([f"x{r}"] if (r + j) % 7 == 0 else []) for j in range(12)})
        vals.append(v)
    recs = [to_datum(sc, v) for v in vals]
    _check(recs, SCHEMAS["wide97"], 3)
    bad = list(recs)
    bad[1500] = bad[1500][:40]
    bad[700] = bad[700][:-1] + b"\x7f"
    with pytest.raises(ValueError) as g:
        P.deserialize_array_threaded(bad, SCHEMAS["wide97"], 2)
    with pytest.raises(ValueError) as e:
        c_walker.decode_threaded(bad, SCHEMAS["wide97"], 2)
    assert str(g.value) == str(e.value)


@pytest.mark.gpu
def test_a_tile_past_the_window_without_the_ranged_pair_is_repeated_on_the_generic_kernels(monkeypatch):
    """AUTO kernels, a schema whose history knows nothing of large tiles, RUHVRO_HIP_RANGED=0 (the pair is never launched): the
    size kernel refuses the call (LF_NEED_RANGED), the engine repeats it on the generic kernels -- same buffers."""
    monkeypatch.setenv("RUHVRO_HIP_RANGED", "0")
    monkeypatch.setenv("RUHVRO_HIP_WIN_BYTES", "8192")
    old = P.set_kernel_mode("auto")
    try:
        data, offsets = fastgen.generate("full_skewed", 60_000)
        cabi.prebuild(SCHEMAS["full_skewed"])
        c0 = cabi.engine_counters()
        got = cabi.decode_packed(data, offsets, SCHEMAS["full_skewed"], 4, kernel=cabi.KERNEL_AUTO)
        c1 = cabi.engine_counters()
        exp = c_walker.decode_packed(c_walker.CompiledSchema(SCHEMAS["full_skewed"]), data, offsets, 4, threaded=True)
        for g, e in zip(got, exp):
            assert_batches_identical(g, e)
        assert c1["ranged_retries"] > c0["ranged_retries"]
    finally:
        P.set_kernel_mode(old)


# ---- nesting beyond rounds 1-5's limits ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", cases.deep_nesting_cases(), ids=lambda c: c[0])
def test_oracles_agree_on_deep_nesting(case):
    name, schema, recs = case
    assert_batches_identical(c_walker.decode(recs, schema), py_walker.decode(recs, schema))
    assert P.arrow_schema(schema) is not None            # the engine's front-end takes the schema (8 / 30 / 8 levels were the limits)


@pytest.mark.gpu
@pytest.mark.parametrize("case", cases.deep_nesting_cases(), ids=lambda c: c[0])
def test_deep_nesting(case, kernel):
    """12 nested arrays, 36 nested nullable records, 10 nested N-variant unions: 64-bit bit stacks and a 128-bit selector stack
    (walk.h RH_DEEP) in the interpreter and in the specialised kernels of such schemas."""
    name, schema, recs = case
    for k in (1, 3):
        _check(recs, schema, k)
