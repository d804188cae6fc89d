"""GPU parity of the Arrow -> Avro direction: pyruhvro_amd.serialize_record_batch (rh_encode, HIP
kernels rh_e_size / rh_k_scan / rh_e_emit) vs the encode oracle (oracle/py_encoder.py, the restatement of
ruhvro/src/fast_encode.rs + serialize.rs), datum for datum, byte for byte.  Needs an MI355X."""
import json
import os

import numpy as np
import pyarrow as pa
import pytest

import cases
import random_cases
from avrogen import fastgen, synth
from avrogen.schemas import SCHEMAS
from oracle import c_walker, py_encoder

import pyruhvro_amd as P
from pyruhvro_amd import cabi

pytestmark = pytest.mark.gpu

KERNELS = {"generic": cabi.KERNEL_GENERIC, "specialized": cabi.KERNEL_SPECIALIZED}


@pytest.fixture(params=sorted(KERNELS), autouse=True)
def kernel(request):
    """Every test runs on both kernel forms: the generic schema-program interpreter (encode.hip) and the
    schema-specialised kernels (specialize.cpp -> hiprtc); same handlers (encode_walk.h), same bytes."""
    old = P.set_kernel_mode(request.param)
    yield KERNELS[request.param]
    P.set_kernel_mode(old)


GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")


def _same(got, exp):
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert g.type == pa.binary()
        g.validate(full=True)
        assert g.null_count == 0 and g.offset == 0
        assert g.equals(e), (len(g), len(e))
        # buffers too: offsets start at 0 and the data buffer holds exactly the datums
        go = np.frombuffer(g.buffers()[1], dtype=np.int32, count=len(g) + 1)
        eo = np.frombuffer(e.buffers()[1], dtype=np.int32, count=e.offset + len(e) + 1)[e.offset:]
        assert np.array_equal(go, eo - eo[0])


def _check(batch, schema, k):
    got = P.serialize_record_batch(batch, schema, k)
    _same(got, py_encoder.serialize_record_batch(batch, schema, k))
    return got


def _datums(arrays):
    return [b for a in arrays for b in a.to_pylist()]


@pytest.mark.parametrize("name", sorted(synth.GENERATORS))
def test_generated_records_reencode_byte_for_byte(name):
    recs = synth.records(name, 1500, seed=11)
    batch = c_walker.decode(recs, SCHEMAS[name])
    for k in (1, 4):
        out = _check(batch, SCHEMAS[name], k)
        assert _datums(out) == recs            # the generators write the reference's single-block form
    assert P.serialize_record_batch_spawn(batch, SCHEMAS[name], 3)[0].equals(
        py_encoder.serialize_record_batch(batch, SCHEMAS[name], 3)[0])


def test_golden_datums():
    g = json.load(open(GOLDEN))
    for v in g["vectors"]:
        schema = json.dumps(g["schemas"][v["schema"]])
        rec = bytes.fromhex(v["hex"])
        batch = c_walker.decode([rec], schema)
        out = _check(batch, schema, 1)
        assert _datums(out) == [rec[: v.get("consumed", len(rec))]], v["name"]


@pytest.mark.parametrize("case", cases.nesting_cases() + cases.differential_cases() + cases.wire_cases(), ids=lambda c: c[0])
def test_nested_and_differential_schemas(case):
    schema, recs = case[1], case[2]
    batch = c_walker.decode(recs, schema)
    for k in (1, 3):
        _check(batch, schema, k)


@pytest.mark.parametrize("seed", range(24))
def test_random_schemas(seed):
    schema, recs = random_cases.random_case(seed, 300)
    batch = c_walker.decode(recs, schema)
    out = _check(batch, schema, 2)
    # and back through the GPU decoder: decode(encode(x)) == x
    back = P.deserialize_array(_datums(out), schema)
    assert back.equals(batch)


def test_round_trip_through_both_gpu_directions():
    recs = synth.records("full", 20000, seed=3)
    batches = P.deserialize_array_threaded(recs, SCHEMAS["full"], 5)
    out = []
    for b in batches:
        out += _datums(P.serialize_record_batch(b, SCHEMAS["full"], 2))
    assert out == recs


def test_columns_matched_by_name_and_reference_error_texts():
    recs = synth.records("cfg3", 300)
    batch = c_walker.decode(recs, SCHEMAS["cfg3"])
    order = (4, 2, 0, 3, 1)
    shuffled = pa.RecordBatch.from_arrays([batch.column(i) for i in order], names=[batch.schema.names[i] for i in order])
    assert _datums(P.serialize_record_batch(shuffled, SCHEMAS["cfg3"], 2)) == recs      # fast_encode.rs:155-181
    extra = shuffled.append_column("unused", pa.array(range(300)))
    assert _datums(P.serialize_record_batch(extra, SCHEMAS["cfg3"], 2)) == recs
    with pytest.raises(ValueError) as ei:
        P.serialize_record_batch(batch.drop_columns(["age"]), SCHEMAS["cfg3"], 1)
    assert str(ei.value) == ("Arrow struct missing column 'age' required by Avro schema. "
                             'Available columns: ["id", "name", "s", "class"]')


def test_data_dependent_errors_carry_the_reference_text_and_lowest_row_wins():
    s = SCHEMAS["t_enum"]
    syms = ["A"] * 700
    syms[431] = "Z"
    syms[650] = "QQ"
    bad = pa.RecordBatch.from_arrays([pa.array(syms)], names=["e"])
    for k in (1, 3):
        with pytest.raises(ValueError) as ei:
            P.serialize_record_batch(bad, s, k)
        assert str(ei.value) == "fast_encode: enum symbol 'Z' not in schema"        # fast_encode.rs:575-577
        with pytest.raises(ValueError) as eo:
            py_encoder.serialize_record_batch(bad, s, k)
        assert str(eo.value) == str(ei.value)
    su = SCHEMAS["t_union"]
    batch = c_walker.decode(cases._enc(su, [{"u": None}, {"u": "x"}] * 200), su)
    u = batch.column(0)
    ids = np.frombuffer(u.buffers()[1], dtype=np.int8, count=len(u)).copy()
    ids[333] = 9
    ids[390] = -3
    broken = pa.UnionArray.from_sparse(pa.array(ids, type=pa.int8()), [u.field(i) for i in range(u.type.num_fields)],
                                       field_names=[u.type.field(i).name for i in range(u.type.num_fields)])
    with pytest.raises(ValueError) as ei:
        P.serialize_record_batch(pa.RecordBatch.from_arrays([broken], names=["u"]), su, 2)
    assert str(ei.value) == "fast_encode: union type_id 9 out of range"          # fast_encode.rs:540-542


def test_chunking_and_empty_batches():
    recs = synth.records("flat4", 7)
    batch = c_walker.decode(recs, SCHEMAS["flat4"])
    for k, want in ((1, [7]), (3, [2, 2, 3]), (0, [7]), (50, [1] * 7)):
        got = _check(batch, SCHEMAS["flat4"], k)
        assert [len(a) for a in got] == want                                     # serialize.rs:15-30
    empty = c_walker.decode([], SCHEMAS["flat4"])
    got = _check(empty, SCHEMAS["flat4"], 4)
    assert [len(a) for a in got] == [0]


@pytest.mark.parametrize("name", ["full", "cfg3", "array_and_map", "nullable_primitives", "nested_struct"])
def test_sliced_inputs_honour_every_offset(name):
    """A sliced batch has non-zero offsets on the struct, its children and (for bitmaps) non-byte-aligned bit
    offsets: value(row) in the reference applies them all."""
    recs = synth.records(name, 1000, seed=5)
    batch = c_walker.decode(recs, SCHEMAS[name])
    for lo, ln in ((1, 999), (13, 700), (511, 3), (999, 1), (1000, 0)):
        part = batch.slice(lo, ln)
        out = _check(part, SCHEMAS[name], 3)
        assert _datums(out) == recs[lo: lo + ln]


def test_non_nullable_leaves_ignore_validity_and_nulls_under_null_parents_are_skipped():
    """fast_encode.rs:391-399: Int/Long/... write value(row) whatever the validity bit says."""
    s = cases.ENC_VALIDITY_SCHEMA
    a = pa.array([1, None, 3, None], type=pa.int64())
    b = pa.array(["x", None, "", "zz"])
    rb = pa.RecordBatch.from_arrays([a, b], names=["a", "b"])
    _check(rb, s, 1)


def test_large_batch_matches_generator_bytes():
    """BASELINE config-4 shape at 1M rows: encode(decode(records)) == records, checked through the packed payload."""
    n = 1_000_000
    data, offsets = fastgen.generate("full", n)
    batches = cabi.decode_packed(data, offsets, SCHEMAS["full"], 1)
    out, st = P.serialize_record_batch_with_stats(batches[0], SCHEMAS["full"], 8)
    assert [len(a) for a in out] == [n // 8] * 8
    pos = 0
    for a in out:
        o = np.frombuffer(a.buffers()[1], dtype=np.int32, count=len(a) + 1)
        d = np.frombuffer(a.buffers()[2], dtype=np.uint8, count=int(o[-1]))
        r0 = pos
        pos += len(a)
        assert np.array_equal(o.astype(np.uint64), offsets[r0: pos + 1] - offsets[r0])
        assert np.array_equal(d, data[int(offsets[r0]): int(offsets[pos])])
    assert st["records"] == n and st["emit_kernel_ms"] > 0


def test_the_requested_kernel_form_is_the_one_that_runs(kernel):
    recs = synth.records("full", 3000)
    batch = c_walker.decode(recs, SCHEMAS["full"])
    out, st = P.serialize_record_batch_with_stats(batch, SCHEMAS["full"], 2)
    assert st["specialized"] == (1 if kernel == cabi.KERNEL_SPECIALIZED else 0)
    assert _datums(out) == recs


def test_rows_beyond_the_staging_window_take_the_direct_store_path():
    """rh_e_emit stages a workgroup's bytes in LDS when they fit its window and stores straight to HBM when they
    do not; both forms in one call, plus a row far larger than any window."""
    s = cases.ENC_WINDOW_SCHEMA
    n = 2000
    t = ["s%d" % i for i in range(n)]
    o = [None if i % 3 else "v" * (i % 40) for i in range(n)]
    xs = [["a" * (i % 7)] * (i % 4) for i in range(n)]
    for i in range(300, 312):
        t[i] = "L" * 9000                      # 12 rows x 9 KB: past the window of this workgroup only
    t[1500] = "H" * 300_000                    # one row larger than the whole LDS
    xs[777] = ["w" * 5000] * 20
    o[1999] = "tail" * 4000
    rb = pa.RecordBatch.from_arrays([pa.array(range(n), pa.int64()), pa.array(t), pa.array(o), pa.array(xs, pa.list_(pa.string()))],
                                    names=["id", "t", "o", "xs"])
    # the decoder's arrow schema names list items "item"-style fields; build the batch through a decode so types match
    exp = py_encoder.serialize_record_batch(rb, s, 1)
    batch = c_walker.decode(_datums(exp), s)
    for k in (1, 3, 7):
        _check(batch, s, k)


def test_string_fetch_classes_and_the_cooperative_range_limit():
    """The specialised emit kernel fetches string bytes three ways: cooperatively through the wave's LDS staging area
    (top-level column, the 64 rows' bytes within 1 KB), per lane with 8 / 16 / 32 bytes by the wave's longest string,
    and 32 bytes at a time past the first 32.  Stretches of 256 rows drive every class and both sides of the range
    limit, with nulls, empty strings and list items of every class in between."""
    s = cases.ENC_WINDOW_SCHEMA
    lens = [0, 1, 7, 8, 9, 15, 16, 17, 24, 31, 32, 33, 40, 63, 64, 65, 100]
    t, o, xs = [], [], []

    def stretch(tl, ol, xl):
        for i in range(256):
            j = len(t)
            t.append(chr(97 + j % 26) * tl(i))
            n = ol(i)
            o.append(None if n is None else chr(65 + j % 26) * n)
            xs.append([chr(48 + (j + q) % 10) * xl(i, q) for q in range(i % 4)])

    stretch(lambda i: lens[i % len(lens)], lambda i: None if i % 3 == 0 else lens[(i * 7) % len(lens)], lambda i, q: lens[(i + q) % len(lens)])
    stretch(lambda i: i % 9, lambda i: i % 8, lambda i, q: (i + q) % 9)                  # nothing over 8 bytes
    stretch(lambda i: 9 + i % 8, lambda i: None if i % 2 else 16, lambda i, q: 10 + q)   # 9..16 bytes
    stretch(lambda i: 16, lambda i: 15, lambda i, q: 17 + q)                             # 64 x 16 = the range limit exactly
    stretch(lambda i: 16 + (i % 64 == 63), lambda i: 16 if i % 64 else 17, lambda i, q: 33)   # one byte past it
    stretch(lambda i: 200 if i % 64 == 5 else 3, lambda i: None if i % 64 else 1500, lambda i, q: 2 if q else 90)
    n = len(t)
    rb = pa.RecordBatch.from_arrays([pa.array(range(n), pa.int64()), pa.array(t), pa.array(o), pa.array(xs, pa.list_(pa.string()))],
                                    names=["id", "t", "o", "xs"])
    exp = py_encoder.serialize_record_batch(rb, s, 1)
    batch = c_walker.decode(_datums(exp), s)
    for k in (1, 5):
        _check(batch, s, k)
    _check(batch.slice(37, n - 100), s, 3)


def test_enum_symbols_short_and_longer_than_sixteen_bytes():
    """The specialised kernel folds symbols of <= 16 bytes into constant compares and sends longer ones through the
    symbol table; both must find every symbol and reject near misses with the reference's message."""
    s = cases.ENC_LONG_ENUM_SCHEMA
    e = ["SHORT", "A_SYMBOL_LONGER_THAN_SIXTEEN_BYTES", "MID_LENGTH_SYM16"] * 100
    f = [None, "x", "yy", "zzz", "wwww", "vvvvv"] * 50
    rb = pa.RecordBatch.from_arrays([pa.array(e), pa.array(f)], names=["e", "f"])
    _check(rb, s, 2)
    for bad_e, bad_f, msg in (("SHORX", "x", "SHORX"), ("SHORT", "yyy", "yyy"), ("A_SYMBOL_LONGER_THAN_SIXTEEN_BYTEZ", "x", "A_SYMBOL_LONGER_THAN_SIXTEEN_BYTEZ"),
                              ("MID_LENGTH_SYM1", "x", "MID_LENGTH_SYM1"), ("SHORT", "", "")):
        e2, f2 = list(e), list(f)
        e2[77], f2[77] = bad_e, bad_f
        with pytest.raises(ValueError) as ei:
            P.serialize_record_batch(pa.RecordBatch.from_arrays([pa.array(e2), pa.array(f2)], names=["e", "f"]), s, 1)
        assert str(ei.value) == "fast_encode: enum symbol '%s' not in schema" % msg


# ---- device-resident form (rh_encode_device): Arrow buffers in HBM -> "z" arrays in HBM ---------------------------

def _device_batch(name, n, seed=3):
    """Records decoded on the GPU with the result left in HBM: (DeviceResult, records, schema json)."""
    import hipmem
    recs = synth.records(name, n, seed=seed)
    data, offsets = c_walker.pack(recs)
    d_data, d_off = hipmem.upload_packed(data, offsets)
    r = cabi.decode_device(d_data.ptr, d_off.ptr, int(offsets[-1]), len(recs), SCHEMAS[name], 1, device=0)
    return r, recs, SCHEMAS[name]


@pytest.mark.parametrize("name,n,k", [("full", 5003, 7), ("cfg3", 3000, 2), ("array_and_map", 2500, 3), ("full", 1, 1),
                                      ("nullable_primitives", 4000, 5), ("flat4", 777, 1)])
def test_device_resident_encode(name, n, k, kernel):
    """GPU decode (result stays in HBM) -> rh_encode_device reads those buffers in place -> BinaryArrays in HBM: the
    datums equal the records that went in (the generator writes canonical Avro), the oracle encoder's output for the
    same batch, and the host entry point's; the device export is walked buffer by buffer."""
    import ctypes as C
    import hipmem
    r, recs, schema = _device_batch(name, n)
    view = r.export(0)                                       # ArrowDeviceArray: device pointers
    sch = cabi.schema_struct(schema)
    enc = cabi.encode_device(C.addressof(view.array), C.addressof(sch), schema, k, device=0, kernel=kernel)
    assert enc.stats["records"] == n and enc.stats["emit_kernel_ms"] > 0 and enc.stats["h2d_ms"] >= 0
    got = enc.to_host()
    batch = r.to_host()[0]
    exp = py_encoder.serialize_record_batch(batch, schema, k)
    _same(got, exp)
    assert _datums(got) == recs
    _same(P.serialize_record_batch(batch, schema, k), exp)
    assert enc.chunks == len(exp)
    assert enc.output_bytes == sum(4 * (len(a) + 1) + len(b"".join(a.to_pylist())) for a in exp)
    # the Arrow C Device view of every chunk: i32 offsets + data, both in HBM
    for c, e in enumerate(exp):
        d = enc.export(c)
        assert d.device_type == 10 and d.array.length == len(e) and d.array.n_buffers == 3 and d.array.null_count == 0
        offs = hipmem.d2h(d.array.buffers[1], 4 * (len(e) + 1)).view(np.int32)
        eo = np.frombuffer(e.buffers()[1], dtype=np.int32, count=e.offset + len(e) + 1)[e.offset:]
        assert np.array_equal(offs, eo - eo[0])
        if offs[-1]:
            assert hipmem.d2h(d.array.buffers[2], int(offs[-1])).tobytes() == b"".join(e.to_pylist())
        C.CFUNCTYPE(None, C.POINTER(cabi.ArrowArray))(d.array.release)(C.byref(d.array))
    C.CFUNCTYPE(None, C.POINTER(cabi.ArrowArray))(view.array.release)(C.byref(view.array))
    enc.free()
    r.free()


def test_device_resident_round_trip_at_ten_million_rows(kernel):
    """The bench's device-resident configuration at its full size, both directions back to back in HBM: 10M records of the
    generate_avro.py schema -> rh_decode_device (8 chunks' worth in one batch) -> rh_encode_device on those buffers -> the
    datums and offsets that went in, chunk by chunk (the generator writes the reference's single-block form, so
    encode(decode(x)) == x byte for byte; fast_decode.rs:945-953 assert_round_trip is the reference's own form of this)."""
    import ctypes as C
    import hipmem
    if kernel != cabi.KERNEL_SPECIALIZED:
        pytest.skip("the full size runs on the kernels the bench times")
    n, k = 10_000_000, 8
    data, offsets = fastgen.generate("full", n)
    d_data, d_off = hipmem.upload_packed(data, offsets)
    r = cabi.decode_device(d_data.ptr, d_off.ptr, int(offsets[-1]), n, SCHEMAS["full"], 1, device=0, kernel=kernel)
    view = r.export(0)
    sch = cabi.schema_struct(SCHEMAS["full"])
    enc = cabi.encode_device(C.addressof(view.array), C.addressof(sch), SCHEMAS["full"], k, device=0, kernel=kernel)
    assert enc.stats["records"] == n and enc.chunks == k
    pos = 0
    for c in range(k):
        d = enc.export(c)
        rows = d.array.length
        offs = hipmem.d2h(d.array.buffers[1], 4 * (rows + 1)).view(np.int32)
        assert np.array_equal(offs.astype(np.uint64), offsets[pos: pos + rows + 1] - offsets[pos])
        got = hipmem.d2h(d.array.buffers[2], int(offs[-1]))
        assert np.array_equal(got, data[int(offsets[pos]): int(offsets[pos + rows])])
        pos += rows
        C.CFUNCTYPE(None, C.POINTER(cabi.ArrowArray))(d.array.release)(C.byref(d.array))
    assert pos == n
    C.CFUNCTYPE(None, C.POINTER(cabi.ArrowArray))(view.array.release)(C.byref(view.array))
    enc.free()
    r.free()


def test_device_resident_encode_errors(kernel):
    """Data-dependent failures of the device-resident form carry the reference's text too (the enum symbol is fetched
    from HBM for the message, fast_encode.rs:576)."""
    import ctypes as C
    import hipmem
    schema = json.dumps({"type": "record", "name": "E", "fields": [{"name": "e", "type": {"type": "enum", "name": "Suit", "symbols": ["hearts", "clubs"]}}]})
    # a device batch with a symbol the schema does not have: build the Arrow buffers by hand in HBM
    syms = [b"hearts", b"spades", b"clubs"]
    offs = np.zeros(len(syms) + 1, dtype=np.int32)
    offs[1:] = np.cumsum([len(x) for x in syms])
    d_offs = hipmem.DevBuf(256).upload(offs)
    d_data = hipmem.DevBuf(256).upload(np.frombuffer(b"".join(syms), dtype=np.uint8))
    child_bufs = (C.c_void_p * 3)(None, d_offs.ptr, d_data.ptr)
    child = cabi.ArrowArray()
    child.length = 3; child.null_count = 0; child.offset = 0; child.n_buffers = 3; child.n_children = 0
    child.buffers = C.cast(child_bufs, C.POINTER(C.c_void_p))
    top_bufs = (C.c_void_p * 1)(None)
    kids = (C.POINTER(cabi.ArrowArray) * 1)(C.pointer(child))
    top = cabi.ArrowArray()
    top.length = 3; top.null_count = 0; top.offset = 0; top.n_buffers = 1; top.n_children = 1
    top.buffers = C.cast(top_bufs, C.POINTER(C.c_void_p))
    top.children = C.cast(kids, C.POINTER(C.POINTER(cabi.ArrowArray)))
    sch = cabi.schema_struct(schema)
    with pytest.raises(ValueError) as ei:
        cabi.encode_device(C.addressof(top), C.addressof(sch), schema, 1, device=0, kernel=kernel)
    assert str(ei.value) == "fast_encode: enum symbol 'spades' not in schema"
