"""bytes, fixed, decimal, uuid, time-millis, time-micros, duration on the GPU decode path.

The reference TRANSLATES these schemas to Arrow (schema_translate.rs:58,133-140 -- restated in the oracle and matched
by the product, tested here) but never DECODES them: its direct path rejects them (fast_decode.rs:59) and the
Value-tree fallback it falls back to has no column builder for the Arrow types it just chose (complex.rs:414-431:
unimplemented!()).  So there is no reference behaviour to pin: the expected values below are known-answer vectors
worked out from the Avro 1.11 specification, the oracle restates that specification (py_walker `extended=True`), and
DESIGN.md marks parity for these types "specification-pinned, unpinned by the reference"."""
import datetime
import decimal
import json

import pyarrow as pa
import pytest

from arrow_compare import assert_batches_identical
from avrogen.encoder import Branch, Dur, Unscaled, UuidText, to_datum, zigzag
from oracle import avro_schema as S
from oracle import py_encoder, py_walker

import pyruhvro_amd as P
from pyruhvro_amd import cabi

F = lambda name, size, **kw: dict({"type": "fixed", "name": name, "size": size}, **kw)      # noqa: E731
SCHEMA = json.dumps({"type": "record", "name": "N4", "fields": [
    {"name": "b", "type": "bytes"},
    {"name": "nb", "type": ["null", "bytes"]},
    {"name": "f", "type": F("F5", 5)},
    {"name": "nf", "type": [F("F3", 3), "null"]},
    {"name": "d", "type": {"type": "bytes", "logicalType": "decimal", "precision": 10, "scale": 2}},
    {"name": "nd", "type": ["null", {"type": "bytes", "logicalType": "decimal", "precision": 38, "scale": 0}]},
    {"name": "df", "type": F("D8", 8, logicalType="decimal", precision=18, scale=4)},
    {"name": "u", "type": {"type": "string", "logicalType": "uuid"}},
    {"name": "nu", "type": ["null", F("U16", 16, logicalType="uuid")]},
    {"name": "tm", "type": {"type": "int", "logicalType": "time-millis"}},
    {"name": "tu", "type": ["null", {"type": "long", "logicalType": "time-micros"}]},
    {"name": "arr", "type": {"type": "array", "items": F("F2", 2)}},
    {"name": "m", "type": {"type": "map", "values": ["null", {"type": "bytes", "logicalType": "decimal", "precision": 5, "scale": 1}]}},
    {"name": "un", "type": ["null", "bytes", F("F4", 4), "int", {"type": "string", "logicalType": "uuid"}]},
    {"name": "rec", "type": ["null", {"type": "record", "name": "In", "fields": [
        {"name": "x", "type": F("F1", 1)}, {"name": "y", "type": {"type": "int", "logicalType": "time-millis"}},
        {"name": "z", "type": {"type": "bytes", "logicalType": "decimal", "precision": 4, "scale": 4}}]}]},
    {"name": "tail", "type": "string"},
]})
UUID = "550e8400-e29b-41d4-a716-446655440000"
UUID_BYTES = bytes.fromhex(UUID.replace("-", ""))


def _rows(n):
    out = []
    for i in range(n):
        out.append({
            "b": bytes(range(i % 40)), "nb": None if i % 3 == 0 else b"\xff" * (i % 5),
            "f": bytes([i % 256] * 5), "nf": None if i % 2 else b"xyz",
            "d": Unscaled(1234 - 77 * i), "nd": None if i % 4 == 1 else Unscaled((-1) ** i * (10 ** (i % 36)) + i, pad=i % 3 if i % 36 < 28 else 0),
            "df": Unscaled(-5 * i), "u": UuidText(UUID if i % 2 else UUID.replace("-", "").upper()),
            "nu": None if i % 5 == 0 else UuidText(UUID_BYTES[::-1]),
            "tm": (i * 7919) % 86_400_000, "tu": None if i % 7 == 0 else i * 1_000_003,
            "arr": [bytes([i % 256, j]) for j in range(i % 4)],
            "m": [(f"k{j}", None if (i + j) % 3 == 0 else Unscaled(j - i)) for j in range(i % 3)],
            "un": [None, b"bytes!" * (i % 3), Branch(2, b"FOUR"), i, Branch(4, UuidText(UUID))][i % 5],
            "rec": None if i % 3 == 2 else {"x": bytes([i % 256]), "y": i, "z": Unscaled(i % 9999)},
            "tail": f"row-{i}",
        })
    return out


def _records(n):
    sc = S.parse_schema(SCHEMA)
    return [to_datum(sc, r) for r in _rows(n)]


# ---------------------------------------------------------------------------------------------------- CPU
def test_reference_gate_rejects_what_the_gpu_path_now_accepts():
    for body in ('"bytes"', json.dumps(F("f", 4)), '{"type":"int","logicalType":"time-millis"}',
                 '{"type":"long","logicalType":"time-micros"}', '{"type":"string","logicalType":"uuid"}',
                 '{"type":"bytes","logicalType":"decimal","precision":9,"scale":2}'):
        js = '{"type":"record","name":"x","fields":[{"name":"a","type":%s}]}' % body
        avro = S.parse_schema(js)
        assert not S.is_supported(avro) and S.is_supported_extended(avro)        # fast_decode.rs:59 vs N4
        with pytest.raises(ValueError):
            S.build_tree(avro)
        assert P.arrow_schema(js).equals(S.to_arrow_schema(avro), check_metadata=True)   # schema_translate.rs:58,133-140


def test_arrow_translation_matches_the_oracle_and_the_reference_type_map():
    got = P.arrow_schema(SCHEMA)
    assert got.equals(S.to_arrow_schema(S.parse_schema(SCHEMA)), check_metadata=True)
    t = {f.name: f.type for f in got}
    assert t["b"] == pa.binary() and t["f"] == pa.binary(5) and t["d"] == pa.decimal128(10, 2) and t["df"] == pa.decimal128(18, 4)
    assert t["u"] == pa.binary(16) and t["nu"] == pa.binary(16) and t["tm"] == pa.time32("ms") and t["tu"] == pa.time64("us")
    assert [c.name for c in t["un"]] == ["null", "varbinary", "fixedsizebinary", "int", "fixedsizebinary"]       # default_field_name
    # decimal attributes that do not make a decimal keep the underlying type (apache-avro warns): precision 0 / scale > precision /
    # a fixed too small for the precision; uuid on a fixed that is not 16 bytes
    for body, want in (('{"type":"bytes","logicalType":"decimal","precision":0}', pa.binary()),
                       ('{"type":"bytes","logicalType":"decimal","precision":3,"scale":4}', pa.binary()),
                       (json.dumps(F("q", 2, logicalType="decimal", precision=5)), pa.binary(2)),
                       (json.dumps(F("q", 8, logicalType="uuid")), pa.binary(8))):
        js = '{"type":"record","name":"x","fields":[{"name":"a","type":%s}]}' % body
        assert P.arrow_schema(js).field("a").type == want == S.to_arrow_schema(S.parse_schema(js)).field("a").type
    for body in ('{"type":"bytes","logicalType":"decimal","precision":39}', json.dumps(F("q", 17, logicalType="decimal", precision=39))):
        with pytest.raises(ValueError):          # beyond Decimal128
            P.arrow_schema('{"type":"record","name":"x","fields":[{"name":"a","type":%s}]}' % body)


def test_specification_known_answers_pin_the_oracle():
    """Hand-assembled datums (Avro 1.11: bytes = length + raw; fixed = raw; decimal = big-endian two's complement of the
    unscaled value; uuid = RFC 4122 text) against literal expected values."""
    js = json.dumps({"type": "record", "name": "K", "fields": [
        {"name": "d", "type": {"type": "bytes", "logicalType": "decimal", "precision": 10, "scale": 2}},
        {"name": "df", "type": F("D2", 2, logicalType="decimal", precision=4, scale=1)},
        {"name": "b", "type": "bytes"}, {"name": "f", "type": F("F3", 3)},
        {"name": "u", "type": {"type": "string", "logicalType": "uuid"}},
        {"name": "tm", "type": {"type": "int", "logicalType": "time-millis"}},
        {"name": "tu", "type": {"type": "long", "logicalType": "time-micros"}}]})
    recs = [
        bytes.fromhex("04" "04d2") + bytes.fromhex("ff85") + bytes.fromhex("06" "00ff10") + b"abc" + bytes([72]) + UUID.encode()
        + zigzag(3_723_004) + zigzag(86_399_999_999),
        bytes.fromhex("02" "ff") + bytes.fromhex("0000") + bytes.fromhex("00") + b"\x00\x01\x02" + bytes([64]) + UUID.replace("-", "").encode()
        + zigzag(0) + zigzag(1),
    ]
    rows = py_walker.decode(recs, js, extended=True).to_pylist()
    assert rows[0] == {"d": decimal.Decimal("12.34"), "df": decimal.Decimal("-12.3"), "b": b"\x00\xff\x10", "f": b"abc", "u": UUID_BYTES,
                       "tm": datetime.time(1, 2, 3, 4000), "tu": datetime.time(23, 59, 59, 999999)}
    assert rows[1] == {"d": decimal.Decimal("-0.01"), "df": decimal.Decimal("0.0"), "b": b"", "f": b"\x00\x01\x02", "u": UUID_BYTES,
                       "tm": datetime.time(0, 0), "tu": datetime.time(0, 0, 0, 1)}
    for bad, msg in ((bytes.fromhex("22") + b"\x01" * 17, "decimal value of 17 bytes does not fit Decimal128"),
                     (bytes.fromhex("02" "01" "00"), "unexpected end of buffer (fixed)")):
        with pytest.raises(ValueError, match=msg.replace("(", r"\(").replace(")", r"\)")):
            py_walker.decode([bad], js, extended=True)


def test_encode_oracle_writes_the_specification_wire_forms():
    """The encode side of N4 (also beyond the reference): the oracle encoder against literal bytes, and
    decode(encode(x)) == x over the generator's rows (minimal decimals and canonical uuid text re-encode byte for byte)."""
    js = json.dumps({"type": "record", "name": "K", "fields": [
        {"name": "d", "type": {"type": "bytes", "logicalType": "decimal", "precision": 10, "scale": 2}},
        {"name": "df", "type": F("D2", 2, logicalType="decimal", precision=4, scale=1)},
        {"name": "b", "type": "bytes"}, {"name": "f", "type": F("F3", 3)},
        {"name": "u", "type": {"type": "string", "logicalType": "uuid"}},
        {"name": "uf", "type": F("U16", 16, logicalType="uuid")},
        {"name": "tm", "type": {"type": "int", "logicalType": "time-millis"}},
        {"name": "tu", "type": {"type": "long", "logicalType": "time-micros"}}]})
    D = decimal.Decimal
    rb = pa.RecordBatch.from_arrays([
        pa.array([D("12.34"), D("-0.01"), D("0.00"), D("1.27"), D("1.28"), D("-1.28"), D("-1.29")], pa.decimal128(10, 2)),
        pa.array([D("-12.3"), D("0.0"), D("99.9"), D("-0.1"), D("0.1"), D("1.0"), D("-1.0")], pa.decimal128(4, 1)),
        pa.array([b"\x00\xff\x10", b"", b"a", b"bc", b"d" * 40, b"e", b"f"], pa.binary()),
        pa.array([b"abc", b"\x00\x01\x02", b"xyz", b"123", b"456", b"789", b"000"], pa.binary(3)),
        pa.array([UUID_BYTES] * 7, pa.binary(16)), pa.array([UUID_BYTES[::-1]] * 7, pa.binary(16)),
        pa.array([3_723_004, 0, 1, 2, 3, 4, 86_399_999], pa.time32("ms")),
        pa.array([86_399_999_999, 1, 0, 5, 6, 7, 8], pa.time64("us"))], names=["d", "df", "b", "f", "u", "uf", "tm", "tu"])
    out = [x for a in py_encoder.serialize_record_batch(rb, js, 2, extended=True) for x in a.to_pylist()]
    assert out[0] == (bytes.fromhex("04" "04d2") + bytes.fromhex("ff85") + bytes.fromhex("06" "00ff10") + b"abc" + bytes([72]) + UUID.encode()
                      + UUID_BYTES[::-1] + zigzag(3_723_004) + zigzag(86_399_999_999))
    assert out[1][:4] == bytes.fromhex("02" "ff" "0000")          # -1 is one byte, 0 on fixed(2) is two
    assert [o[:3] for o in out[2:7]] == [bytes.fromhex("0200" "03"), bytes.fromhex("027f" "ff"), bytes.fromhex("040080"),
                                         bytes.fromhex("0280" "00"), bytes.fromhex("04ff7f")]      # d = 0, 127, 128, -128, -129 (+ the first byte of df)
    assert py_walker.decode(out, js, extended=True).to_pylist() == rb.to_pylist()
    recs = _records(300)
    batch = py_walker.decode(recs, SCHEMA, extended=True)
    enc = [x for a in py_encoder.serialize_record_batch(batch, SCHEMA, 4, extended=True) for x in a.to_pylist()]
    assert py_walker.decode(enc, SCHEMA, extended=True).equals(batch)
    assert sum(a == b for a, b in zip(enc, recs)) > 60            # the rows whose decimals / uuid text were canonical already


# ---------------------------------------------------------------------------------------------------- GPU
KERNELS = {"generic": cabi.KERNEL_GENERIC, "specialized": cabi.KERNEL_SPECIALIZED}


@pytest.fixture(params=sorted(KERNELS))
def kernel(request):
    old = P.set_kernel_mode(request.param)
    yield KERNELS[request.param]
    P.set_kernel_mode(old)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(1, 1), (64, 1), (257, 3), (1500, 8)])
def test_gpu_matches_the_specification_oracle(n, k, kernel):
    recs = _records(n)
    whole = py_walker.decode(recs, SCHEMA, extended=True)
    got = P.deserialize_array_threaded(recs, SCHEMA, k)
    kk = min(k, n)
    sz = n // kk
    assert [b.num_rows for b in got] == [sz] * (kk - 1) + [n - (kk - 1) * sz]
    for i, g in enumerate(got):
        g.validate(full=True)
        lo = i * sz
        exp = py_walker.decode(recs[lo: lo + g.num_rows], SCHEMA, extended=True)
        assert_batches_identical(g, exp)
    assert pa.Table.from_batches(got).to_pylist() == whole.to_pylist()
    old = P.set_devices([0, 0, 0])
    try:
        for g, e in zip(P.deserialize_array_threaded(recs, SCHEMA, k), got):
            assert_batches_identical(g, e)
    finally:
        P.set_devices(old)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(1, 1), (257, 3), (3000, 8)])
def test_gpu_encode_matches_the_oracle_encoder_and_round_trips(n, k, kernel):
    """Arrow -> Avro for the N4 types on the GPU (rh_encode, both kernel forms): byte for byte the oracle encoder's
    datums, and back through the GPU decoder to the batch it started from; sliced input; wrong Arrow types refused."""
    recs = _records(n)
    batch = py_walker.decode(recs, SCHEMA, extended=True)
    got = P.serialize_record_batch(batch, SCHEMA, k)
    exp = py_encoder.serialize_record_batch(batch, SCHEMA, k, extended=True)
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert g.type == pa.binary() and g.equals(e)
    datums = [x for a in got for x in a.to_pylist()]
    back = P.deserialize_array_threaded(datums, SCHEMA, 1)[0]
    assert_batches_identical(back, batch)
    if n > 100:
        sl = batch.slice(37, n - 50)
        for g, e in zip(P.serialize_record_batch(sl, SCHEMA, 2), py_encoder.serialize_record_batch(sl, SCHEMA, 2, extended=True)):
            assert g.equals(e)
        js = '{"type":"record","name":"x","fields":[{"name":"a","type":%s}]}'
        for body, arr in ((json.dumps(F("q", 4)), pa.array([b"abc"], pa.binary(3))),
                          ('{"type":"bytes","logicalType":"decimal","precision":9,"scale":2}', pa.array([decimal.Decimal("1.5")], pa.decimal128(9, 1))),
                          ('"bytes"', pa.array(["s"])), ('{"type":"int","logicalType":"time-millis"}', pa.array([1], pa.int32()))):
            with pytest.raises(ValueError, match="arrow array downcast failed"):
                P.serialize_record_batch(pa.RecordBatch.from_arrays([arr], names=["a"]), js % body, 1)


@pytest.mark.gpu
def test_gpu_error_texts(kernel):
    js = json.dumps({"type": "record", "name": "E", "fields": [
        {"name": "f", "type": F("F6", 6)}, {"name": "d", "type": {"type": "bytes", "logicalType": "decimal", "precision": 9, "scale": 0}},
        {"name": "u", "type": ["null", {"type": "string", "logicalType": "uuid"}]}]})
    good = b"sixsix" + bytes.fromhex("02" "07") + b"\x02" + bytes([72]) + UUID.encode()
    assert P.deserialize_array([good] * 3, js).to_pylist()[2] == {"f": b"sixsix", "d": decimal.Decimal(7), "u": UUID_BYTES}
    for bad, msg in ((b"six", "unexpected end of buffer (fixed)"),
                     (b"sixsix" + bytes.fromhex("22") + b"\x01" * 17 + b"\x00", "decimal value of 17 bytes does not fit Decimal128"),
                     (b"sixsix" + bytes.fromhex("02" "07") + b"\x02" + bytes([72]) + UUID.replace("e", "g", 1).encode(), "invalid uuid string"),
                     (b"sixsix" + bytes.fromhex("02" "07") + b"\x02" + bytes([70]) + UUID[:35].encode(), "invalid uuid string"),
                     (b"sixsix" + bytes.fromhex("02" "07") + b"\x02" + bytes([72]) + UUID.replace("-", "+").encode(), "invalid uuid string"),
                     (b"sixsix" + bytes.fromhex("0a" "07"), "unexpected end of buffer (string)")):
        for recs in ([bad], [good] * 70 + [bad] + [good] * 5):
            with pytest.raises(ValueError) as ei:
                P.deserialize_array_threaded(recs, js, 2)
            assert str(ei.value) == msg
            with pytest.raises(ValueError) as eo:
                py_walker.decode(recs, js, extended=True)
            assert str(eo.value) == msg


@pytest.mark.gpu
def test_gpu_encode_refuses_a_decimal_that_does_not_fit_its_fixed(kernel):
    """A Decimal128 on a fixed(N) base is written as its low N bytes.  Avro caps the precision of a fixed(N) decimal so
    that every in-precision value fits, but Arrow does not police values against the precision: a batch whose raw 128-bit
    value is not an N-byte two's complement number would silently decode to a different number -- the encoder refuses it
    (lowest row wins), like e_enum_put refuses an unknown symbol; the decoder is strict the other way round (E_DECIMAL)."""
    js = json.dumps({"type": "record", "name": "x", "fields": [
        {"name": "a", "type": F("D2", 2, logicalType="decimal", precision=4, scale=0)}]})

    def raw(vals):        # Decimal128(4, 0) from raw little-endian i128 values, in or out of precision
        buf = b"".join(int(v).to_bytes(16, "little", signed=True) for v in vals)
        arr = pa.Array.from_buffers(pa.decimal128(4, 0), len(vals), [None, pa.py_buffer(buf)])
        return pa.RecordBatch.from_arrays([arr], names=["a"])

    ok = [0, 1, -1, 9999, -9999, 127, -129, 32767, -32768]           # the last two: beyond precision 4, but 2-byte numbers
    datums = [x for a in P.serialize_record_batch(raw(ok), js, 2) for x in a.to_pylist()]
    assert datums[3] == b"\x27\x0f" and datums[6] == b"\xff\x7f" and datums[7] == b"\x7f\xff" and datums[8] == b"\x80\x00"
    assert [int(v) for v in P.deserialize_array(datums, js).column("a").to_pylist()] == ok
    for bad_row, bad in ((2, 32768), (5, -32769), (0, 10 ** 8)):
        vals = list(ok)
        vals[bad_row] = bad
        vals[6] = 70000                                      # a later offender: the lowest row is the one reported
        with pytest.raises(ValueError) as ei:
            P.serialize_record_batch(raw(vals), js, 2)
        assert str(ei.value) == "decimal value at row %d does not fit fixed(2)" % bad_row


# ---------------------------------------------------------------------------------------------------- duration
# Avro 1.11 "Duration": a fixed(12) of months, days, milliseconds (three little-endian u32).  The reference maps it to
# Duration(Millisecond) (schema_translate.rs:143), one i64 count of milliseconds: days x 86 400 000 + milliseconds is the
# only reading of that; a months component has no length in milliseconds, so a datum that carries one is a decode error.
DAY = 86_400_000
DUR = F("Dur", 12, logicalType="duration")
DUR_SCHEMA = json.dumps({"type": "record", "name": "D", "fields": [
    {"name": "d", "type": DUR},
    {"name": "nd", "type": ["null", F("Dur2", 12, logicalType="duration")]},
    {"name": "arr", "type": {"type": "array", "items": F("Dur3", 12, logicalType="duration")}},
    {"name": "un", "type": ["null", "string", F("Dur4", 12, logicalType="duration"), "long"]},
    {"name": "m", "type": {"type": "map", "values": [F("Dur5", 12, logicalType="duration"), "null"]}},
    {"name": "tail", "type": "string"},
]})


def _dur_rows(n):
    return [{"d": Dur(0, i, (i * 7919) % (1 << 32)), "nd": None if i % 3 == 0 else Dur(0, (i * 65537) % (1 << 32), 0xFFFFFFFF - i),
             "arr": [Dur(0, j, i) for j in range(i % 5)],
             "un": [None, f"s{i}", Branch(2, Dur(0, 0xFFFFFFFF, 0xFFFFFFFF)), i][i % 4],
             "m": [(f"k{j}", None if (i + j) % 2 else Dur(0, 1, j)) for j in range(i % 3)],
             "tail": f"t{i}"} for i in range(n)]


def _dur_records(n):
    sc = S.parse_schema(DUR_SCHEMA)
    return [to_datum(sc, r) for r in _dur_rows(n)]


def test_duration_translation_and_known_answers():
    js = '{"type":"record","name":"x","fields":[{"name":"a","type":%s}]}'
    avro = S.parse_schema(js % json.dumps(DUR))
    assert not S.is_supported(avro) and S.is_supported_extended(avro)
    assert P.arrow_schema(js % json.dumps(DUR)).field("a").type == pa.duration("ms") == S.to_arrow_schema(avro).field("a").type
    got = P.arrow_schema(DUR_SCHEMA)
    assert got.equals(S.to_arrow_schema(S.parse_schema(DUR_SCHEMA)), check_metadata=True)
    assert [c.name for c in got.field("un").type] == ["null", "varchar", "duration", "bigint"]      # schema_translate.rs:195
    # a duration is a fixed(12); on any other size the attribute is ignored and the fixed stays (as for uuid on fixed(8))
    odd = js % json.dumps(F("q", 11, logicalType="duration"))
    assert P.arrow_schema(odd).field("a").type == pa.binary(11) == S.to_arrow_schema(S.parse_schema(odd)).field("a").type
    recs = [bytes.fromhex("00000000" "01000000" "02000000"), bytes.fromhex("00000000" "ffffffff" "ffffffff"), bytes(12),
            bytes.fromhex("00000000" "00000000" "00005265")]                 # 0x65520000 ms = 1 700 003 840: more than a day, kept as is
    rows = py_walker.decode(recs, js % json.dumps(DUR), extended=True).column("a").cast(pa.int64()).to_pylist()
    assert rows == [DAY + 2, 0xFFFFFFFF * DAY + 0xFFFFFFFF, 0, 0x65520000]
    with pytest.raises(ValueError, match="duration with 3 months has no value in Duration\\(ms\\)"):
        py_walker.decode([bytes.fromhex("03000000" "01000000" "02000000")], js % json.dumps(DUR), extended=True)
    with pytest.raises(ValueError, match="unexpected end of buffer \\(fixed\\)"):
        py_walker.decode([bytes(11)], js % json.dumps(DUR), extended=True)
    # the encoder writes months = 0 and the days / milliseconds split the decoder's sum inverts
    rb = pa.RecordBatch.from_arrays([pa.array([DAY + 2, 0, DAY - 1, 0xFFFFFFFF * DAY + DAY - 1, 0xFFFFFFFF * DAY + 0xFFFFFFFF], pa.duration("ms"))], names=["a"])
    out = [x for a in py_encoder.serialize_record_batch(rb, js % json.dumps(DUR), 1, extended=True) for x in a.to_pylist()]
    assert out == [bytes.fromhex("00000000" "01000000" "02000000"), bytes(12), bytes.fromhex("00000000" "00000000" "ff5b2605"),
                   bytes.fromhex("00000000" "ffffffff" "ff5b2605"), bytes.fromhex("00000000" "ffffffff" "ffffffff")]
    for bad in (-1, 0xFFFFFFFF * DAY + (1 << 32)):
        with pytest.raises(ValueError, match="no Avro duration form"):
            py_encoder.serialize_record_batch(pa.RecordBatch.from_arrays([pa.array([0, bad], pa.duration("ms"))], names=["a"]),
                                              js % json.dumps(DUR), 1, extended=True)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(1, 1), (300, 3), (2500, 8)])
def test_gpu_duration_both_directions(n, k, kernel):
    recs = _dur_records(n)
    whole = py_walker.decode(recs, DUR_SCHEMA, extended=True)
    got = P.deserialize_array_threaded(recs, DUR_SCHEMA, k)
    sz = n // min(k, n)
    for i, g in enumerate(got):
        g.validate(full=True)
        assert_batches_identical(g, py_walker.decode(recs[i * sz: i * sz + g.num_rows], DUR_SCHEMA, extended=True))
    assert pa.Table.from_batches(got).combine_chunks().equals(pa.Table.from_batches([whole]))     # (to_pylist: timedelta cannot hold 2^32 days)
    enc = P.serialize_record_batch(whole, DUR_SCHEMA, k)
    for g, e in zip(enc, py_encoder.serialize_record_batch(whole, DUR_SCHEMA, k, extended=True)):
        assert g.equals(e)
    datums = [x for a in enc for x in a.to_pylist()]
    assert_batches_identical(P.deserialize_array(datums, DUR_SCHEMA), whole)
    # days and milliseconds are folded into one count: a datum with more than a day of milliseconds comes back normalised
    assert sum(a == b for a, b in zip(datums, recs)) < n or n == 1


@pytest.mark.gpu
def test_gpu_duration_errors(kernel):
    js = json.dumps({"type": "record", "name": "x", "fields": [{"name": "s", "type": "string"}, {"name": "a", "type": ["null", DUR]}]})
    good = b"\x02g" + b"\x02" + bytes.fromhex("00000000" "01000000" "02000000")
    assert P.deserialize_array([good] * 3, js).column("a").cast(pa.int64()).to_pylist() == [DAY + 2] * 3
    for bad, msg in ((b"\x02g" + b"\x02" + bytes.fromhex("07000000" "01000000" "02000000"), "duration with 7 months has no value in Duration(ms)"),
                     (b"\x02g" + b"\x02" + bytes.fromhex("00000001" "01000000" "02000000"), "duration with 16777216 months has no value in Duration(ms)"),
                     (b"\x02g" + b"\x02" + bytes(11), "unexpected end of buffer (fixed)")):
        for recs in ([bad], [good] * 300 + [bad] + [good] * 5):
            with pytest.raises(ValueError) as ei:
                P.deserialize_array_threaded(recs, js, 2)
            assert str(ei.value) == msg
            with pytest.raises(ValueError) as eo:
                py_walker.decode(recs, js, extended=True)
            assert str(eo.value) == msg
    one = '{"type":"record","name":"x","fields":[{"name":"a","type":%s}]}' % json.dumps(DUR)
    for bad_row, bad in ((1, -1), (0, 0xFFFFFFFF * DAY + (1 << 32)), (2, -(1 << 62))):
        vals = [5, 6, 7, -9]
        vals[bad_row] = bad
        with pytest.raises(ValueError) as ei:
            P.serialize_record_batch(pa.RecordBatch.from_arrays([pa.array(vals, pa.duration("ms"))], names=["a"]), one, 2)
        assert str(ei.value) == "duration value at row %d has no Avro duration form (negative, or beyond 2^32-1 days + 2^32-1 ms)" % bad_row
    with pytest.raises(ValueError, match="arrow array downcast failed"):
        P.serialize_record_batch(pa.RecordBatch.from_arrays([pa.array([1], pa.duration("us"))], names=["a"]), one, 1)
