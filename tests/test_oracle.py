"""Oracle pinned against the reference's golden vectors and differential tests (CPU only)."""
import json
import os

import numpy as np
import pyarrow as pa
import pytest

import cases
from arrow_compare import assert_batches_identical
from avrogen import fastgen, synth
from avrogen.schemas import SCHEMAS
from oracle import avro_schema as S
from oracle import c_walker, py_walker

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")


def _norm(v):
    if isinstance(v, list):
        return [_norm(x) for x in v]
    if isinstance(v, tuple):
        return [_norm(x) for x in v]
    if isinstance(v, dict):
        return {k: _norm(x) for k, x in v.items()}
    return v


@pytest.mark.parametrize("walker", [py_walker, c_walker])
def test_reference_golden_vectors(walker):
    g = json.load(open(GOLDEN))
    for v in g["vectors"]:
        schema = json.dumps(g["schemas"][v["schema"]])
        rb = walker.decode([bytes.fromhex(v["hex"])], schema)
        rb.validate(full=True)
        row = rb.to_pylist()[0]
        exp = dict(v["expected"])
        if "status_type_id" in exp:
            assert rb.column("status").type_codes[0].as_py() == exp.pop("status_type_id")
        if "registrationDate_ms" in exp:
            assert rb.column("registrationDate").cast("int64")[0].as_py() == exp.pop("registrationDate_ms")
            row.pop("registrationDate")
        assert _norm(row) == exp, v["name"]


def test_golden_shapes_like_reference_tests():
    # deserialize.rs:246-249: 4 copies -> 8 columns x 4 rows; deserialize.rs:305-308: 1 row x 5 columns
    g = json.load(open(GOLDEN))
    by = {v["name"]: v for v in g["vectors"]}
    v = by["deserialize_rs_244"]
    rb = c_walker.decode([bytes.fromhex(v["hex"])] * 4, json.dumps(g["schemas"][v["schema"]]))
    assert (rb.num_columns, rb.num_rows) == (8, 4)
    v = by["deserialize_rs_303"]
    rb = c_walker.decode([bytes.fromhex(v["hex"])], json.dumps(g["schemas"][v["schema"]]))
    assert (rb.num_columns, rb.num_rows) == (5, 1)
    # lib.rs:171-172: the three User datums decode together
    recs = [bytes.fromhex(by[n]["hex"]) for n in ("lib_rs_165_avro_datum", "lib_rs_166_avro_datum2", "lib_rs_167_avro_datum3")]
    rb = c_walker.decode(recs, SCHEMAS["kat_user"])
    assert rb.num_rows == 3 and rb.column("status").type_codes.to_pylist() == [1, 1, 0]


@pytest.mark.parametrize("case", cases.differential_cases(), ids=lambda c: c[0])
def test_differential_restatements(case):
    _, schema, recs, expected = case
    a = py_walker.decode(recs, schema)
    b = c_walker.decode(recs, schema)
    a.validate(full=True)
    assert_batches_identical(a, b)
    assert _norm(a.to_pylist()) == _norm(expected)


def test_logical_types():
    schema, recs, vals = cases.logical_case()
    rb = c_walker.decode(recs, schema)
    assert rb.schema.field("d").type == pa.date32()
    assert rb.schema.field("tm").type == pa.timestamp("ms") and rb.schema.field("tu").type == pa.timestamp("us")
    assert rb.column("d").cast("int32").to_pylist() == [v["d"] for v in vals]
    assert rb.column("tm").cast("int64").to_pylist() == [v["tm"] for v in vals]
    assert rb.column("tu").cast("int64").to_pylist() == [v["tu"] for v in vals]
    assert_batches_identical(rb, py_walker.decode(recs, schema))


@pytest.mark.parametrize("case", cases.wire_cases() + cases.nesting_cases(), ids=lambda c: c[0])
def test_walkers_agree_on_wire_and_nesting(case):
    _, schema, recs = case
    a = py_walker.decode(recs, schema)
    a.validate(full=True)
    assert_batches_identical(a, c_walker.decode(recs, schema))


def test_wire_forms_values():
    by = {c[0]: c for c in cases.wire_cases()}
    _, schema, recs = by["array_blocks"]
    rows = c_walker.decode(recs, schema).to_pylist()
    assert rows[0]["tags"] == ["a", "bb", "ccc", "d"]
    assert rows[1]["tags"] == ["x"] * 70 + ["y"] * 3 and rows[2]["tags"] == [] and rows[3]["tags"] == ["solo"]
    _, schema, recs = by["map_blocks"]
    rows = c_walker.decode(recs, schema).to_pylist()
    assert rows[0]["props"] == [("k1", "v1"), ("k2", "v2"), ("k1", "dup")]   # wire order, duplicates kept
    _, schema, recs = by["extremes"]
    rb = c_walker.decode(recs, schema)
    assert rb.column("i").to_pylist()[:3] == [-(2**31) + 5, 2**31 - 7, 3]     # `as i32` truncation
    assert rb.column("l").to_pylist()[:3] == [2**63 - 1, -(2**63), 0]
    f = np.frombuffer(rb.column("f").buffers()[1], dtype=np.uint32, count=3)
    assert list(f) == [0x7FC00001, 0xFF800000, 1]                            # NaN payloads survive the bit copy


@pytest.mark.parametrize("case", cases.error_cases(), ids=lambda c: c[0])
def test_error_messages(case):
    _, schema, good, bad, msg = case
    for walker, exc in ((py_walker, py_walker.DecodeError), (c_walker, ValueError)):
        with pytest.raises(exc) as ei:
            walker.decode(good + [bad] + good, schema)
        assert str(ei.value) == msg
    # first failing record in list order wins across chunks (deserialize.rs:115-119)
    with pytest.raises(ValueError) as ei:
        c_walker.decode_threaded(good + [bad] + good + [b"\x80" * 11], schema, 3)
    assert str(ei.value) == msg


def test_canonical_form():
    # leaf validity only when a null was appended; nullable struct always; non-nullable never
    recs = synth.records("full", 64)
    rb = c_walker.decode(recs, SCHEMAS["full"])
    assert rb.column("created_at").buffers()[0] is None and rb.column("class").buffers()[0] is None
    assert rb.column("address").buffers()[0] is not None
    assert rb.column("emails").buffers()[0] is None and rb.column("emails").values.buffers()[0] is None
    s = parse = SCHEMAS["t_nullable_nested"]
    all_present = cases._enc(s, [{"inner": {"x": 1}}, {"inner": {"x": 2}}])
    rb = c_walker.decode(all_present, s)
    assert rb.column("inner").buffers()[0] is not None and rb.column("inner").null_count == 0   # fast_decode.rs:629
    assert rb.column("inner").field(0).buffers()[0] is None                                     # lazy leaf null buffer
    # sparse union children are null-filled and full length; type_id 0 under a null parent
    rb = c_walker.decode(synth.records("full", 50), SCHEMAS["full"])
    st = rb.column("status")
    assert all(len(st.field(i)) == 50 for i in range(4))
    tids = st.type_codes.to_pylist()
    for i, t in enumerate(tids):
        for j in (1, 2, 3):
            assert st.field(j)[i].is_valid == (t == j)


def test_chunk_boundaries():
    recs = synth.records("cfg3", 103)
    for k, want in ((1, [103]), (8, [12] * 7 + [19]), (0, [103]), (103, [1] * 103), (500, [1] * 103), (2, [51, 52])):
        out = c_walker.decode_threaded(recs, SCHEMAS["cfg3"], k)
        assert [b.num_rows for b in out] == want                       # deserialize.rs:53-68
        whole = c_walker.decode(recs, SCHEMAS["cfg3"])
        assert pa.Table.from_batches(out).equals(pa.Table.from_batches([whole]))
    out = c_walker.decode_threaded([], SCHEMAS["cfg3"], 4)
    assert [b.num_rows for b in out] == [0]


@pytest.mark.parametrize("name", sorted(synth.GENERATORS))
def test_walkers_agree_on_generated(name):
    recs = synth.records(name, 200, seed=11)
    a = py_walker.decode(recs, SCHEMAS[name])
    b = c_walker.decode(recs, SCHEMAS[name])
    assert_batches_identical(a, b)
    exp = [synth.GENERATORS[name](11, i) for i in range(200)]
    if name.startswith("full_realistic"):                              # (created_at is a timestamp-micros there: compare the raw integers)
        a = a.set_column(a.schema.get_field_index("created_at"), "created_at", a.column("created_at").cast(pa.int64()))
    assert _norm(a.to_pylist()) == _norm(exp)                          # end-to-end known answer


@pytest.mark.parametrize("seed", range(0, 40, 4))
def test_walkers_agree_on_random_schemas(seed):
    """The two oracle walkers (pure Python / C) are independent restatements: they must agree on random
    schemas x random records too (tests/random_cases.py), incl. multi-block and negative-count containers."""
    import random_cases
    js, recs = random_cases.random_case(seed, 60)
    a = py_walker.decode(recs, js)
    b = c_walker.decode(recs, js)
    assert_batches_identical(a, b)
    a.validate(full=True)


def test_schema_translation_details():
    s = S.parse_schema(SCHEMAS["full"])
    sch = S.to_arrow_schema(s)
    assert sch.field("status").type.field(0).name == "null" and sch.field("status").type.field(1).name == "varchar"
    assert sch.field("status").type.field(2).name == "int" and sch.field("status").type.field(3).name == "bit"
    assert sch.field("status").nullable and not sch.field("created_at").nullable
    assert sch.field("address").type.field("street").nullable          # children of a nullable record inherit
    assert not sch.field("emails").nullable and sch.field("emails").type.value_field.nullable
    assert sch.field("class").type == pa.string()
    # doc / aliases metadata (schema_translate.rs:222-266) and enum-fullname child names (127-130)
    js = json.dumps({"type": "record", "name": "T", "namespace": "a.b", "fields": [
        {"name": "r", "type": {"type": "record", "name": "R", "doc": "rdoc", "aliases": ["Old", "x.Y"], "fields": [
            {"name": "f", "type": "int", "doc": "fdoc"}]}},
        {"name": "u", "type": ["int", {"type": "enum", "name": "E", "symbols": ["s"], "doc": "ignored"}]}]})
    sch = S.to_arrow_schema(S.parse_schema(js))
    assert sch.field("r").metadata == {b"avro::doc": b"rdoc", b"avro::aliases": b"[a.b.Old,x.Y]"}
    assert sch.field("r").type.field("f").metadata == {b"avro::doc": b"fdoc"}
    assert [sch.field("u").type.field(i).name for i in range(2)] == ["int", "a.b.E"]
    for bad in ('{"type":"record","name":"x","fields":[{"name":"a","type":"bytes"}]}', '"string"',
                '{"type":"record","name":"x","fields":[{"name":"a","type":["null",{"type":"map","values":"int"}]}]}',
                '{"type":"record","name":"x","fields":[{"name":"a","type":{"type":"fixed","name":"f","size":4}}]}',
                '{"type":"record","name":"x","fields":[{"name":"a","type":{"type":"int","logicalType":"time-millis"}}]}'):
        with pytest.raises(S.SchemaError):
            S.build_tree(S.parse_schema(bad))


def test_fastgen_matches_python_spec():
    for name in ("full", "flat4", "cfg3"):
        d, o = fastgen.generate(name, 3000, seed=5, start=17, nthreads=3)
        assert fastgen.split(d, o) == synth.records(name, 3000, seed=5, start=17)
        d1, o1 = fastgen.generate(name, 3000, seed=5, start=17, nthreads=1)
        assert np.array_equal(d, d1) and np.array_equal(o, o1)
    # round 6: the workloads off the friendly distribution (long varints, 8 KiB strings, > 8,191-item arrays, record-size skew, wide records)
    for name, n in (("full_realistic", 1500), ("full_realistic_nogiant", 1500), ("full_realistic_heavy", 120), ("full_skewed", 1500), ("wide97", 200), ("wide200", 120), ("wide400", 60)):
        d, o = fastgen.generate(name, n, seed=5, start=17, nthreads=3)
        assert fastgen.split(d, o) == synth.records(name, n, seed=5, start=17), name
