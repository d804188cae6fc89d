"""Named-type references (`{"name": "work", "type": "Addr"}`): the reference stops at them (schema_translate.rs:51 is
`todo!("Add support for AvroSchema::Ref")`, and fast_decode.rs:59 gates them out), so any schema that uses a record, enum
or fixed type twice cannot be decoded by it at all.  Here a reference IS the type it names (Avro 1.11 specification, "Names"):
the front-end substitutes a copy of the definition, so the Arrow translation, both GPU directions and the error behaviour
are those of the same schema written out in full -- which is what the tests check (no reference behaviour exists to pin
against: beyond-reference, specification-pinned, like the N4 types)."""
import json

import pyarrow as pa
import pytest

from arrow_compare import assert_batches_identical
from avrogen.encoder import Branch, to_datum
from oracle import avro_schema as S
from oracle import py_encoder, py_walker

import pyruhvro_amd as P

ADDR = {"type": "record", "name": "Addr", "fields": [{"name": "street", "type": "string"}, {"name": "zip", "type": ["null", "int"]}]}
COLOR = {"type": "enum", "name": "Color", "symbols": ["RED", "GREEN", "B"]}
WITH_REFS = json.dumps({"type": "record", "name": "Top", "namespace": "a.b", "fields": [
    {"name": "home", "type": ADDR}, {"name": "work", "type": ["null", "Addr"]},
    {"name": "others", "type": {"type": "array", "items": "a.b.Addr"}},
    {"name": "color", "type": COLOR}, {"name": "colors", "type": {"type": "map", "values": "Color"}},
    {"name": "pick", "type": ["null", "Color", "string"]}, {"name": "id", "type": "long"}]})


def _inline(name):
    """The same schema with every reference written out (distinct names: a full name may be defined only once)."""
    a = dict(ADDR, name=name("Addr"))
    c = dict(COLOR, name=name("Color"))
    return a, c


WRITTEN_OUT = json.dumps({"type": "record", "name": "Top", "namespace": "a.b", "fields": [
    {"name": "home", "type": ADDR}, {"name": "work", "type": ["null", dict(ADDR, name="Addr2")]},
    {"name": "others", "type": {"type": "array", "items": dict(ADDR, name="Addr3")}},
    {"name": "color", "type": COLOR}, {"name": "colors", "type": {"type": "map", "values": dict(COLOR, name="Color2")}},
    {"name": "pick", "type": ["null", dict(COLOR, name="Color3"), "string"]}, {"name": "id", "type": "long"}]})


def _rows(n):
    out = []
    for i in range(n):
        addr = lambda j: {"street": f"{i}-{j} Main St" * (1 + j % 3), "zip": None if (i + j) % 3 == 0 else 10000 + i + j}  # noqa: E731
        out.append({"home": addr(0), "work": None if i % 4 == 0 else addr(1), "others": [addr(2 + j) for j in range(i % 5)],
                    "color": ["RED", "GREEN", "B"][i % 3], "colors": [(f"k{j}", ["B", "RED"][j % 2]) for j in range(i % 3)],
                    "pick": [None, "GREEN", Branch(2, f"s{i}")][i % 3], "id": i * 7})
    return out


def test_reference_stops_at_refs_and_the_engine_resolves_them():
    strict = S.parse_schema(WITH_REFS)
    assert not S.is_supported(strict)                                     # fast_decode.rs:59
    with pytest.raises(ValueError):
        S.to_arrow_schema(strict)                                         # schema_translate.rs:51 todo!()
    got = P.arrow_schema(WITH_REFS)
    assert got.equals(S.to_arrow_schema(S.parse_schema(WITH_REFS, resolve_refs=True)), check_metadata=True)
    # ... and it is the translation of the schema written out in full (the enum children of `pick` are named by their
    # full names, which differ: compare types column by column, names of named union children aside)
    full = P.arrow_schema(WRITTEN_OUT)
    for a, b in zip(got, full):
        if a.name != "pick":
            assert a.equals(b), (a, b)
    assert [c.name for c in got.field("pick").type] == ["null", "a.b.Color", "varchar"]


def test_recursive_and_unknown_names_are_schema_errors():
    rec = json.dumps({"type": "record", "name": "Node", "fields": [{"name": "v", "type": "int"}, {"name": "next", "type": ["null", "Node"]}]})
    with pytest.raises(ValueError, match="recursive named type Node"):
        P.arrow_schema(rec)
    with pytest.raises(ValueError, match="recursive named type Node"):
        S.parse_schema(rec, resolve_refs=True)
    with pytest.raises(ValueError, match="Unknown type: Nope"):
        P.arrow_schema('{"type":"record","name":"r","fields":[{"name":"a","type":"Nope"}]}')
    dup = json.dumps({"type": "record", "name": "r", "fields": [{"name": "a", "type": ADDR}, {"name": "u", "type": ["Addr", "Addr"]}]})
    with pytest.raises(ValueError, match="duplicate"):
        P.arrow_schema(dup)


def test_oracle_round_trip_with_refs():
    tree = S.parse_schema(WITH_REFS, resolve_refs=True)
    recs = [to_datum(tree, r) for r in _rows(40)]
    rb = py_walker.decode(recs, WITH_REFS, extended=True)
    assert rb.num_rows == 40 and rb.column("others").to_pylist()[4][0]["street"].startswith("4-2 Main St")
    # the datums are those of the written-out schema: same wire format, same values
    rb2 = py_walker.decode(recs, WRITTEN_OUT, extended=True)
    for name in ("home", "work", "others", "color", "colors", "id"):
        assert rb.column(name).to_pylist() == rb2.column(name).to_pylist()
    enc = [x for a in py_encoder.serialize_record_batch(rb, WITH_REFS, 3, extended=True) for x in a.to_pylist()]
    assert enc == recs


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["generic", "specialized"])
@pytest.mark.parametrize("n,k", [(1, 1), (300, 3), (5000, 8)])
def test_gpu_decode_and_encode_with_refs(n, k, mode):
    old = P.set_kernel_mode(mode)
    try:
        tree = S.parse_schema(WITH_REFS, resolve_refs=True)
        recs = [to_datum(tree, r) for r in _rows(n)]
        got = P.deserialize_array_threaded(recs, WITH_REFS, k)
        off = 0
        for g in got:
            g.validate(full=True)
            assert_batches_identical(g, py_walker.decode(recs[off: off + g.num_rows], WITH_REFS, extended=True))
            off += g.num_rows
        assert off == n
        whole = pa.Table.from_batches(got).combine_chunks().to_batches()[0] if n > 1 else got[0]
        back = [x for a in P.serialize_record_batch(whole, WITH_REFS, k) for x in a.to_pylist()]
        assert back == recs
        with pytest.raises(ValueError) as ei:
            P.deserialize_array_threaded(recs[:10] + [recs[0][:3]], WITH_REFS, 2)
        with pytest.raises(ValueError) as eo:
            py_walker.decode(recs[:10] + [recs[0][:3]], WITH_REFS, extended=True)
        assert str(ei.value) == str(eo.value)
    finally:
        P.set_kernel_mode(old)
