"""The encode oracle (oracle/py_encoder.py: restatement of ruhvro/src/fast_encode.rs + serialize.rs chunking),
pinned against the decode oracle, the test encoder and the reference's golden datums.  CPU only; the GPU path it
checks (rh_encode) is compared with it in tests/test_gpu_encode.py."""
import json
import os

import pyarrow as pa
import pytest

import cases
from arrow_compare import assert_batches_identical
from avrogen import synth
from avrogen.schemas import SCHEMAS
from oracle import c_walker, py_encoder, py_walker

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")


def _datums(arrays):
    return [b for a in arrays for b in a.to_pylist()]


@pytest.mark.parametrize("name", sorted(synth.GENERATORS))
def test_encode_inverts_decode_on_generated_records(name):
    """The test encoder writes the same single-block form the reference writes (fast_encode.rs:525-561),
    so encode(decode(records)) must give the records back byte for byte, and decode(encode(x)) == x."""
    recs = synth.records(name, 257, seed=4)
    batch = c_walker.decode(recs, SCHEMAS[name])
    out = py_encoder.serialize_record_batch(batch, SCHEMAS[name], 4)
    assert [len(a) for a in out] == [64, 64, 64, 65]                      # serialize.rs:19-30
    assert _datums(out) == recs
    assert_batches_identical(c_walker.decode(_datums(out), SCHEMAS[name]), batch)


def test_golden_datums_reencode_to_their_consumed_bytes():
    g = json.load(open(GOLDEN))
    for v in g["vectors"]:
        schema = json.dumps(g["schemas"][v["schema"]])
        rec = bytes.fromhex(v["hex"])
        batch = py_walker.decode([rec], schema)
        out = py_encoder.serialize_record_batch(batch, schema, 1)
        assert _datums(out) == [rec[: v.get("consumed", len(rec))]], v["name"]


@pytest.mark.parametrize("case", cases.nesting_cases() + cases.differential_cases(), ids=lambda c: c[0])
def test_round_trip_on_nested_and_differential_schemas(case):
    schema, recs = case[1], case[2]
    batch = c_walker.decode(recs, schema)
    out = _datums(py_encoder.serialize_record_batch(batch, schema, 3))
    assert_batches_identical(c_walker.decode(out, schema), batch)
    # multi-block / negative-count inputs re-encode to ONE block per container: shorter or equal, never different content
    assert all(len(a) <= len(b) for a, b in zip(out, recs))


def test_columns_are_matched_by_name_and_missing_ones_are_named():
    recs = synth.records("cfg3", 10)
    batch = c_walker.decode(recs, SCHEMAS["cfg3"])
    shuffled = pa.RecordBatch.from_arrays([batch.column(i) for i in (4, 2, 0, 3, 1)],
                                          names=[batch.schema.names[i] for i in (4, 2, 0, 3, 1)])
    assert _datums(py_encoder.serialize_record_batch(shuffled, SCHEMAS["cfg3"], 1)) == recs   # fast_encode.rs:155-181
    missing = batch.drop_columns(["age"])
    with pytest.raises(ValueError) as ei:
        py_encoder.serialize_record_batch(missing, SCHEMAS["cfg3"], 1)
    assert str(ei.value) == ("Arrow struct missing column 'age' required by Avro schema. "
                             'Available columns: ["id", "name", "s", "class"]')


def test_encode_errors():
    s = SCHEMAS["t_enum"]
    bad = pa.RecordBatch.from_arrays([pa.array(["A", "Z"])], names=["e"])
    with pytest.raises(ValueError) as ei:
        py_encoder.serialize_record_batch(bad, s, 1)
    assert str(ei.value) == "fast_encode: enum symbol 'Z' not in schema"        # fast_encode.rs:575-577
    su = SCHEMAS["t_union"]
    batch = c_walker.decode(cases._enc(su, [{"u": None}, {"u": "x"}]), su)
    u = batch.column(0)
    broken = pa.UnionArray.from_sparse(pa.array([0, 9], type=pa.int8()), [u.field(i) for i in range(u.type.num_fields)],
                                       field_names=[u.type.field(i).name for i in range(u.type.num_fields)])
    with pytest.raises(ValueError) as ei:
        py_encoder.serialize_record_batch(pa.RecordBatch.from_arrays([broken], names=["u"]), su, 1)
    assert str(ei.value) == "fast_encode: union type_id 9 out of range"          # fast_encode.rs:540-542


def test_chunking_and_empty_input():
    recs = synth.records("flat4", 7)
    batch = c_walker.decode(recs, SCHEMAS["flat4"])
    for k, want in ((1, [7]), (3, [2, 2, 3]), (0, [7]), (50, [1] * 7)):
        assert [len(a) for a in py_encoder.serialize_record_batch(batch, SCHEMAS["flat4"], k)] == want
    empty = c_walker.decode([], SCHEMAS["flat4"])
    assert [len(a) for a in py_encoder.serialize_record_batch(empty, SCHEMAS["flat4"], 4)] == [0]
