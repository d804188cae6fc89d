"""Host-side logic of the engine, no GPU needed: schema front-end vs the oracle, the C ABI surface,
the Python boundary's argument handling, and the loud failure without a device."""
import ctypes as C
import json
import os
import re

import pyarrow as pa
import pytest
from hypothesis import given, settings, strategies as st

import cases
from avrogen.schemas import SCHEMAS
from oracle import avro_schema as S

import pyruhvro_amd as P
from pyruhvro_amd import cabi
from conftest import ROOT, has_gpu


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ruhvro_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(rh_[a-z_]+)\s*\(", hdr))
    assert {"rh_schema_compile", "rh_decode", "rh_decode_packed", "rh_decode_device", "rh_schema_export"} <= names
    lib = C.CDLL(cabi.LIB_PATH)
    for n in sorted(names):
        assert hasattr(lib, n), f"libruhvro_hip.so does not export {n}"
    assert cabi.lib().rh_abi_version() == 7


def test_gather_pool_hands_out_every_task_exactly_once():
    """The host thread pool behind the gather of pipelined calls (engine_host.cpp CallPool: lock-free phase hand-over, polling
    workers that fall asleep between calls): many short phases, few and many workers, more workers than cores."""
    f = C.CDLL(cabi.LIB_PATH).rh_selftest_pool
    f.restype = C.c_uint32
    f.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    for workers, phases, max_tasks in ((1, 2000, 7), (3, 20000, 5), (8, 20000, 64), (32, 6000, 33), (64, 1500, 200)):
        assert f(workers, phases, max_tasks) == 0, (workers, phases, max_tasks)


def test_clamp_chunks_matches_reference():   # deserialize.rs:53-55
    f = cabi.lib().rh_clamp_chunks
    assert [f(10, k) for k in (0, 1, 3, 10, 11, 1000)] == [1, 1, 3, 10, 10, 10]
    assert f(0, 0) == 1 and f(0, 8) == 1 and f(1, 8) == 1


@pytest.mark.parametrize("name", sorted(SCHEMAS))
def test_arrow_schema_matches_oracle(name):
    got = P.arrow_schema(SCHEMAS[name])
    exp = S.to_arrow_schema(S.parse_schema(SCHEMAS[name]))
    assert got.equals(exp, check_metadata=True), f"{got}\n!=\n{exp}"


def test_arrow_schema_nesting_cases_and_metadata():
    for _, schema, _ in cases.nesting_cases():
        assert P.arrow_schema(schema).equals(S.to_arrow_schema(S.parse_schema(schema)), check_metadata=True)
    js = json.dumps({"type": "record", "name": "T", "namespace": "a.b", "fields": [
        {"name": "r", "type": {"type": "record", "name": "R", "doc": "rdoc", "aliases": ["Old", "x.Y"], "fields": [
            {"name": "f", "type": "int", "doc": "fdoc"}, {"name": "g", "type": ["null", "R2x"] if False else "long"}]}},
        {"name": "u", "type": ["int", {"type": "enum", "name": "E", "symbols": ["s"], "doc": "ignored"}]},
        {"name": "e", "type": {"type": "enum", "name": "c.d.E2", "symbols": ["s"], "doc": "dropped"}}]})
    got = P.arrow_schema(js)
    assert got.equals(S.to_arrow_schema(S.parse_schema(js)), check_metadata=True)
    assert got.field("r").metadata == {b"avro::doc": b"rdoc", b"avro::aliases": b"[a.b.Old,x.Y]"}
    assert got.field("e").metadata is None


BAD_SCHEMAS = [
    ("not json", "Failed to parse schema"),
    ('"string"', "record"),
    ('{"type":"record","name":"x","fields":[{"name":"a","type":{"type":"long","logicalType":"local-timestamp-millis"}}]}', "local-timestamp-millis"),
    ('{"type":"record","name":"x","fields":[{"name":"a","type":{"type":"long","logicalType":"timestamp-nanos"}}]}', "timestamp-nanos"),
    ('{"type":"record","name":"x","fields":[{"name":"a","type":{"type":"fixed","name":"f"}}]}', "size"),
    ('{"type":"record","name":"x","fields":[{"name":"a","type":"nosuchtype"}]}', "Unknown type"),
    ('{"type":"record","name":"x","fields":[{"name":"a","type":{"type":"record","name":"y","fields":[]}},{"name":"b","type":"y"}]}', ""),
    ('{"type":"record","name":"x","fields":[{"name":"a","type":["null",{"type":"map","values":"int"}]}]}', "Map support"),
    ('{"type":"record","name":"x","fields":[{"name":"a","type":["int","int"]}]}', "duplicate"),
    ('{"type":"record","name":"x","fields":[]}', "0 fields"),
]


@pytest.mark.parametrize("schema,frag", BAD_SCHEMAS)
def test_bad_schema_is_value_error(schema, frag):
    with pytest.raises(ValueError) as ei:          # src/lib.rs:25-27,52: parse failure -> ValueError
        P.arrow_schema(schema)
    assert frag.lower() in str(ei.value).lower()
    with pytest.raises(ValueError):
        S.build_tree(S.parse_schema(schema))       # the oracle rejects the same documents


# ---- random schemas: product's translation == oracle's -------------------------------------------
_prims = st.sampled_from(["int", "long", "float", "double", "boolean", "string",
                          {"type": "int", "logicalType": "date"}, {"type": "long", "logicalType": "timestamp-millis"},
                          {"type": "long", "logicalType": "timestamp-micros"}])


def _types(depth, counter):
    def named(kind):
        counter[0] += 1
        return f"{kind}{counter[0]}"
    if depth <= 0:
        return _prims
    sub = st.deferred(lambda: _types(depth - 1, counter))
    rec = st.lists(sub, min_size=1, max_size=3).map(
        lambda ts: {"type": "record", "name": named("R"), "fields": [{"name": f"f{i}", "type": t} for i, t in enumerate(ts)]})
    enum = st.lists(st.sampled_from(["A", "B", "CC", "DDD"]), min_size=1, max_size=3, unique=True).map(
        lambda sy: {"type": "enum", "name": named("E"), "symbols": sy})
    arr = sub.map(lambda t: {"type": "array", "items": t})
    mp = sub.map(lambda t: {"type": "map", "values": t})
    return st.one_of(_prims, rec, enum, arr, mp)


@st.composite
def schemas(draw):
    counter = [0]
    base = _types(2, counter)

    def wrap(t):
        mode = draw(st.integers(0, 3))
        is_map = isinstance(t, dict) and t.get("type") == "map"
        if mode == 1 and not is_map:
            return ["null", t]
        if mode == 2 and not is_map:
            return [t, "null"]
        if mode == 3 and not is_map:
            return ["null", t, "boolean"] if t not in ("boolean",) else ["null", t, "int"]
        return t
    n = draw(st.integers(1, 4))
    fields = [{"name": f"c{i}", "type": wrap(draw(base))} for i in range(n)]
    return json.dumps({"type": "record", "name": "Top", "namespace": draw(st.sampled_from(["", "ns", "a.b"])), "fields": fields})


@settings(max_examples=150, deadline=None)
@given(schemas())
def test_random_schema_translation(js):
    try:
        exp = S.to_arrow_schema(S.parse_schema(js))
        S.build_tree(S.parse_schema(js))
    except S.SchemaError:
        with pytest.raises(ValueError):
            P.arrow_schema(js)
        return
    got = P.arrow_schema(js)
    assert got.equals(exp, check_metadata=True), f"{js}\n{got}\n!=\n{exp}"


# ---- Python boundary -----------------------------------------------------------------------------
def test_argument_errors_without_touching_the_gpu():
    with pytest.raises(TypeError):
        P.deserialize_array_threaded("notalist", SCHEMAS["flat4"], 2)
    with pytest.raises(TypeError):
        P.deserialize_array_threaded([b"\x00", "str"], SCHEMAS["flat4"], 2)       # non-bytes element
    with pytest.raises(TypeError):
        P.deserialize_array_threaded([b"\x00"], 123, 2)
    with pytest.raises(ValueError):
        P.deserialize_array_threaded([b"\x00"], "{", 2)
    with pytest.raises(TypeError):
        P.serialize_record_batch(None, SCHEMAS["flat4"], 1)
    assert set(["deserialize_array", "deserialize_array_threaded", "deserialize_array_threaded_spawn",
                "serialize_record_batch", "serialize_record_batch_spawn"]) <= set(dir(P))   # src/lib.rs:150-158


@pytest.mark.skipif(has_gpu(), reason="checks the no-device failure mode")
def test_binary_array_argument_errors():
    import pyarrow as pa
    with pytest.raises(TypeError):
        P.deserialize_binary_array([b"x"], SCHEMAS["flat4"], 1)
    with pytest.raises(ValueError):
        P.deserialize_binary_array(pa.array([b"x", None], type=pa.binary()), SCHEMAS["flat4"], 1)
    with pytest.raises(ValueError):
        P.deserialize_binary_array(pa.array([b"x"], type=pa.binary()), '{"type":"string"}', 1)


def test_no_device_fails_loudly_no_cpu_fallback():
    with pytest.raises(RuntimeError) as ei:
        P.deserialize_array([b"\x00\x00" + b"\x00" * 8 + b"\x00"], SCHEMAS["flat4"])
    assert "no HIP device" in str(ei.value)


def test_list_extraction_holds_and_returns_every_reference():
    """The CPython boundary takes one reference per record while the GIL is released and recovers the object from its
    payload pointer afterwards: after a call that fails (no device here, or a bad element half way) every refcount is
    back where it was, and a non-bytes element is named by index like PyO3's extraction error."""
    import sys
    recs = [bytes([i % 251]) * (1 + i % 40) for i in range(5000)]
    probe = [recs[0], recs[2500], recs[-1]]
    before = [sys.getrefcount(o) for o in probe]
    with pytest.raises(RuntimeError):
        P.deserialize_array_threaded(recs, SCHEMAS["cfg3"], 4)
    assert [sys.getrefcount(o) for o in probe] == before
    bad = list(recs)
    bad[3000] = 5
    before = [sys.getrefcount(o) for o in probe]     # (the second list holds a reference too)
    with pytest.raises(TypeError) as ei:
        P.deserialize_array_threaded(bad, SCHEMAS["cfg3"], 4)
    assert "list element 3000: expected bytes, got int" in str(ei.value)
    assert [sys.getrefcount(o) for o in probe] == before
    mixed = [bytearray(b"ab"), b"cd"] * 10           # bytearray elements are copied into bytes the call owns
    with pytest.raises(RuntimeError):
        P.deserialize_array_threaded(mixed, SCHEMAS["cfg3"], 2)
    with pytest.raises(TypeError):
        P.deserialize_array_threaded((b"x",), SCHEMAS["cfg3"], 1)


def test_product_does_not_reference_the_oracle():
    """Nothing under pyruhvro_amd/ (nor the drop-in alias pyruhvro/) may import, link or execute the oracle -- directly
    or through a module that does: every import of every product module is followed, and the oracle, the test
    suite and its case tables must not be reachable.  Native sources are searched for the oracle's symbols."""
    import ast
    banned = {"oracle", "tests", "cases", "random_cases", "conftest", "arrow_compare", "known_schemas", "hipmem"}
    pkgs = [os.path.join(ROOT, "pyruhvro_amd"), os.path.join(ROOT, "pyruhvro")]
    seen, todo = set(), []
    for pkg in pkgs:
        for dirpath, _, files in os.walk(pkg):
            for f in files:
                path = os.path.join(dirpath, f)
                if f.endswith(".py"):
                    todo.append(path)
                elif f.endswith((".cpp", ".hip", ".h", ".hpp", ".inc")):
                    txt = open(path).read()
                    assert "oracle_walk" not in txt and "orc_decode" not in txt and "liboracle" not in txt, path

    def resolve(mod):          # a module of THIS repository -> its file (third-party / stdlib modules are not followed)
        rel = mod.replace(".", os.sep)
        for cand in (os.path.join(ROOT, rel + ".py"), os.path.join(ROOT, rel, "__init__.py"),
                     os.path.join(ROOT, "tests", rel + ".py"), os.path.join(ROOT, "scripts", rel + ".py")):
            if os.path.exists(cand):
                return cand
        return None

    while todo:
        path = todo.pop()
        if path in seen:
            continue
        seen.add(path)
        tree = ast.parse(open(path).read(), path)
        pkg_parts = os.path.relpath(os.path.dirname(path), ROOT).split(os.sep)
        for node in ast.walk(tree):
            mods = []
            if isinstance(node, ast.Import):
                mods = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                base = node.module or ""
                if node.level:
                    base = ".".join(pkg_parts[: len(pkg_parts) - node.level + 1] + ([node.module] if node.module else []))
                mods = [base] + [base + "." + a.name for a in node.names]
            for m in mods:
                assert m.split(".")[0] not in banned, f"{os.path.relpath(path, ROOT)} imports {m}"
                target = resolve(m)
                if target:
                    rel = os.path.relpath(target, ROOT).split(os.sep)[0]
                    assert rel not in ("oracle", "tests", "scripts"), f"{os.path.relpath(path, ROOT)} reaches {os.path.relpath(target, ROOT)}"
                    todo.append(target)
    assert any(p.endswith(os.path.join("pyruhvro_amd", "prebuild.py")) for p in seen)


def test_encode_binding_rejects_mismatched_batches_before_any_device_work():
    """rh_encode walks schema and batch together first (encoder construction, fast_encode.rs:151-358): columns are
    matched by name, Arrow types are checked, and the reference's messages come back as ValueError."""
    import pyarrow as pa
    s = SCHEMAS["flat4"]
    rb = pa.RecordBatch.from_arrays([pa.array([1], pa.int32()), pa.array([2], pa.int64()), pa.array([0.5]), pa.array([True])],
                                    names=["i", "l", "d", "b"])
    with pytest.raises(ValueError) as ei:
        P.serialize_record_batch(rb.drop_columns(["d"]), s, 1)
    assert str(ei.value) == ("Arrow struct missing column 'd' required by Avro schema. "
                             'Available columns: ["i", "l", "b"]')                     # fast_encode.rs:173-177
    wrong = rb.set_column(0, "i", pa.array([1], pa.int64()))
    with pytest.raises(ValueError) as ei:
        P.serialize_record_batch(wrong, s, 1)
    assert str(ei.value) == "fast_encode: arrow array downcast failed"
    with pytest.raises(ValueError):
        P.serialize_record_batch(rb, "{", 1)
    with pytest.raises(TypeError):
        P.serialize_record_batch(rb, s, "2")
    with pytest.raises(OverflowError):
        P.serialize_record_batch_spawn(rb, s, -1)
    su = SCHEMAS["t_union"]
    with pytest.raises(ValueError) as ei:
        P.serialize_record_batch(pa.RecordBatch.from_arrays([pa.array(["x"])], names=["u"]), su, 1)
    assert str(ei.value) == "fast_encode: expected UnionArray for multi-variant union"   # fast_encode.rs:261-263
    if not has_gpu():
        with pytest.raises(RuntimeError):          # a matching batch then needs the device: no CPU fallback
            P.serialize_record_batch(rb, s, 1)
