"""Shared test inputs: (schema JSON, list[bytes]) cases for the oracle tests (CPU)
and the GPU parity tests.

  differential_cases()  the reference's differential tests restated with the same
                        schemas and value formulas (ruhvro/src/fast_decode.rs:1008-1231)
                        plus the expected Python values;
  wire_cases()          valid-but-unusual wire forms the reference accepts (F4/F5 of
                        SURVEY.md, fast_decode.rs:689-700,825-828);
  error_cases()         malformed datums + the exact message the reference raises
                        (fast_decode.rs:575,591,646,849,866,874,884,898,906,910);
  nesting_cases()       schemas the reference never tests (union-of-record, list of
                        lists, nullable containers, children domains with bitmaps).
"""
from __future__ import annotations

import json
import struct
from typing import List, Tuple

from avrogen.encoder import Blocks, Branch, Unscaled, to_datum, zigzag
from avrogen.schemas import SCHEMAS
from oracle.avro_schema import parse_schema


def _enc(schema_json: str, values) -> List[bytes]:
    s = parse_schema(schema_json)
    return [to_datum(s, v) for v in values]


def differential_cases():
    """-> list of (name, schema_json, records, expected_pylist)."""
    out = []
    f32 = lambda x: struct.unpack("<f", struct.pack("<f", x))[0]  # noqa: E731
    # decodes_flat_primitives, fast_decode.rs:1008-1030
    vals = [{"i": i, "l": i * 100, "f": f32(i * 1.5), "d": i * 2.25, "b": i % 2 == 0, "s": f"row-{i}"} for i in range(5)]
    out.append(("flat_primitives", SCHEMAS["flat_primitives"], _enc(SCHEMAS["flat_primitives"], vals), vals))
    # decodes_nullable_primitives, 1033-1057 (both null orders)
    vals = [{"i": i if i % 2 == 0 else None, "s": None if i % 3 == 0 else f"v-{i}"} for i in range(6)]
    out.append(("nullable_primitives", SCHEMAS["t_nullable"], _enc(SCHEMAS["t_nullable"], vals), vals))
    # decodes_enum, 1079-1092
    vals = [{"e": "ABC"[i % 3]} for i in range(6)]
    out.append(("enum", SCHEMAS["t_enum"], _enc(SCHEMAS["t_enum"], vals), vals))
    # decodes_nested_record, 1095-1116
    vals = [{"outer_id": i, "inner": {"x": i, "label": f"lbl-{i}"}} for i in range(5)]
    out.append(("nested_record", SCHEMAS["t_nested"], _enc(SCHEMAS["t_nested"], vals), vals))
    # decodes_nullable_nested_record, 1119-1144
    vals = [{"inner": {"x": i} if i % 2 == 0 else None} for i in range(6)]
    out.append(("nullable_nested_record", SCHEMAS["t_nullable_nested"], _enc(SCHEMAS["t_nullable_nested"], vals), vals))
    # decodes_multi_variant_union, 1147-1165
    vals = []
    for i in range(8):
        vals.append({"u": [None, f"s-{i}", i * 11, i % 8 == 3][i % 4]})
    out.append(("multi_variant_union", SCHEMAS["t_union"], _enc(SCHEMAS["t_union"], vals), vals))
    # decodes_array_of_string, 1168-1184
    vals = [{"tags": [f"t-{i}-a", f"t-{i}-b"]} for i in range(6)]
    out.append(("array_of_string", SCHEMAS["t_array_str"], _enc(SCHEMAS["t_array_str"], vals), vals))
    # decodes_empty_array, 1187-1199
    vals = [{"tags": []} for _ in range(3)]
    out.append(("empty_array", SCHEMAS["t_array_int"], _enc(SCHEMAS["t_array_int"], vals), vals))
    # decodes_map_of_string, 1202-1231 (wire order preserved on the fast path)
    vals = [{"props": [(f"k{i}-1", f"v{i}-1"), (f"k{i}-2", f"v{i}-2")]} for i in range(4)]
    out.append(("map_of_string", SCHEMAS["t_map_str"], _enc(SCHEMAS["t_map_str"], vals), vals))
    # test_enum, deserialize.rs:312-354 (the one value-level assert the reference has on decode output)
    vals = [{"a": 27, "b": "foo", "c": "clubs"}, {"a": 28, "b": "bar", "c": "hearts"}]
    out.append(("test_enum", SCHEMAS["kat_enum"], _enc(SCHEMAS["kat_enum"], vals), vals))
    return out


def logical_case():
    """decodes_logical_types, fast_decode.rs:1060-1076 -> (schema, records, expected raw ints)."""
    vals = [{"d": i * 7, "tm": 1_700_000_000_000 + i, "tu": 1_700_000_000_000_000 + i} for i in range(4)]
    return SCHEMAS["t_logical"], _enc(SCHEMAS["t_logical"], vals), vals


def wire_cases():
    """-> list of (name, schema_json, records).  All valid for the reference's fast path."""
    out = []
    s_arr = SCHEMAS["t_array_str"]
    sa = parse_schema(s_arr)
    recs = [
        to_datum(sa, {"tags": Blocks([(["a", "bb"], True), (["ccc"], False), (["d"], True)])}),  # negative counts + byte sizes, multi-block
        to_datum(sa, {"tags": Blocks([(["x"] * 70, False), (["y"] * 3, True)])}),
        to_datum(sa, {"tags": []}),
        to_datum(sa, {"tags": ["solo"]}) + b"\xde\xad\xbe\xef",            # trailing bytes are ignored (fast_decode.rs:825-828)
    ]
    out.append(("array_blocks", s_arr, recs))
    # a block count of i64::MIN: `-n` wraps to itself (release build), `0..n` is empty, and it is not the terminator --
    # an EMPTY block with a byte size, after which the array goes on (fast_decode.rs:689-700, 703-719)
    i64min = zigzag(-(1 << 63))
    recs = [
        i64min + zigzag(5) + zigzag(2) + zigzag(1) + b"p" + zigzag(1) + b"q" + zigzag(0),
        zigzag(1) + zigzag(2) + b"ab" + i64min + zigzag(0) + i64min + zigzag(123456) + zigzag(0),
        to_datum(sa, {"tags": ["plain"]}),
    ]
    out.append(("array_block_count_i64_min", s_arr, recs))
    s_map = SCHEMAS["t_map_str"]
    sm = parse_schema(s_map)
    recs = [
        to_datum(sm, {"props": Blocks([([("k1", "v1")], True), ([("k2", "v2"), ("k1", "dup")], False)])}),
        to_datum(sm, {"props": []}),
        to_datum(sm, {"props": [("", "")]}),
    ]
    out.append(("map_blocks", s_map, recs))
    # [T, "null"] ordering next to ["null", T]
    recs = _enc(SCHEMAS["t_nullable"], [{"i": None, "s": None}, {"i": -1, "s": ""}, {"i": 2**31 - 1, "s": "x" * 300}])
    out.append(("null_orders", SCHEMAS["t_nullable"], recs))
    # int truncation (`as i32`) and extreme longs / NaN payloads
    s = json.dumps({"type": "record", "name": "X", "fields": [
        {"name": "i", "type": "int"}, {"name": "l", "type": "long"},
        {"name": "f", "type": "float"}, {"name": "d", "type": "double"}]})
    raw = []
    for iv, lv, fb, db in [(2**31 + 5, 2**63 - 1, 0x7FC00001, 0x7FF8000000000123), (-2**31 - 7, -2**63, 0xFF800000, 0x8000000000000000),
                           (2**40 + 3, 0, 0x00000001, 0x0000000000000001)]:
        raw.append(zigzag(iv) + zigzag(lv) + struct.pack("<I", fb) + struct.pack("<Q", db))
    # 10-byte varint whose high bits are dropped (fast_decode.rs:860: bits past 63 vanish)
    raw.append(b"\x02" + b"\xff" * 9 + b"\x7f" + struct.pack("<f", 1.5) + struct.pack("<d", -2.5))
    out.append(("extremes", s, raw))
    # empty strings, long strings
    recs = _enc(SCHEMAS["flat_primitives"], [
        {"i": 0, "l": 0, "f": 0.0, "d": 0.0, "b": False, "s": ""},
        {"i": -1, "l": -1, "f": -1.0, "d": -1.0, "b": True, "s": "é" * 1000},
        {"i": 1, "l": 1, "f": 1.0, "d": 1.0, "b": True, "s": "z" * 70000}])
    out.append(("strings", SCHEMAS["flat_primitives"], recs))
    # non-canonical (padded) varints everywhere a varint can stand: union branch, string length, int, long,
    # enum index, block count -- legal for the reference's reader (fast_decode.rs:854-869 accepts any
    # continuation chain up to 10 bytes); these leave the GPU's single-read fast path.
    def padded(v: int, width: int) -> bytes:
        z = (v << 1) ^ (v >> 63)
        b = bytearray()
        for i in range(width):
            b.append((z & 0x7F) | (0x80 if i < width - 1 else 0))
            z >>= 7
        assert z == 0
        return bytes(b)
    s_nc = json.dumps({"type": "record", "name": "NC", "fields": [
        {"name": "ns", "type": ["null", "string"]}, {"name": "ni", "type": ["int", "null"]},
        {"name": "l", "type": "long"}, {"name": "e", "type": {"type": "enum", "name": "E9", "symbols": ["a", "bb", "ccc"]}},
        {"name": "arr", "type": {"type": "array", "items": "int"}},
        {"name": "u", "type": ["null", "string", "int"]}, {"name": "b", "type": "boolean"}]})
    raw = []
    for w in (1, 2, 3, 5, 9, 10):
        r = padded(1, w) + padded(3, max(w, 1)) + b"abc"            # ns: branch 1, len 3
        r += padded(0, w) + padded(-123456 if w >= 3 else 7, max(w, 3))  # ni: branch 0 (int first), value
        r += padded(2**40 + 5, max(w, 6))                               # l
        r += padded(2, w)                                               # e -> "ccc"
        r += padded(2, w) + padded(10, w) + padded(-20, w) + padded(0, w)   # arr: [10, -20]
        r += padded(2, w) + padded(77, max(w, 2))                             # u: int 77
        r += b"\x01"
        raw.append(r)
    raw.append(b"\x00" + b"\x02" + zigzag(-2**62) + b"\x00" + b"\x00" + b"\x00" + b"\x00")   # nulls / minimal forms
    out.append(("noncanonical_varints", s_nc, raw))
    return out


def error_cases():
    """-> list of (name, schema_json, good_prefix_records, bad_record, message)."""
    out = []
    sj = SCHEMAS["flat_primitives"]
    s = parse_schema(sj)
    good = to_datum(s, {"i": 1, "l": 2, "f": 1.0, "d": 2.0, "b": True, "s": "ok"})
    goods = [good] * 3
    out.append(("eob_varint", sj, goods, b"", "unexpected end of buffer"))
    out.append(("eob_mid_varint", sj, goods, b"\x80", "unexpected end of buffer"))
    out.append(("varint_too_long", sj, goods, b"\x80" * 10 + b"\x00", "zigzag varint too long"))
    out.append(("eob_f32", sj, goods, zigzag(1) + zigzag(2) + b"\x00\x00", "unexpected end of buffer (f32)"))
    out.append(("eob_f64", sj, goods, zigzag(1) + zigzag(2) + b"\x00" * 4 + b"\x00" * 7, "unexpected end of buffer (f64)"))
    out.append(("bad_bool", sj, goods, zigzag(1) + zigzag(2) + b"\x00" * 12 + b"\x07", "invalid boolean byte: 7"))
    out.append(("eob_bool", sj, goods, zigzag(1) + zigzag(2) + b"\x00" * 12, "unexpected end of buffer"))
    out.append(("neg_strlen", sj, goods, zigzag(1) + zigzag(2) + b"\x00" * 12 + b"\x01" + zigzag(-3), "negative string length"))
    out.append(("eob_string", sj, goods, zigzag(1) + zigzag(2) + b"\x00" * 12 + b"\x01" + zigzag(10) + b"abc",
                "unexpected end of buffer (string)"))
    se = SCHEMAS["t_enum"]
    ge = [b"\x00", b"\x02", b"\x04"]
    out.append(("enum_oor", se, ge, zigzag(3), "enum index 3 out of range"))
    out.append(("enum_negative", se, ge, zigzag(-1), "enum index 18446744073709551615 out of range"))
    sn = SCHEMAS["t_nullable"]
    gn = _enc(sn, [{"i": 1, "s": "a"}, {"i": None, "s": None}])
    out.append(("bad_branch", sn, gn, zigzag(2), "invalid union branch index: 2"))
    out.append(("bad_branch_neg", sn, gn, zigzag(-1), "invalid union branch index: -1"))
    su = SCHEMAS["t_union"]
    gu = _enc(su, [{"u": None}, {"u": "x"}, {"u": 3}, {"u": True}])
    out.append(("union_oor", su, gu, zigzag(4), "union branch index out of range: 4"))
    out.append(("union_neg", su, gu, zigzag(-2), "union branch index out of range: -2"))
    sa = SCHEMAS["t_array_str"]
    ga = _enc(sa, [{"tags": ["a"]}, {"tags": []}])
    out.append(("array_unterminated", sa, ga, zigzag(1) + zigzag(1) + b"a", "unexpected end of buffer"))
    out.append(("array_huge_count", sa, ga, zigzag(2**40) + zigzag(1) + b"a", "unexpected end of buffer"))
    out.append(("array_item_eob", sa, ga, zigzag(2) + zigzag(1) + b"a" + zigzag(5) + b"ab", "unexpected end of buffer (string)"))
    return out


def nesting_cases():
    """Schemas beyond the reference's own tests; expected values come from the oracle."""
    out = []
    # record / array / map as variants of an N-variant union; enum variant named by its fullname
    s = json.dumps({"type": "record", "name": "U", "namespace": "ns.x", "fields": [
        {"name": "u", "type": ["null", {"type": "record", "name": "R", "fields": [
            {"name": "a", "type": "int"}, {"name": "b", "type": ["null", "string"]}]},
            {"type": "array", "items": "long"}, {"type": "enum", "name": "E", "symbols": ["X", "YY"]}, "double"]},
        {"name": "tail", "type": "boolean"}]})
    vals = []
    for i in range(40):
        m = i % 6
        u = [None, {"a": i, "b": None}, {"a": -i, "b": f"b{i}"}, list(range(i % 5)), Branch(3, i % 2), float(i) / 3][m]
        vals.append({"u": u, "tail": i % 3 == 0})
    out.append(("union_of_containers", s, _enc(s, vals)))
    # list of lists of nullable strings, list of nullable records with bools (child-domain bitmaps)
    s = json.dumps({"type": "record", "name": "N", "fields": [
        {"name": "ll", "type": {"type": "array", "items": {"type": "array", "items": ["null", "string"]}}},
        {"name": "lr", "type": {"type": "array", "items": ["null", {"type": "record", "name": "Q", "fields": [
            {"name": "flag", "type": "boolean"}, {"name": "v", "type": ["null", "int"]},
            {"name": "m", "type": {"type": "map", "values": ["null", "boolean"]}}]}]}},
        {"name": "id", "type": "int"}]})
    vals = []
    for i in range(60):
        ll = [[None if (i + j + k) % 3 == 0 else f"s{i}-{j}-{k}" for k in range((i + j) % 4)] for j in range(i % 3)]
        lr = [None if (i + j) % 4 == 0 else {"flag": (i + j) % 2 == 0, "v": None if j % 2 else i * j,
                                             "m": [(f"k{j}{t}", [None, True, False][(i + t) % 3]) for t in range(j % 3)]}
              for j in range(i % 5)]
        vals.append({"ll": ll, "lr": lr, "id": i})
    out.append(("nested_lists", s, _enc(s, vals)))
    # nullable array, [T,"null"] order, nullable enum, date/timestamps nullable.  (A nullable MAP cannot be
    # translated by the reference at all: default_field_name panics on Map, schema_translate.rs:88,148,212.)
    s = json.dumps({"type": "record", "name": "C", "fields": [
        {"name": "na", "type": ["null", {"type": "array", "items": "int"}]},
        {"name": "ne", "type": ["null", {"type": "enum", "name": "E2", "symbols": ["p", "qq", "rrr"]}]},
        {"name": "nd", "type": ["null", {"type": "int", "logicalType": "date"}]},
        {"name": "nt", "type": [{"type": "long", "logicalType": "timestamp-micros"}, "null"]},
        {"name": "nf", "type": ["null", "float"]}, {"name": "nb", "type": ["boolean", "null"]}]})
    vals = []
    for i in range(50):
        vals.append({"na": None if i % 3 == 0 else list(range(i % 4)),
                     "ne": None if i % 5 == 0 else ["p", "qq", "rrr"][i % 3],
                     "nd": None if i % 2 else i * 100, "nt": None if i % 3 == 2 else 1_700_000_000_000_000 + i,
                     "nf": None if i % 7 == 0 else float(i) * 0.5, "nb": None if i % 4 == 3 else i % 2 == 0})
    out.append(("nullable_containers", s, _enc(s, vals)))
    # deep nullable nesting: null-fill must cascade through records, unions and lists
    s = json.dumps({"type": "record", "name": "D", "fields": [
        {"name": "o", "type": ["null", {"type": "record", "name": "O1", "fields": [
            {"name": "p", "type": ["null", {"type": "record", "name": "O2", "fields": [
                {"name": "u", "type": ["null", "int", "string"]},
                {"name": "arr", "type": {"type": "array", "items": "string"}},
                {"name": "plain", "type": {"type": "record", "name": "O3", "fields": [
                    {"name": "z", "type": "long"}, {"name": "e", "type": {"type": "enum", "name": "E3", "symbols": ["a", "b"]}}]}}]}]},
            {"name": "q", "type": "double"}]}]}]})
    vals = []
    for i in range(45):
        if i % 3 == 0:
            o = None
        else:
            p = None if i % 4 == 1 else {"u": [None, i, f"u{i}"][i % 3], "arr": [f"a{i}"] * (i % 3),
                                         "plain": {"z": i * 1000, "e": "ab"[i % 2]}}
            o = {"p": p, "q": i / 7}
        vals.append({"o": o})
    out.append(("deep_nullfill", s, _enc(s, vals)))
    return out


# schemas that only the Arrow -> Avro GPU tests use (tests/test_gpu_encode.py); listed here so that
# scripts/known_schemas.py compiles their specialised kernels ahead of the GPU run
ENC_WINDOW_SCHEMA = json.dumps({"type": "record", "name": "r", "fields": [
    {"name": "id", "type": "long"}, {"name": "t", "type": "string"}, {"name": "o", "type": ["null", "string"]},
    {"name": "xs", "type": {"type": "array", "items": "string"}}]})
ENC_VALIDITY_SCHEMA = json.dumps({"type": "record", "name": "r", "fields": [
    {"name": "a", "type": "long"}, {"name": "b", "type": ["null", "string"]}]})
ENC_LONG_ENUM_SCHEMA = json.dumps({"type": "record", "name": "r", "fields": [
    {"name": "e", "type": {"type": "enum", "name": "E", "symbols": ["SHORT", "A_SYMBOL_LONGER_THAN_SIXTEEN_BYTES", "MID_LENGTH_SYM16"]}},
    {"name": "f", "type": ["null", {"type": "enum", "name": "F", "symbols": ["x", "yy", "zzz", "wwww", "vvvvv"]}]}]})


def encode_extra_schemas():
    return [ENC_WINDOW_SCHEMA, ENC_VALIDITY_SCHEMA, ENC_LONG_ENUM_SCHEMA]


def long_string_case(n=1500, seed=5):
    """Strings of 0..3000 bytes at the top level, inside a nullable record, in an array and as map keys / values:
    wave spans from a few bytes to far beyond any staging area, so columns take the staged AND the per-lane path."""
    import random
    import json
    s = json.dumps({"type": "record", "name": "LS", "fields": [
        {"name": "a", "type": "string"},
        {"name": "b", "type": ["null", "string"]},
        {"name": "r", "type": ["null", {"type": "record", "name": "RR", "fields": [
            {"name": "x", "type": "string"}, {"name": "y", "type": ["string", "null"]}]}]},
        {"name": "l", "type": {"type": "array", "items": "string"}},
        {"name": "m", "type": {"type": "map", "values": ["null", "string"]}},
        {"name": "ll", "type": {"type": "array", "items": {"type": "array", "items": "string"}}},
        {"name": "z", "type": "string"}]})
    r = random.Random(seed)

    def st(big=False):
        k = r.random()
        ln = 0 if k < 0.15 else r.randint(1, 7) if k < 0.5 else r.randint(8, 40) if k < 0.93 else r.randint(41, 3000 if big else 300)
        return "".join(chr(r.randint(33, 126)) for _ in range(ln))
    vals = []
    for i in range(n):
        vals.append({"a": st(i % 97 == 0), "b": None if r.random() < 0.4 else st(),
                     "r": None if r.random() < 0.3 else {"x": st(), "y": None if r.random() < 0.5 else st()},
                     "l": [st() for _ in range(r.choice([0, 0, 1, 2, 5]))],
                     "m": {f"k{j}{st()[:6]}": (None if r.random() < 0.3 else st()) for j in range(r.choice([0, 1, 3]))},
                     "ll": [[st() for _ in range(r.choice([0, 1, 3]))] for _ in range(r.choice([0, 1, 2]))],
                     "z": st()})
    return s, _enc(s, vals)


def wide_counter_cases():
    """Schemas with MORE THAN 64 scanned counters (one per string column, one per child row domain; the engine's limit is
    96, program.h kMaxCounters): the specialised emit kernel's workgroup prefix takes the counters from lanes with
    v_readlane, which selects among 64 lanes only (counters 64.. got counter k - 64's prefix).  70 and 96
    counters, every column with its own length pattern so that a swapped prefix cannot go unnoticed.
    -> list of (name, schema_json, records)."""
    out = []
    for total in (70, 96):
        nstr = total - 2                                   # + the array's row domain and its item bytes
        fields = [{"name": f"s{i}", "type": "string" if i % 3 else ["null", "string"]} for i in range(nstr // 2)]
        fields.append({"name": "tags", "type": {"type": "array", "items": "string"}})
        fields += [{"name": f"s{i}", "type": "string"} for i in range(nstr // 2, nstr)]
        sj = json.dumps({"type": "record", "name": f"Wide{total}", "fields": fields})
        vals = []
        for r in range(700):
            v = {}
            for i in range(nstr):
                if i < nstr // 2 and i % 3 == 0 and (r + i) % 4 == 0:
                    v[f"s{i}"] = None
                else:
                    # (short strings: 256 records of either schema still fit one LDS window, so the tiles take the staged walk)
                    v[f"s{i}"] = chr(65 + i % 26) * ((r * (i + 1) + i) % (4 if total == 70 else 3)) + (f"<{r}>" if (r + i) % 23 == 0 else "")
            v["tags"] = [f"t{r}.{j}" for j in range((r * 3) % 5)]
            vals.append(v)
        out.append((f"wide_{total}_counters", sj, _enc(sj, vals)))
    return out


def enum_form_cases():
    """Enums on both sides of the specialised kernels' symbols-as-immediates rule (specialize.cpp Spec::enum_sym: at most 16
    symbols of at most 8 bytes get their length from a select chain in the size pass; anything else reads the symbol
    table): lengths 1..8 mixed, exactly 16 and 17 symbols, a 9-byte symbol, nullable / in a list / in a union.
    -> list of (name, schema_json, records)."""
    sy8 = ["a", "bb", "ccc", "dddd", "eeeee", "ffffff", "ggggggg", "hhhhhhhh"]
    sy16 = [f"S{i:02d}" for i in range(16)]
    sy17 = [f"T{i}" for i in range(17)]
    sy9 = ["short", "ninebytes", "x"]
    s = json.dumps({"type": "record", "name": "EF", "fields": [
        {"name": "e8", "type": {"type": "enum", "name": "E8", "symbols": sy8}},
        {"name": "e16", "type": ["null", {"type": "enum", "name": "E16", "symbols": sy16}]},
        {"name": "e17", "type": {"type": "enum", "name": "E17", "symbols": sy17}},
        {"name": "e9", "type": {"type": "array", "items": {"type": "enum", "name": "E9", "symbols": sy9}}},
        {"name": "u", "type": ["null", "int", {"type": "enum", "name": "EU", "symbols": ["only"]}]},
        {"name": "l8", "type": {"type": "array", "items": {"type": "enum", "name": "E8b", "symbols": sy8[::-1]}}},
        {"name": "tail", "type": "string"}]})
    vals = [{"e8": sy8[i % 8], "e16": None if i % 5 == 0 else sy16[(i * 7) % 16], "e17": sy17[(i * 3) % 17],
             "e9": [sy9[(i + j) % 3] for j in range(i % 4)], "u": [None, i, "only"][i % 3],
             "l8": [sy8[(i * j) % 8] for j in range(i % 6)], "tail": f"t{i}"} for i in range(700)]
    return [("enum_forms", s, _enc(s, vals))]


def dense_list_cases():
    """Top-level arrays / maps in the shapes the specialised emit kernel handles one lane per ITEM (spec_body.h
    dense_list): every body kind, more items per wavefront than its position table holds (several rounds), one huge list
    among short ones, positive multi-block lists, two-byte block counts, nullable lists, N4 leaves in list items.  All
    records take the fast wire forms, so the tiles stay on the trusted fast walk.  -> list of (name, schema_json, records)."""
    out = []
    s = json.dumps({"type": "record", "name": "DL1", "fields": [
        {"name": "id", "type": "long"}, {"name": "tags", "type": {"type": "array", "items": "string"}},
        {"name": "m", "type": {"type": "map", "values": "string"}}, {"name": "tail", "type": ["null", "int"]}]})
    vals = [{"id": i, "tags": [f"t{i}-{j}" * (1 + (i + j) % 4) for j in range(10 + i % 31)],
             "m": [(f"k{j}", "v" * ((i * j) % 23)) for j in range(i % 7)], "tail": None if i % 3 else i} for i in range(600)]
    out.append(("dense_long_lists", s, _enc(s, vals)))
    vals = [{"id": i, "tags": [f"x{j}" for j in range(700 if i % 97 == 5 else i % 3)],
             "m": [(f"key{i}{j}", f"value-{j}") for j in range(300 if i % 131 == 7 else i % 2)], "tail": i} for i in range(700)]
    out.append(("dense_skewed", s, _enc(s, vals)))
    vals = []
    for i in range(520):
        items = [f"it{i}.{j}" for j in range(i % 9)]
        blocks = [(items[b:b + 2 + i % 2], False) for b in range(0, len(items), 2 + i % 2)]        # positive counts only
        pairs = [(f"k{j}", f"{i}:{j}") for j in range(i % 5)]
        vals.append({"id": -i, "tags": Blocks(blocks), "m": Blocks([(pairs[b:b + 1], False) for b in range(len(pairs))]),
                     "tail": None})
    out.append(("dense_multiblock_positive", s, _enc(s, vals)))
    vals = [{"id": i, "tags": ["q" * (j % 5) for j in range(64 + i % 120)], "m": [(f"{j}", "") for j in range(70)], "tail": 1}
            for i in range(300)]
    out.append(("dense_two_byte_counts", s, _enc(s, vals)))
    s2 = json.dumps({"type": "record", "name": "DL2", "fields": [
        {"name": "rs", "type": {"type": "array", "items": {"type": "record", "name": "It", "fields": [
            {"name": "a", "type": "int"}, {"name": "s", "type": ["null", "string"]},
            {"name": "e", "type": {"type": "enum", "name": "DE", "symbols": ["red", "green", "b"]}},
            {"name": "u", "type": ["null", "long", "string", "boolean"]}, {"name": "d", "type": "double"},
            {"name": "nb", "type": ["null", "boolean"]}]}}},
        {"name": "mr", "type": {"type": "map", "values": ["null", {"type": "record", "name": "Mv", "fields": [
            {"name": "x", "type": "float"}, {"name": "t", "type": "string"}]}]}},
        {"name": "es", "type": {"type": "array", "items": {"type": "enum", "name": "DE2", "symbols": ["A", "BB", "CCC"]}}},
        {"name": "ni", "type": {"type": "array", "items": ["null", "int"]}},
        {"name": "after", "type": "string"}]})
    vals = []
    for i in range(560):
        rs = [{"a": i * j, "s": None if (i + j) % 3 == 0 else f"s{i}/{j}", "e": ["red", "green", "b"][(i + j) % 3],
               "u": [None, i * 1000 + j, f"u{j}", j % 2 == 0][(i + j) % 4], "d": i / (j + 1), "nb": [None, True, False][(i * j) % 3]}
              for j in range(i % 6)]
        mr = [(f"m{j}", None if (i + j) % 4 == 0 else {"x": float(j), "t": "t" * (j % 9)}) for j in range(i % 4)]
        vals.append({"rs": rs, "mr": mr, "es": ["A", "BB", "CCC"][: i % 4] * (1 + i % 3), "ni": [None if j % 2 else j for j in range(i % 8)],
                     "after": f"after{i}"})
    out.append(("dense_record_items", s2, _enc(s2, vals)))
    s3 = json.dumps({"type": "record", "name": "DL3", "fields": [
        {"name": "na", "type": ["null", {"type": "array", "items": "string"}]},
        {"name": "an", "type": [{"type": "array", "items": "long"}, "null"]},
        {"name": "fx", "type": {"type": "array", "items": {"type": "fixed", "name": "F4", "size": 4}}},
        {"name": "dm", "type": {"type": "map", "values": {"type": "bytes", "logicalType": "decimal", "precision": 12, "scale": 2}}},
        {"name": "k", "type": "int"}]})
    vals = [{"na": None if i % 4 == 0 else [f"n{j}" for j in range(i % 5)], "an": None if i % 3 == 1 else [i * j for j in range(i % 6)],
             "fx": [bytes([i % 256, j, 3, 4]) for j in range(i % 4)], "dm": [(f"d{j}", Unscaled((-1) ** j * (i * 37 + j))) for j in range(i % 3)],
             "k": i} for i in range(530)]
    out.append(("dense_nullable_and_n4", s3, _enc(s3, vals)))
    return out


def wide_form_cases():
    """Round 6: values at both edges of every single-read wire form of the fast walk (walk.h: int <= 5
    bytes, long <= 10 bytes also behind a branch byte, length / block count <= 3 bytes, index <= 2 bytes) -- what production data
    carries where the benchmark generator does not: microsecond timestamps, epoch seconds as int, snowflake ids, strings of 8 KiB,
    arrays of more than 8,191 items.  Whole wavefronts stay inside the forms (the first 640 records are small: their tiles take the
    staged fast walk), the tail holds the large values (over-window tiles).  -> list of (name, schema_json, records)."""
    out = []
    s = json.dumps({"type": "record", "name": "WF", "fields": [
        {"name": "ni", "type": ["null", "int"]}, {"name": "i", "type": "int"},
        {"name": "nl", "type": ["null", "long"]}, {"name": "l", "type": "long"},
        {"name": "ts", "type": [{"type": "long", "logicalType": "timestamp-micros"}, "null"]},
        {"name": "s", "type": "string"}, {"name": "ns", "type": ["null", "string"]},
        {"name": "arr", "type": {"type": "array", "items": "string"}},
        {"name": "ai", "type": {"type": "array", "items": "int"}},
        {"name": "e", "type": {"type": "enum", "name": "WE", "symbols": ["x", "yy", "zzz"]}},
        {"name": "u", "type": ["null", "string", "int", "long"]},
        {"name": "tail", "type": "boolean"}]})
    ints = [0, 63, -64, 64, -65, 8191, -8192, 8192, 2**20 - 1, 2**20, -2**20 - 1, 2**27 - 1, -2**27, 2**27, 2**31 - 1, -2**31,
            1_750_000_000, -1_750_000_000, -1]
    longs = [0, 2**27, 2**34 - 1, 2**34, 2**41, 2**48 - 1, 2**48, -2**48 - 1, 2**55 - 1, 2**55, -2**55 - 1, 2**62 - 1, 2**62,
             -2**62 - 1, 2**63 - 1, -2**63, 1_750_000_000_000_000, 1_934_000_000_000_000_000, -1]
    vals = []
    for r in range(700):
        big = r >= 640
        sl = [0, 1, 7, 8, 40, 41, 63, 64][r % 8] if not big else [8191, 8192, 8193, 20000, 70000, 2**20 - 1, 2**20, 300][r % 8]
        na = r % 4 if not big else [8191, 8192, 10000, 3][r % 4]
        vals.append({"ni": None if r % 5 == 0 else ints[r % len(ints)], "i": ints[(r * 7) % len(ints)],
                     "nl": None if r % 7 == 0 else longs[r % len(longs)], "l": longs[(r * 5) % len(longs)],
                     "ts": None if r % 11 == 0 else 1_750_000_000_000_000 + r * 1_000_003,
                     "s": chr(97 + r % 26) * sl, "ns": None if r % 3 == 0 else "N" * (sl // 3),
                     "arr": [f"a{r}.{j}" for j in range(na)], "ai": [ints[(r + j) % len(ints)] for j in range(na if na < 100 else 8200)],
                     "e": ["x", "yy", "zzz"][r % 3],
                     "u": [None, f"u{r}", ints[r % len(ints)], Branch(3, longs[r % len(longs)])][r % 4], "tail": r % 2 == 0})
    out.append(("wide_forms", s, _enc(s, vals)))

    # the same widths as PADDED encodings of small values (legal: fast_decode.rs:854-869 takes any continuation chain of up to 10
    # bytes), at and one byte beyond every form's width: the value is the same, which walk decodes it differs
    def padded(v: int, width: int) -> bytes:
        z = ((v << 1) ^ (v >> 63)) & ((1 << 64) - 1)
        b = bytearray()
        for k in range(width):
            b.append((z & 0x7F) | (0x80 if k < width - 1 else 0))
            z >>= 7
        assert z == 0
        return bytes(b)
    s2 = json.dumps({"type": "record", "name": "WP", "fields": [
        {"name": "ni", "type": ["null", "int"]}, {"name": "nl", "type": ["null", "long"]},
        {"name": "ns", "type": ["null", "string"]}, {"name": "arr", "type": {"type": "array", "items": "int"}},
        {"name": "e", "type": {"type": "enum", "name": "PE", "symbols": ["x", "yy", "zzz"]}}]})
    raw = []
    for r in range(200):
        wi = [1, 4, 5, 6, 5][r % 5]          # int: 5 is the form's width, 6 beyond it
        wl = [1, 8, 9, 10, 10][r % 5]        # long: all inside the 10-byte form
        ws = [1, 2, 3, 4, 3][r % 5]          # length: 3 inside, 4 beyond
        wc = [1, 2, 3, 4, 2][r % 5]          # block count
        we = [1, 2, 3, 2, 1][r % 5]          # index: 2 inside, 3 beyond
        only_fast = r < 128                  # two wavefronts that never leave the forms
        if only_fast:
            wi, ws, wc, we = min(wi, 5), min(ws, 3), min(wc, 3), min(we, 2)
        rec = b"\x02" + padded(-(r + 1) * 1000 if r % 2 else r * 77, max(wi, 3))
        rec += b"\x02" + padded((r - 100) * (1 << 33), max(wl, 6))
        rec += b"\x02" + padded(3, ws) + b"abc"
        rec += padded(2, wc) + padded(r, 2) + padded(-r, 2) + padded(0, wc)
        rec += padded(r % 3, we)
        raw.append(rec)
    # ten-byte longs whose tenth byte carries more than bit 0 (the reference drops those bits: `<< 63`), an eleventh byte is an error
    raw.append(b"\x02\x00" + b"\x02" + b"\xff" * 9 + b"\x7f" + b"\x00" + b"\x00" + b"\x00")
    raw.append(b"\x02\x00" + b"\x02" + b"\x80" * 9 + b"\x01" + b"\x00" + b"\x00" + b"\x00")
    out.append(("padded_at_form_widths", s2, raw))
    return out


def giant_record_cases():
    """Round 6: records far larger than any LDS window between ordinary ones -- arrays of tens of
    thousands of strings / ints / nullable records (child-domain bitmaps), a map, a nested array, megabyte strings in front of and
    behind them -- so that a tile holds ranges that fit, ranges of one sliding record, and ordinary records again.
    -> list of (name, schema_json, records)."""
    out = []
    s = json.dumps({"type": "record", "name": "G", "fields": [
        {"name": "id", "type": "long"},
        {"name": "pre", "type": ["null", "string"]},
        {"name": "tags", "type": {"type": "array", "items": "string"}},
        {"name": "nums", "type": {"type": "array", "items": ["null", "int"]}},
        {"name": "recs", "type": {"type": "array", "items": {"type": "record", "name": "GI", "fields": [
            {"name": "k", "type": "string"}, {"name": "v", "type": ["null", "long"]}, {"name": "f", "type": "boolean"}]}}},
        {"name": "m", "type": {"type": "map", "values": "int"}},
        {"name": "nest", "type": {"type": "array", "items": {"type": "array", "items": "int"}}},
        {"name": "post", "type": "string"}]})
    vals = []
    for r in range(420):
        giant = r in (3, 64, 65, 200, 419)
        n = [0, 1, 2, 5][r % 4]
        if giant:
            n = [30_000, 9_000, 70_000, 12_345, 40_000][(3, 64, 65, 200, 419).index(r)]
        vals.append({"id": r * 1_000_003, "pre": None if r % 3 == 0 else ("P" * (1_200_000 if r == 200 else r % 50)),
                     "tags": [f"tag-{r}-{j}" * (1 + j % 3) for j in range(n)],
                     "nums": [None if (r + j) % 5 == 0 else j * 7 - r for j in range(n // 2)],
                     "recs": [{"k": f"k{j}", "v": None if j % 3 == 0 else j * 1_000_000_007, "f": j % 2 == 0} for j in range(n // 3)],
                     "m": [(f"key{j}", j) for j in range(n if n < 100 else 2_000)],
                     "nest": [[j, j + 1, j + 2][: j % 4] for j in range(n if n < 100 else 5_000)],
                     "post": "z" * (r % 20) if r != 65 else "Q" * 300_000})
    out.append(("giant_arrays", s, _enc(s, vals)))
    # Nested (non item-dense) lists in FRONT of dense ones inside a record larger than the window, long strings inside the nested
    # lists and behind them: the compiler may run a block loop with its live lanes only, so a window refill, a long string's
    # cooperative copy and the dense list behind the loop all meet lanes that sat out earlier refills (profiles/r06_o_*).
    s = json.dumps({"type": "record", "name": "GN", "fields": [
        {"name": "id", "type": "int"},
        {"name": "words", "type": {"type": "array", "items": {"type": "array", "items": "string"}}},
        {"name": "mid", "type": "string"},
        {"name": "tags", "type": {"type": "array", "items": "string"}},
        {"name": "groups", "type": {"type": "map", "values": {"type": "array", "items": ["null", "long"]}}},
        {"name": "opt", "type": ["null", {"type": "array", "items": "string"}]},
        {"name": "tail", "type": ["null", "string"]}]})
    vals = []
    for r in range(300):
        giant = r in (7, 130, 131, 299)
        n = [0, 1, 3, 2][r % 4]
        if giant:
            n = [6_000, 20_000, 3_000, 9_000][(7, 130, 131, 299).index(r)]
        vals.append({"id": r - 150,
                     "words": [[("w%d-%d-%d" % (r, j, i)) * (1 + (40 if (j + i) % 97 == 0 and giant else (j + i) % 3)) for i in range(j % 4)]
                               for j in range(n)],
                     "mid": "M" * (70_000 if r == 130 else r % 9),
                     "tags": [f"t{j}" * (1 + j % 2) for j in range(n // 2)],
                     "groups": [(f"g{j}", [None if (i + j) % 4 == 0 else (j << 20) - i for i in range(j % 5)]) for j in range(min(n, 1_500))],
                     "opt": None if r % 5 == 0 else [chr(48 + j % 70) * (300 if j % 501 == 0 and giant else j % 6) for j in range(n // 4)],
                     "tail": None if r % 2 else "x" * (r % 300)})
    out.append(("giant_nested", s, _enc(s, vals)))
    return out


def deep_nesting_cases():
    """Round 6: nesting beyond rounds 1-5's limits (8 list levels, 30 nullable-record / union / list levels, 8 N-variant unions):
    12 nested arrays, 36 nested nullable records, 10 nested N-variant unions.  -> list of (name, schema_json, records)."""
    out = []
    # array<array<...<int>>> twelve deep
    t = "int"
    for _ in range(12):
        t = {"type": "array", "items": t}
    s = json.dumps({"type": "record", "name": "DL", "fields": [{"name": "a", "type": t}, {"name": "z", "type": "string"}]})

    def nest(d, r):
        if d == 0:
            return r
        return [nest(d - 1, r + j) for j in range((r + d) % 3)]
    vals = [{"a": nest(12, r), "z": f"z{r}"} for r in range(90)]
    out.append(("lists_12_deep", s, _enc(s, vals)))
    # 36 nested nullable records, a string at every level
    t = ["null", {"type": "record", "name": "N0", "fields": [{"name": "s", "type": "string"}]}]
    for i in range(1, 36):
        t = ["null", {"type": "record", "name": f"N{i}", "fields": [{"name": "s", "type": ["null", "string"]}, {"name": "c", "type": t}]}]
    s = json.dumps({"type": "record", "name": "DR", "fields": [{"name": "r", "type": t}, {"name": "tail", "type": "int"}]})

    def rec(level, depth, r):
        if depth == 0:
            return None
        if level == 0:
            return {"s": f"leaf{r}"}
        return {"s": None if (r + level) % 3 == 0 else f"s{level}.{r}", "c": rec(level - 1, depth - 1, r)}
    vals = [{"r": rec(35, (r * 7) % 38, r), "tail": r} for r in range(150)]
    out.append(("nullable_records_36_deep", s, _enc(s, vals)))
    # ten nested N-variant unions: ["null", "int", record{u: <next>}]
    t = ["null", "int", "string"]
    for i in range(10):
        t = ["null", "int", {"type": "record", "name": f"U{i}", "fields": [{"name": "u", "type": t}, {"name": "k", "type": "long"}]}]
    s = json.dumps({"type": "record", "name": "DU", "fields": [{"name": "u", "type": t}]})

    def un(level, r):
        if level == 0:
            return [None, r, f"str{r}"][r % 3]
        m = (r + level) % 4
        if m == 0:
            return None
        if m == 1:
            return r * level
        return {"u": un(level - 1, r + 1), "k": r * 1_000_003 + level}
    vals = [{"u": un(10, r)} for r in range(160)]
    out.append(("unions_10_deep", s, _enc(s, vals)))
    return out
