"""Named Avro schemas (JSON strings) used by tests and bench.py.

Sources (all restated, none copied verbatim beyond the schema definitions the
reference publishes as its benchmark inputs):
  full                 scripts/generate_avro.py:12-41  (BASELINE.json configs 1, 4, 5)
  flat4                BASELINE.json config 2 = FLAT_PRIMITIVES (benches/common/mod.rs:37-50) minus f, s
  cfg3                 BASELINE.json config 3
  flat_primitives ...  ruhvro/benches/common/mod.rs:37-50, 67-81, 102-117, 135-146
  t_*                  the differential-test schemas of ruhvro/src/fast_decode.rs:1008-1231
  kat_*                golden-vector schemas: ruhvro/src/lib.rs:65-161, deserialize.rs:187-242, 254-301, 313-331
"""
import json

_ADDRESS = {"type": "record", "name": "Address", "fields": [
    {"name": "street", "type": "string"}, {"name": "city", "type": "string"}, {"name": "zipcode", "type": "string"}]}
_PREFS = {"type": "record", "name": "Preferences", "fields": [
    {"name": "contact_method", "type": ["null", "string"], "default": None},
    {"name": "newsletter", "type": "boolean"}]}

_USER_COMMON = [
    {"name": "name", "type": ["null", "string"], "default": None},
    {"name": "age", "type": ["null", "int"], "default": None},
    {"name": "emails", "type": {"type": "array", "items": "string"}},
    {"name": "address", "type": ["null", _ADDRESS], "default": None},
    {"name": "phone_numbers", "type": {"type": "map", "values": "string"}},
    {"name": "preferences", "type": ["null", _PREFS], "default": None},
]

FULL = {"type": "record", "name": "User", "fields": _USER_COMMON + [
    {"name": "status", "type": ["null", "string", "int", "boolean"], "default": None},
    {"name": "created_at", "type": "long"},
    {"name": "class", "type": {"type": "enum", "name": "enum_col", "symbols": ["A", "B", "C"]}},
]}

KAT_USER = {"type": "record", "name": "User", "fields": _USER_COMMON + [
    {"name": "status", "type": ["string", "int", "boolean"]},
]}

FLAT4 = {"type": "record", "name": "FlatPrim", "fields": [
    {"name": "i", "type": "int"}, {"name": "l", "type": "long"},
    {"name": "d", "type": "double"}, {"name": "b", "type": "boolean"}]}

CFG3 = {"type": "record", "name": "Cfg3", "fields": [
    {"name": "id", "type": "long"},
    {"name": "name", "type": ["null", "string"], "default": None},
    {"name": "age", "type": ["null", "int"], "default": None},
    {"name": "s", "type": "string"},
    {"name": "class", "type": {"type": "enum", "name": "enum_col", "symbols": ["A", "B", "C"]}}]}

FLAT_PRIMITIVES = {"type": "record", "name": "FlatPrim", "fields": [
    {"name": "i", "type": "int"}, {"name": "l", "type": "long"}, {"name": "f", "type": "float"},
    {"name": "d", "type": "double"}, {"name": "b", "type": "boolean"}, {"name": "s", "type": "string"}]}

NULLABLE_PRIMITIVES = {"type": "record", "name": "NullPrim", "fields": [
    {"name": "i", "type": ["null", "int"], "default": None},
    {"name": "l", "type": ["null", "long"], "default": None},
    {"name": "d", "type": ["null", "double"], "default": None},
    {"name": "b", "type": ["null", "boolean"], "default": None},
    {"name": "s", "type": ["null", "string"], "default": None}]}

NESTED_STRUCT = {"type": "record", "name": "Outer", "fields": [
    {"name": "outer_id", "type": "long"},
    {"name": "inner", "type": {"type": "record", "name": "Inner", "fields": [
        {"name": "x", "type": "int"}, {"name": "y", "type": "int"}, {"name": "label", "type": "string"}]}}]}

ARRAY_AND_MAP = {"type": "record", "name": "Collection", "fields": [
    {"name": "id", "type": "long"},
    {"name": "tags", "type": {"type": "array", "items": "string"}},
    {"name": "props", "type": {"type": "map", "values": "string"}}]}

T_NULLABLE = {"type": "record", "name": "P", "fields": [
    {"name": "i", "type": ["null", "int"], "default": None},
    {"name": "s", "type": ["string", "null"], "default": ""}]}

T_LOGICAL = {"type": "record", "name": "L", "fields": [
    {"name": "d", "type": {"type": "int", "logicalType": "date"}},
    {"name": "tm", "type": {"type": "long", "logicalType": "timestamp-millis"}},
    {"name": "tu", "type": {"type": "long", "logicalType": "timestamp-micros"}}]}

T_ENUM = {"type": "record", "name": "R", "fields": [
    {"name": "e", "type": {"type": "enum", "name": "E", "symbols": ["A", "B", "C"]}}]}

T_NESTED = {"type": "record", "name": "O", "fields": [
    {"name": "outer_id", "type": "long"},
    {"name": "inner", "type": {"type": "record", "name": "I", "fields": [
        {"name": "x", "type": "int"}, {"name": "label", "type": "string"}]}}]}

T_NULLABLE_NESTED = {"type": "record", "name": "O", "fields": [
    {"name": "inner", "type": ["null", {"type": "record", "name": "I", "fields": [
        {"name": "x", "type": "int"}]}], "default": None}]}

T_UNION = {"type": "record", "name": "M", "fields": [
    {"name": "u", "type": ["null", "string", "int", "boolean"]}]}

T_ARRAY_STR = {"type": "record", "name": "C", "fields": [
    {"name": "tags", "type": {"type": "array", "items": "string"}}]}

T_ARRAY_INT = {"type": "record", "name": "C", "fields": [
    {"name": "tags", "type": {"type": "array", "items": "int"}}]}

T_MAP_STR = {"type": "record", "name": "C", "fields": [
    {"name": "props", "type": {"type": "map", "values": "string"}}]}

KAT_USERDATA = {"type": "record", "name": "UserData", "namespace": "com.example", "fields": [
    {"name": "userId", "type": "string"},
    {"name": "age", "type": "int"},
    {"name": "fullName", "type": {"type": "record", "name": "FullName", "fields": [
        {"name": "firstName", "type": "string"}, {"name": "lastName", "type": "string"}]}},
    {"name": "email", "type": ["null", "string"], "default": None},
    {"name": "phoneNumbers", "type": {"type": "array", "items": "string"}},
    {"name": "isPremiumMember", "type": "boolean"},
    {"name": "favoriteItems", "type": {"type": "map", "values": "int"}},
    {"name": "registrationDate", "type": {"type": "long", "logicalType": "timestamp-millis"}}]}

KAT_ADDRESSES = {"type": "record", "name": "User", "namespace": "com.example", "fields": [
    {"name": "firstName", "type": "string"},
    {"name": "lastName", "type": "string"},
    {"name": "age", "type": "int"},
    {"name": "addresses", "type": {"type": "array", "items": {"type": "record", "name": "Address", "fields": [
        {"name": "street", "type": "string"}, {"name": "city", "type": "string"},
        {"name": "zipCode", "type": "string"}]}}},
    {"name": "email", "type": ["null", "string"], "default": "null"}]}

KAT_ENUM = {"type": "record", "name": "test", "fields": [
    {"name": "a", "type": "long", "default": 42},
    {"name": "b", "type": "string"},
    {"name": "c", "type": {"type": "enum", "name": "suit",
                           "symbols": ["diamonds", "spades", "clubs", "hearts"]}, "default": "spades"}]}

# Round 6: the same record as a production event stream carries it: `created_at` a nullable
# timestamp-micros at today's epoch (an 8-byte varint behind a branch byte), an `int` of epoch seconds (5 bytes), a
# snowflake id (9 bytes), a mostly-null free-text column whose tail is > 8 KiB.  generate_avro.py's columns stay as they are.
FULL_REALISTIC = {"type": "record", "name": "UserEvent", "fields": _USER_COMMON + [
    {"name": "status", "type": ["null", "string", "int", "boolean"], "default": None},
    {"name": "created_at", "type": ["null", {"type": "long", "logicalType": "timestamp-micros"}], "default": None},
    {"name": "class", "type": {"type": "enum", "name": "enum_col", "symbols": ["A", "B", "C"]}},
    {"name": "login_ts", "type": "int"},
    {"name": "event_id", "type": "long"},
    {"name": "note", "type": ["null", "string"], "default": None},
]}

WIDE_ARRAYS = 12


def wide_schema(ncols: int) -> dict:
    """`ncols` nullable string columns with WIDE_ARRAYS arrays of strings spread between them: ncols + 2 * WIDE_ARRAYS scanned
    counters (100-300-field event schemas are ordinary; the reference has no width limit,
    fast_decode.rs:342-370)."""
    fields = []
    every = max(ncols // WIDE_ARRAYS, 1)
    na = 0
    for i in range(ncols):
        fields.append({"name": f"c{i}", "type": ["null", "string"], "default": None})
        if (i + 1) % every == 0 and na < WIDE_ARRAYS:
            fields.append({"name": f"a{na}", "type": {"type": "array", "items": "string"}})
            na += 1
    while na < WIDE_ARRAYS:
        fields.append({"name": f"a{na}", "type": {"type": "array", "items": "string"}})
        na += 1
    return {"type": "record", "name": f"Wide{ncols}", "fields": fields}


WIDE_COLS = (97, 200, 400)

SCHEMAS = {k: json.dumps(v) for k, v in {
    "full": FULL, "flat4": FLAT4, "cfg3": CFG3,
    "flat_primitives": FLAT_PRIMITIVES, "nullable_primitives": NULLABLE_PRIMITIVES,
    "nested_struct": NESTED_STRUCT, "array_and_map": ARRAY_AND_MAP,
    "t_nullable": T_NULLABLE, "t_logical": T_LOGICAL, "t_enum": T_ENUM, "t_nested": T_NESTED,
    "t_nullable_nested": T_NULLABLE_NESTED, "t_union": T_UNION, "t_array_str": T_ARRAY_STR,
    "t_array_int": T_ARRAY_INT, "t_map_str": T_MAP_STR,
    "full_realistic": FULL_REALISTIC, "full_realistic_heavy": FULL_REALISTIC, "full_realistic_nogiant": FULL_REALISTIC, "full_skewed": FULL,
    **{f"wide{n}": wide_schema(n) for n in WIDE_COLS},
    "kat_user": KAT_USER, "kat_userdata": KAT_USERDATA, "kat_addresses": KAT_ADDRESSES, "kat_enum": KAT_ENUM,
}.items()}
