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

SCHEMAS = {k: json.dumps(v) for k, v in {
    "full": FULL, "flat4": FLAT4, "cfg3": CFG3,
    "flat_primitives": FLAT_PRIMITIVES, "nullable_primitives": NULLABLE_PRIMITIVES,
    "nested_struct": NESTED_STRUCT, "array_and_map": ARRAY_AND_MAP,
    "t_nullable": T_NULLABLE, "t_logical": T_LOGICAL, "t_enum": T_ENUM, "t_nested": T_NESTED,
    "t_nullable_nested": T_NULLABLE_NESTED, "t_union": T_UNION, "t_array_str": T_ARRAY_STR,
    "t_array_int": T_ARRAY_INT, "t_map_str": T_MAP_STR,
    "kat_user": KAT_USER, "kat_userdata": KAT_USERDATA, "kat_addresses": KAT_ADDRESSES, "kat_enum": KAT_ENUM,
}.items()}
