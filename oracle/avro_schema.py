"""ORACLE (test infrastructure, not product code) -- Avro schema front-end.

CPU restatement of the two schema steps that sit in front of the reference's
direct-decode hot path:

  * ``apache_avro::Schema::parse_str`` (third-party crate apache-avro 0.21.0,
    pinned in /root/reference/Cargo.lock:62-63; not vendored) -- only the
    subset the fast path can reach is restated, following the Avro 1.11 spec
    and the call site ruhvro/src/deserialize.rs:18-20.
  * ``to_arrow_schema`` / ``schema_to_field_with_props`` /
    ``default_field_name`` / ``external_props``
    (ruhvro/src/schema_translate.rs:19-37, 43-153, 155-220, 222-266).
  * the decoder-tree construction of ruhvro/src/fast_decode.rs:176-414
    (``make_decoder`` & friends, ``split_null_union``) and the gate
    ``is_supported`` (fast_decode.rs:38-61).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The shipped engine (pyruhvro_amd) never does.

The pyarrow types built here are in the form pyarrow shows AFTER importing
what arrow-rs exports over the C Data Interface (pyarrow renames map children
to key/value and drops the entries-struct nullability on import), because
that is what a pyruhvro user observes.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field as dc_field
from typing import Dict, List, Optional, Tuple

import pyarrow as pa

PRIMITIVES = ("null", "boolean", "int", "long", "float", "double", "bytes", "string")

# kinds the fast path accepts as leaves (fast_decode.rs:44-54)
FAST_LEAVES = {
    "null", "boolean", "int", "long", "float", "double", "string",
    "date", "timestamp-millis", "timestamp-micros", "enum",
}


class SchemaError(ValueError):
    pass


@dataclass
class AvroField:
    name: str
    schema: "AvroSchema"
    doc: Optional[str] = None


@dataclass
class AvroSchema:
    kind: str  # primitive name, logical name, record, enum, array, map, union, fixed, ref
    # named types
    name: Optional[str] = None
    namespace: Optional[str] = None
    doc: Optional[str] = None
    aliases: Optional[List[str]] = None
    fields: List[AvroField] = dc_field(default_factory=list)   # record
    symbols: List[str] = dc_field(default_factory=list)        # enum
    items: Optional["AvroSchema"] = None                       # array items / map values
    variants: List["AvroSchema"] = dc_field(default_factory=list)  # union
    size: int = 0                                              # fixed
    precision: int = 0
    scale: int = 0

    def fullname(self) -> str:
        if self.namespace:
            return f"{self.namespace}.{self.name}"
        return self.name or ""


def _split_name(name: str, explicit_ns: Optional[str], enclosing_ns: Optional[str]) -> Tuple[str, Optional[str]]:
    # apache-avro Name::parse: a dotted name carries its own namespace; else
    # the "namespace" attribute; else the enclosing namespace.
    if "." in name:
        ns, _, simple = name.rpartition(".")
        return simple, ns or None
    ns = explicit_ns if explicit_ns is not None else enclosing_ns
    if ns == "":
        ns = None
    return name, ns


class _Parser:
    def __init__(self, resolve_refs: bool = False) -> None:
        self.named: Dict[str, str] = {}
        # resolve_refs (beyond the reference, which stops at Schema::Ref): a reference to a named type IS that type
        # (Avro specification) -- substitute a copy of its completed definition; a reference to a type that is still
        # being defined is a recursive type, which no Arrow schema can express
        self.resolve_refs = resolve_refs
        self.defs: Dict[str, AvroSchema] = {}

    def parse(self, j, enclosing_ns: Optional[str]) -> AvroSchema:
        if isinstance(j, str):
            return self._parse_name(j, enclosing_ns)
        if isinstance(j, list):
            return self._parse_union(j, enclosing_ns)
        if isinstance(j, dict):
            return self._parse_complex(j, enclosing_ns)
        raise SchemaError("Must be a JSON string, object or array")

    def _parse_name(self, s: str, enclosing_ns: Optional[str]) -> AvroSchema:
        if s in PRIMITIVES:
            return AvroSchema(kind=s)
        simple, ns = _split_name(s, None, enclosing_ns)
        full = f"{ns}.{simple}" if ns else simple
        if full in self.named:
            if self.resolve_refs:
                if full not in self.defs:
                    raise SchemaError(f"recursive named type {full}: a type that contains itself has no Arrow schema")
                import copy
                return copy.deepcopy(self.defs[full])
            # apache-avro turns a repeated named type into Schema::Ref, which
            # neither is_supported (fast_decode.rs:59) nor schema_translate
            # (schema_translate.rs:51, todo!()) handles.
            return AvroSchema(kind="ref", name=simple, namespace=ns)
        raise SchemaError(f"Unknown type: {s}")

    def _parse_union(self, arr: list, enclosing_ns: Optional[str]) -> AvroSchema:
        variants = []
        seen = set()
        for v in arr:
            if isinstance(v, list):
                raise SchemaError("Unions may not directly contain a union")
            sch = self.parse(v, enclosing_ns)
            key = sch.fullname() if sch.kind in ("record", "enum", "fixed", "ref") else sch.kind
            if key in seen:
                raise SchemaError("Unions cannot contain duplicate types")
            seen.add(key)
            variants.append(sch)
        return AvroSchema(kind="union", variants=variants)

    def _parse_complex(self, d: dict, enclosing_ns: Optional[str]) -> AvroSchema:
        lt = d.get("logicalType")
        t = d.get("type")
        if isinstance(lt, str) and lt in _LOGICAL_BASE:
            base = self.parse(t, enclosing_ns) if not isinstance(t, str) or t not in ("record", "enum", "array", "map", "fixed") \
                else self._parse_complex({k: v for k, v in d.items() if k != "logicalType"}, enclosing_ns)
            if base.kind in _LOGICAL_BASE[lt]:
                if lt == "decimal":
                    # precision required and positive, scale <= precision, a fixed base must be able to hold the
                    # precision; otherwise apache-avro warns and keeps the underlying type
                    pr, sc = d.get("precision"), d.get("scale", 0)
                    ok = isinstance(pr, int) and not isinstance(pr, bool) and pr >= 1 and isinstance(sc, int) \
                        and not isinstance(sc, bool) and 0 <= sc <= pr
                    if ok and base.kind == "fixed":
                        import math
                        ok = pr <= math.floor((8 * base.size - 1) * math.log10(2))
                    if not ok:
                        return base
                    return AvroSchema(kind="decimal", precision=pr, scale=sc, items=base)
                if lt == "uuid":
                    if base.kind == "fixed" and base.size != 16:
                        return base
                    return AvroSchema(kind="uuid", items=base)
                if lt == "duration":
                    if base.size != 12:         # months / days / milliseconds, 4 bytes each: anything else stays a fixed
                        return base
                    return AvroSchema(kind="duration", items=base)
                return AvroSchema(kind=lt)
            return base  # apache-avro warns and keeps the underlying type
        if isinstance(t, str):
            if t == "record":
                return self._parse_record(d, enclosing_ns)
            if t == "enum":
                return self._parse_enum(d, enclosing_ns)
            if t == "array":
                if "items" not in d:
                    raise SchemaError("No `items` in array")
                return AvroSchema(kind="array", items=self.parse(d["items"], enclosing_ns))
            if t == "map":
                if "values" not in d:
                    raise SchemaError("No `values` in map")
                return AvroSchema(kind="map", items=self.parse(d["values"], enclosing_ns))
            if t == "fixed":
                simple, ns = self._register(d, enclosing_ns)
                sz = d.get("size")
                if not isinstance(sz, int) or isinstance(sz, bool) or sz < 0:
                    raise SchemaError("No `size` in fixed")
                fx = AvroSchema(kind="fixed", name=simple, namespace=ns, size=sz,
                                doc=d.get("doc"), aliases=d.get("aliases"))
                self.defs[fx.fullname()] = fx
                return fx
            return self._parse_name(t, enclosing_ns)
        if isinstance(t, dict):
            return self._parse_complex(t, enclosing_ns)
        if isinstance(t, list):
            return self._parse_union(t, enclosing_ns)
        raise SchemaError("No `type` in complex type")

    def _register(self, d: dict, enclosing_ns: Optional[str]) -> Tuple[str, Optional[str]]:
        name = d.get("name")
        if not isinstance(name, str) or not name:
            raise SchemaError("No `name` field")
        simple, ns = _split_name(name, d.get("namespace"), enclosing_ns)
        full = f"{ns}.{simple}" if ns else simple
        if full in self.named:
            raise SchemaError(f"Two schemas with the same fullname were given: {full}")
        self.named[full] = d["type"]
        return simple, ns

    def _parse_record(self, d: dict, enclosing_ns: Optional[str]) -> AvroSchema:
        simple, ns = self._register(d, enclosing_ns)
        if not isinstance(d.get("fields"), list):
            raise SchemaError("No `fields` in record")
        fields = []
        seen = set()
        for f in d["fields"]:
            if not isinstance(f, dict) or not isinstance(f.get("name"), str):
                raise SchemaError("No `name` in record field")
            if "type" not in f:
                raise SchemaError("No `type` in record field")
            if f["name"] in seen:
                raise SchemaError(f"Duplicate field name {f['name']}")
            seen.add(f["name"])
            fields.append(AvroField(name=f["name"], schema=self.parse(f["type"], ns),
                                    doc=f.get("doc") if isinstance(f.get("doc"), str) else None))
        rec = AvroSchema(kind="record", name=simple, namespace=ns, fields=fields,
                         doc=d.get("doc") if isinstance(d.get("doc"), str) else None,
                         aliases=d.get("aliases") if isinstance(d.get("aliases"), list) else None)
        self.defs[rec.fullname()] = rec
        return rec

    def _parse_enum(self, d: dict, enclosing_ns: Optional[str]) -> AvroSchema:
        simple, ns = self._register(d, enclosing_ns)
        syms = d.get("symbols")
        if not isinstance(syms, list) or not all(isinstance(s, str) for s in syms):
            raise SchemaError("No `symbols` field in enum")
        if len(set(syms)) != len(syms):
            raise SchemaError("Duplicate enum symbol")
        en = AvroSchema(kind="enum", name=simple, namespace=ns, symbols=list(syms),
                        doc=d.get("doc") if isinstance(d.get("doc"), str) else None,
                        aliases=d.get("aliases") if isinstance(d.get("aliases"), list) else None)
        self.defs[en.fullname()] = en
        return en


_LOGICAL_BASE = {
    "date": ("int",),
    "time-millis": ("int",),
    "time-micros": ("long",),
    "timestamp-millis": ("long",),
    "timestamp-micros": ("long",),
    "timestamp-nanos": ("long",),
    "local-timestamp-millis": ("long",),
    "local-timestamp-micros": ("long",),
    "local-timestamp-nanos": ("long",),
    "uuid": ("string", "fixed"),
    "decimal": ("bytes", "fixed"),
    "duration": ("fixed",),
}


def parse_schema(schema_json: str, resolve_refs: bool = False) -> AvroSchema:
    """deserialize.rs:18-20 -> apache_avro::Schema::parse_str.  resolve_refs: named-type references are replaced by the
    definition they name (the engine's behaviour, beyond the reference: schema_translate.rs:51 is a todo!())."""
    try:
        j = json.loads(schema_json)
    except json.JSONDecodeError as e:
        raise SchemaError(f"Failed to parse schema from JSON: {e}") from None
    return _Parser(resolve_refs).parse(j, None)


# ---------------------------------------------------------------------------
# fast_decode.rs:38-61
# ---------------------------------------------------------------------------
def is_supported(s: AvroSchema) -> bool:
    return s.kind == "record" and _is_supported_inner(s)


def _is_supported_inner(s: AvroSchema, extra=frozenset()) -> bool:
    if s.kind in FAST_LEAVES:
        return True
    if s.kind in extra:
        return _n4_ok(s)
    if s.kind == "record":
        return all(_is_supported_inner(f.schema, extra) for f in s.fields)
    if s.kind == "union":
        return all(_is_supported_inner(v, extra) for v in s.variants)
    if s.kind in ("array", "map"):
        return _is_supported_inner(s.items, extra)
    return False


# ---------------------------------------------------------------------------
# SURVEY 8(f) N4 -- BEYOND THE REFERENCE.  The types below are translated to Arrow by the reference
# (schema_translate.rs:58,133-140, restated in schema_to_field) but rejected by its direct-decode gate
# (fast_decode.rs:59), and its Value-tree fallback has no column builder for them (complex.rs:414-431:
# unimplemented!()).  The GPU path decodes them; what "correct" means is therefore the Avro 1.11 specification,
# restated in py_walker under `extended=True`.  PARITY UNPINNED BY THE REFERENCE for these types.
# ---------------------------------------------------------------------------
N4_LEAVES = {"bytes", "fixed", "decimal", "uuid", "time-millis", "time-micros", "duration"}


def is_supported_extended(s: AvroSchema) -> bool:
    return s.kind == "record" and _is_supported_inner(s, N4_LEAVES)


def _n4_ok(s: AvroSchema) -> bool:
    if s.kind == "decimal":
        return s.precision <= 38 and (s.items.kind != "fixed" or s.items.size <= 16)
    if s.kind == "fixed":
        return s.size <= (1 << 20)
    return True


# ---------------------------------------------------------------------------
# schema_translate.rs
# ---------------------------------------------------------------------------
def _default_field_name(dt: pa.DataType) -> str:
    """schema_translate.rs:155-220 (only the types reachable from the fast path)."""
    if pa.types.is_null(dt):
        return "null"
    if pa.types.is_boolean(dt):
        return "bit"
    if pa.types.is_int32(dt):
        return "int"
    if pa.types.is_int64(dt):
        return "bigint"
    if pa.types.is_float32(dt):
        return "float4"
    if pa.types.is_float64(dt):
        return "float8"
    if pa.types.is_date32(dt):
        return "dateday"
    if pa.types.is_timestamp(dt):
        return {"ms": "timestampmilli", "us": "timestampmicro"}[dt.unit]
    if pa.types.is_string(dt):
        return "varchar"
    if pa.types.is_fixed_size_binary(dt):
        return "fixedsizebinary"
    if pa.types.is_binary(dt):
        return "varbinary"
    if pa.types.is_decimal(dt):
        return "decimal"
    if pa.types.is_time(dt):
        return {"ms": "timemilli", "us": "timemicro"}[dt.unit]
    if pa.types.is_duration(dt):
        return "duration"               # schema_translate.rs:195
    if pa.types.is_list(dt):
        return "list"
    if pa.types.is_struct(dt):
        return "struct"
    if pa.types.is_union(dt):
        return "union"
    if pa.types.is_map(dt):
        # schema_translate.rs:212 unimplemented!("Map support not implemented")
        raise SchemaError("Map support not implemented (map as an anonymous union variant)")
    raise SchemaError("data type missing default name")


def _aliased(alias: str, namespace: Optional[str]) -> str:
    """schema_translate.rs:269-280."""
    if "." in alias:
        return alias
    if namespace:
        return f"{namespace}.{alias}"
    return alias


def _external_props(s: AvroSchema) -> Dict[str, str]:
    """schema_translate.rs:222-266."""
    props: Dict[str, str] = {}
    if s.kind in ("record", "enum", "fixed"):
        if s.doc is not None:
            props["avro::doc"] = s.doc
        if s.aliases is not None:
            props["avro::aliases"] = "[" + ",".join(_aliased(a, s.namespace) for a in s.aliases) + "]"
    return props


def schema_to_field(s: AvroSchema, name: Optional[str], nullable: bool,
                    props: Optional[Dict[str, str]]) -> pa.Field:
    """schema_translate.rs:43-153."""
    k = s.kind
    if k == "ref":
        raise SchemaError("Add support for AvroSchema::Ref")  # todo!() at schema_translate.rs:51
    if k == "null":
        dt = pa.null()
    elif k == "boolean":
        dt = pa.bool_()
    elif k == "int":
        dt = pa.int32()
    elif k == "long":
        dt = pa.int64()
    elif k == "float":
        dt = pa.float32()
    elif k == "double":
        dt = pa.float64()
    elif k == "string":
        dt = pa.string()
    elif k == "date":
        dt = pa.date32()
    elif k == "timestamp-millis":
        dt = pa.timestamp("ms")
    elif k == "timestamp-micros":
        dt = pa.timestamp("us")
    elif k == "bytes":                      # schema_translate.rs:58
        dt = pa.binary()
    elif k == "fixed":                      # :133
        dt = pa.binary(s.size)
    elif k == "decimal":                    # :134-136
        dt = pa.decimal128(s.precision, s.scale)
    elif k == "uuid":                       # :137
        dt = pa.binary(16)
    elif k == "time-millis":                # :139
        dt = pa.time32("ms")
    elif k == "time-micros":                # :140
        dt = pa.time64("us")
    elif k == "duration":               # :143
        dt = pa.duration("ms")
    elif k == "array":
        dt = pa.list_(schema_to_field(s.items, "item", True, None))
    elif k == "map":
        value_field = schema_to_field(s.items, "values", False, None)
        # pyarrow renames the children to key/value when importing (see module doc)
        dt = pa.map_(pa.string(), value_field.with_name("value"))
    elif k == "union":
        has_nullable = any(v.kind == "null" for v in s.variants)
        if has_nullable and len(s.variants) == 2:
            nullable = True
            inner = [v for v in s.variants if v.kind != "null"]
            if not inner:
                raise SchemaError("Avro union contains duplicate null variants")
            dt = schema_to_field(inner[0], None, has_nullable, None).type
        else:
            if has_nullable:
                nullable = True
            fields = [schema_to_field(v, None, True, None) for v in s.variants]
            dt = pa.sparse_union(fields, type_codes=list(range(len(fields))))
    elif k == "record":
        fields = []
        for f in s.fields:
            p = {}
            if f.doc is not None:
                p["avro::doc"] = f.doc
            fields.append(schema_to_field(f.schema, f.name, nullable, p))
        dt = pa.struct(fields)
    elif k == "enum":
        field_name = name if name else s.fullname()
        return pa.field(field_name, pa.string(), nullable)  # early return: no metadata (schema_translate.rs:131)
    else:
        raise SchemaError(f"schema kind {k!r} is outside the direct-decode path")
    fname = name if name is not None else _default_field_name(dt)
    f = pa.field(fname, dt, nullable)
    if props:
        f = f.with_metadata(props)
    return f


def to_arrow_schema(s: AvroSchema) -> pa.Schema:
    """schema_translate.rs:19-37."""
    if s.kind != "record":
        raise SchemaError("fast_decode::decode called on non-record schema")
    return pa.schema([schema_to_field(f.schema, f.name, False, _external_props(f.schema)) for f in s.fields])


# ---------------------------------------------------------------------------
# Decoder tree (fast_decode.rs:73-167, 176-414)
# ---------------------------------------------------------------------------
# node kinds (shared with oracle_walk.c)
K_INT, K_LONG, K_FLOAT, K_DOUBLE, K_BOOL, K_STRING, K_DATE, K_TSMILLI, K_TSMICRO, K_ENUM, \
    K_NULL, K_RECORD, K_UNION, K_LIST, K_MAP = range(15)
# N4 (beyond the reference; py_walker only): the wire forms of the Avro 1.11 specification
K_BYTES, K_FIXED, K_DECIMAL, K_UUID, K_TIMEMILLI, K_TIMEMICRO, K_DURATION = range(15, 22)
_N4_KIND = {"bytes": K_BYTES, "fixed": K_FIXED, "decimal": K_DECIMAL, "uuid": K_UUID,
            "time-millis": K_TIMEMILLI, "time-micros": K_TIMEMICRO, "duration": K_DURATION}

_LEAF_KIND = {
    "int": K_INT, "long": K_LONG, "float": K_FLOAT, "double": K_DOUBLE, "boolean": K_BOOL,
    "string": K_STRING, "date": K_DATE, "timestamp-millis": K_TSMILLI,
    "timestamp-micros": K_TSMICRO, "enum": K_ENUM, "null": K_NULL,
}


@dataclass
class Node:
    """One FieldDecoder.  ``nullable`` <=> the Nullable* variant (2-variant null
    union collapsed, fast_decode.rs:376-378,404-414)."""
    kind: int
    field: pa.Field                 # arrow field schema_translate produced for it
    nullable: bool = False
    null_first: bool = False
    children: List["Node"] = dc_field(default_factory=list)  # record fields / union variants / [item] / [value]
    symbols: List[str] = dc_field(default_factory=list)
    idx: int = -1                   # position in the flattened table
    wire_size: int = -1             # N4: bytes of a fixed base (fixed / decimal / uuid), -1 = length-prefixed


def _split_null_union(s: AvroSchema):
    """fast_decode.rs:404-414."""
    if len(s.variants) != 2:
        return None
    a, b = s.variants
    if a.kind == "null":
        return b, True
    if b.kind == "null":
        return a, False
    return None


def _map_value_field(f: pa.Field) -> pa.Field:
    return f.type.item_field


def make_decoder(s: AvroSchema, f: pa.Field, nullable: bool = False, null_first: bool = False) -> Node:
    """make_decoder / make_nullable_decoder / make_union_decoder (fast_decode.rs:176-402)."""
    k = s.kind
    if k in _LEAF_KIND:
        if nullable and k == "null":
            raise SchemaError("fast_decode: unsupported nullable inner type: Null")
        return Node(kind=_LEAF_KIND[k], field=f, nullable=nullable, null_first=null_first,
                    symbols=list(s.symbols))
    if k in _N4_KIND:               # only reachable through build_tree(extended=True)
        base = s if k == "fixed" else s.items
        wire = base.size if base is not None and base.kind == "fixed" else -1
        return Node(kind=_N4_KIND[k], field=f, nullable=nullable, null_first=null_first, wire_size=wire)
    if k == "record":
        inner = [f.type.field(i) for i in range(f.type.num_fields)]
        if len(inner) != len(s.fields):
            raise SchemaError("fast_decode: avro/arrow field count mismatch")
        return Node(kind=K_RECORD, field=f, nullable=nullable, null_first=null_first,
                    children=[make_decoder(af.schema, ff) for af, ff in zip(s.fields, inner)])
    if k == "array":
        return Node(kind=K_LIST, field=f, nullable=nullable, null_first=null_first,
                    children=[make_decoder(s.items, f.type.value_field)])
    if k == "map":
        return Node(kind=K_MAP, field=f, nullable=nullable, null_first=null_first,
                    children=[make_decoder(s.items, _map_value_field(f))])
    if k == "union":
        if nullable:
            raise SchemaError("fast_decode: unsupported nullable inner type: Union")
        sp = _split_null_union(s)
        if sp is not None:
            return make_decoder(sp[0], f, nullable=True, null_first=sp[1])
        ufields = [f.type.field(i) for i in range(f.type.num_fields)]
        return Node(kind=K_UNION, field=f,
                    children=[make_decoder(v, uf) for v, uf in zip(s.variants, ufields)])
    raise SchemaError(f"fast_decode: unsupported schema in make_decoder: {k}")


def build_tree(s: AvroSchema, extended: bool = False) -> Tuple[pa.Schema, Node]:
    """Top-level record decoder (fast_decode.rs:815-824): non-nullable record
    whose arrow fields are the schema's top-level fields.  ``extended`` also admits the N4 leaf types (see
    is_supported_extended: beyond the reference's gate)."""
    if not (is_supported_extended(s) if extended else is_supported(s)):
        raise SchemaError("schema is outside the direct-decode path (fast_decode::is_supported == false)")
    arrow_schema = to_arrow_schema(s)
    top_field = pa.field("", pa.struct(list(arrow_schema)), False)
    root = make_decoder(s, top_field)
    if not root.children:
        raise SchemaError("RecordDecoder produced a record with 0 fields")
    return arrow_schema, root


def flatten(root: Node) -> List[Node]:
    """Pre-order numbering; children of a node are NOT necessarily contiguous,
    so the C table stores explicit child index lists."""
    out: List[Node] = []

    def rec(n: Node):
        n.idx = len(out)
        out.append(n)
        for c in n.children:
            rec(c)
    rec(root)
    return out
