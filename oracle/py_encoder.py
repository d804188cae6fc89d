"""ORACLE (test infrastructure, not product code) -- Arrow -> Avro encoder.

Pure-Python restatement of the reference's fast encode path, SURVEY.md section 8(f) N1 -- the checker of the GPU
path behind pyruhvro_amd.serialize_record_batch (rh_encode; tests/test_gpu_encode.py):

  ruhvro/src/fast_encode.rs:27-53     serialize_chunk: one datum per row into a BinaryArray
  ruhvro/src/fast_encode.rs:151-385   encoder construction: columns matched BY NAME, 2-variant null unions
                                      collapsed to Nullable*, N-variant unions read sparse type_ids
  ruhvro/src/fast_encode.rs:387-580   per-row write: non-nullable leaves write value(row) whatever the
                                      validity bit says; lists/maps are ONE block (count, items, 0; empty = 0)
  ruhvro/src/fast_encode.rs:583-599   write_zigzag_long / write_string
  ruhvro/src/serialize.rs:15-67       clamp_chunks, slice_struct (floor division, remainder in the last chunk)

Pinned by: decode(encode(batch)) == batch against the decode oracle, encode(decode(records)) == records for the
single-block wire form the reference writes, the reference's golden datums (tests/golden/reference_vectors.json)
re-encoded byte for byte, and the error texts of fast_encode.rs:173-177, 541, 575-577.
Only tests/ may import this module.
"""
from __future__ import annotations

from typing import Callable, List

import numpy as np
import pyarrow as pa

from . import avro_schema as S


class EncodeError(ValueError):
    pass


def zigzag(v: int) -> bytes:
    """write_zigzag_long, fast_encode.rs:583-591 (v is an i64)."""
    v = int(v)
    zz = ((v << 1) ^ (v >> 63)) & 0xFFFFFFFFFFFFFFFF
    out = bytearray()
    while zz & ~0x7F:
        out.append((zz & 0x7F) | 0x80)
        zz >>= 7
    out.append(zz)
    return bytes(out)


def _raw_values(arr: pa.Array, dtype) -> np.ndarray:
    """values buffer of a primitive array as written (value(row) ignores validity), honouring arr.offset."""
    buf = arr.buffers()[1]
    n = len(arr)
    if buf is None:
        return np.zeros(n, dtype=dtype)
    return np.frombuffer(buf, dtype=dtype, count=arr.offset + n)[arr.offset:]


def _bool_values(arr: pa.Array) -> np.ndarray:
    buf = arr.buffers()[1]
    n = len(arr)
    if buf is None:
        return np.zeros(n, dtype=bool)
    bits = np.unpackbits(np.frombuffer(buf, dtype=np.uint8), bitorder="little")
    return bits[arr.offset: arr.offset + n].astype(bool)


def _is_null(arr: pa.Array) -> np.ndarray:
    n = len(arr)
    if arr.null_count == 0 or arr.buffers()[0] is None:
        return np.zeros(n, dtype=bool)
    bits = np.unpackbits(np.frombuffer(arr.buffers()[0], dtype=np.uint8), bitorder="little")
    return ~bits[arr.offset: arr.offset + n].astype(bool)


def _string_values(arr: pa.Array) -> List[bytes]:
    if not pa.types.is_string(arr.type):
        raise EncodeError("fast_encode: arrow array downcast failed")
    n = len(arr)
    offs = np.frombuffer(arr.buffers()[1], dtype=np.int32, count=arr.offset + n + 1)[arr.offset:]
    data = arr.buffers()[2].to_pybytes() if arr.buffers()[2] is not None else b""
    return [data[int(offs[i]): int(offs[i + 1])] for i in range(n)]


_PRIM = {
    "int": (pa.int32(), np.int32), "long": (pa.int64(), np.int64), "date": (pa.date32(), np.int32),
    "timestamp-millis": (pa.timestamp("ms"), np.int64), "timestamp-micros": (pa.timestamp("us"), np.int64),
    "float": (pa.float32(), np.uint32), "double": (pa.float64(), np.uint64),
}

Writer = Callable[[bytearray, int], None]


def _leaf_writer(s: S.AvroSchema, arr: pa.Array) -> Writer:
    k = s.kind
    if k in _PRIM:
        typ, dt = _PRIM[k]
        if arr.type != typ:
            raise EncodeError("fast_encode: arrow array downcast failed")
        vals = _raw_values(arr, dt)
        if k in ("float", "double"):
            width = 4 if k == "float" else 8
            return lambda out, row: out.extend(int(vals[row]).to_bytes(width, "little"))
        return lambda out, row: out.extend(zigzag(int(vals[row])))
    if k == "boolean":
        if not pa.types.is_boolean(arr.type):
            raise EncodeError("fast_encode: arrow array downcast failed")
        vals = _bool_values(arr)
        return lambda out, row: out.append(1 if vals[row] else 0)
    if k == "string":
        strs = _string_values(arr)
        return lambda out, row: (out.extend(zigzag(len(strs[row]))), out.extend(strs[row]))[0]
    if k == "enum":
        strs = _string_values(arr)
        index = {sym.encode(): i for i, sym in enumerate(s.symbols)}

        def w(out, row):
            i = index.get(strs[row])
            if i is None:
                raise EncodeError(f"fast_encode: enum symbol '{strs[row].decode('utf-8', 'replace')}' not in schema")
            out.extend(zigzag(i))
        return w
    if k == "null":
        return lambda out, row: None
    w = _n4_writer(s, arr)
    if w is not None:
        return w
    raise EncodeError(f"fast_encode: unsupported schema: {k}")


def _fixed_width_values(arr: pa.Array, width: int) -> List[bytes]:
    """rows of a FixedSizeBinary / Decimal128 values buffer as written (value(row) ignores validity)."""
    n = len(arr)
    buf = arr.buffers()[1]
    if buf is None:
        return [b"\0" * width] * n
    raw = buf.to_pybytes()
    return [raw[(arr.offset + i) * width: (arr.offset + i + 1) * width] for i in range(n)]


def _n4_writer(s: S.AvroSchema, arr: pa.Array):
    """SURVEY 8(f) N4, the encode side (beyond the reference: fast_encode::is_supported is false for these types, and
    so is the decode side's).  The wire forms are the Avro 1.11 specification's -- the mirror of py_walker._decode_n4:
      bytes            varint length + the bytes
      fixed(N)         the N bytes
      decimal / bytes  varint length + the unscaled value as MINIMAL big-endian two's complement (what writers emit:
                       BigInt::to_signed_bytes_be), 1..16 bytes
      decimal / fixed  the unscaled value as N big-endian bytes (its low N bytes: sign-extended when N > its width)
      uuid / string    varint 36 + lower-case 8-4-4-4-12 hex text of the 16 bytes
      uuid / fixed(16) the 16 bytes
      time-millis / time-micros   zig-zag int / long
      duration         months = 0, days = min(v // 86 400 000, 2^32-1), milliseconds = the rest, three little-endian u32
                       (a split py_walker's sum inverts for every value it produces); a negative count or one beyond
                       2^32-1 days + 2^32-1 ms has no wire form"""
    k = s.kind
    if k == "duration":
        if arr.type != pa.duration("ms"):
            raise EncodeError("fast_encode: arrow array downcast failed")
        vals = _raw_values(arr, np.int64)

        def wd(out, row):
            v = int(vals[row])
            days = min(v // 86_400_000, 0xFFFFFFFF)
            ms = v - days * 86_400_000
            if v < 0 or ms > 0xFFFFFFFF:
                raise EncodeError(f"duration value at row {row} has no Avro duration form (negative, or beyond 2^32-1 days + 2^32-1 ms)")
            out.extend((0).to_bytes(4, "little") + days.to_bytes(4, "little") + ms.to_bytes(4, "little"))
        return wd
    if k in ("time-millis", "time-micros"):
        typ, dt = (pa.time32("ms"), np.int32) if k == "time-millis" else (pa.time64("us"), np.int64)
        if arr.type != typ:
            raise EncodeError("fast_encode: arrow array downcast failed")
        vals = _raw_values(arr, dt)
        return lambda out, row: out.extend(zigzag(int(vals[row])))
    if k == "bytes":
        if not pa.types.is_binary(arr.type):
            raise EncodeError("fast_encode: arrow array downcast failed")
        n = len(arr)
        offs = np.frombuffer(arr.buffers()[1], dtype=np.int32, count=arr.offset + n + 1)[arr.offset:]
        data = arr.buffers()[2].to_pybytes() if arr.buffers()[2] is not None else b""
        vals = [data[int(offs[i]): int(offs[i + 1])] for i in range(n)]
        return lambda out, row: (out.extend(zigzag(len(vals[row]))), out.extend(vals[row]))[0]
    if k == "fixed" or (k == "uuid" and s.items is not None and s.items.kind == "fixed"):
        width = s.size if k == "fixed" else 16
        if not pa.types.is_fixed_size_binary(arr.type) or arr.type.byte_width != width:
            raise EncodeError("fast_encode: arrow array downcast failed")
        vals = _fixed_width_values(arr, width)
        return lambda out, row: out.extend(vals[row])
    if k == "uuid":
        if not pa.types.is_fixed_size_binary(arr.type) or arr.type.byte_width != 16:
            raise EncodeError("fast_encode: arrow array downcast failed")
        vals = _fixed_width_values(arr, 16)

        def wu(out, row):
            h = vals[row].hex()
            txt = f"{h[0:8]}-{h[8:12]}-{h[12:16]}-{h[16:20]}-{h[20:32]}".encode()
            out.extend(zigzag(36))
            out.extend(txt)
        return wu
    if k == "decimal":
        if not pa.types.is_decimal128(arr.type) or arr.type.precision != s.precision or arr.type.scale != s.scale:
            raise EncodeError("fast_encode: arrow array downcast failed")
        vals = [int.from_bytes(b, "little", signed=True) for b in _fixed_width_values(arr, 16)]
        if s.items is not None and s.items.kind == "fixed":
            size = s.items.size

            def wf(out, row):
                out.extend((vals[row] & ((1 << (8 * size)) - 1)).to_bytes(size, "big"))
            return wf

        def wb(out, row):
            v = vals[row]
            nb = max(1, ((v if v >= 0 else ~v).bit_length() + 8) // 8)     # minimal two's complement length
            out.extend(zigzag(nb))
            out.extend(v.to_bytes(nb, "big", signed=True))
        return wb
    return None


def _split_null_union(s: S.AvroSchema):
    if len(s.variants) != 2:
        return None
    a, b = s.variants
    if a.kind == "null":
        return b, True
    if b.kind == "null":
        return a, False
    return None


def _record_writer(s: S.AvroSchema, sa: pa.Array) -> Writer:
    """build_record_encoder, fast_encode.rs:151-185: arrow columns are matched to avro fields by NAME."""
    if not pa.types.is_struct(sa.type):
        raise EncodeError("fast_encode: expected StructArray for record")
    names = [sa.type.field(i).name for i in range(sa.type.num_fields)]
    kids = []
    for f in s.fields:
        if f.name not in names:
            avail = ", ".join('"%s"' % n for n in names)
            raise EncodeError(f"Arrow struct missing column '{f.name}' required by Avro schema. Available columns: [{avail}]")
        child = sa.field(names.index(f.name))          # StructArray.field applies the parent's offset/length
        kids.append(_field_writer(f.schema, child))

    def w(out, row):
        for k in kids:
            k(out, row)
    return w


def _list_writer(s: S.AvroSchema, arr: pa.Array, is_map: bool) -> Writer:
    if is_map:
        if not pa.types.is_map(arr.type):
            raise EncodeError("fast_encode: expected MapArray for map schema")
        keys = _string_values(arr.keys) if pa.types.is_string(arr.keys.type) else None
        if keys is None:
            raise EncodeError("fast_encode: map keys must be StringArray")
        inner = _field_writer(s.items, arr.items)
    else:
        if not pa.types.is_list(arr.type):
            raise EncodeError("fast_encode: expected ListArray for array schema")
        keys = None
        inner = _field_writer(s.items, arr.values)
    n = len(arr)
    offs = np.frombuffer(arr.buffers()[1], dtype=np.int32, count=arr.offset + n + 1)[arr.offset:]

    def w(out, row):      # ListEncoder::write / MapEncoder::write, fast_encode.rs:525-561
        start, end = int(offs[row]), int(offs[row + 1])
        if end > start:
            out.extend(zigzag(end - start))
            for i in range(start, end):
                if keys is not None:
                    out.extend(zigzag(len(keys[i])))
                    out.extend(keys[i])
                inner(out, i)
        out.extend(zigzag(0))
    return w


def _nullable(inner: Writer, is_null: np.ndarray, null_first: bool) -> Writer:
    def w(out, row):      # write_nullable, fast_encode.rs:563-572
        if is_null[row]:
            out.extend(zigzag(0 if null_first else 1))
        else:
            out.extend(zigzag(1 if null_first else 0))
            inner(out, row)
    return w


def _field_writer(s: S.AvroSchema, arr: pa.Array) -> Writer:
    """build_field_encoder / build_union_encoder / build_nullable_encoder, fast_encode.rs:187-358."""
    k = s.kind
    if k == "record":
        return _record_writer(s, arr)          # non-null nested record: the struct's own validity is ignored (214-217)
    if k == "array":
        return _list_writer(s, arr, False)
    if k == "map":
        return _list_writer(s, arr, True)
    if k == "union":
        sp = _split_null_union(s)
        if sp is not None:
            inner, null_first = sp
            if inner.kind in ("null", "union"):
                raise EncodeError(f"fast_encode: unsupported nullable inner type: {inner.kind}")
            return _nullable(_field_writer(inner, arr), _is_null(arr), null_first)
        if not pa.types.is_union(arr.type):
            raise EncodeError("fast_encode: expected UnionArray for multi-variant union")
        type_ids = np.frombuffer(arr.buffers()[1], dtype=np.int8, count=arr.offset + len(arr))[arr.offset:]
        kids = []
        for i, v in enumerate(s.variants):      # schema_translate emits type ids 0..N-1 in variant order (263-269)
            kids.append(_field_writer(v, arr.field(i)))     # sparse: pyarrow hands the child back with the union's offset / length applied

        def w(out, row):    # UnionEncoder::write, fast_encode.rs:507-521
            t = int(type_ids[row])
            if t < 0 or t >= len(kids):
                raise EncodeError(f"fast_encode: union type_id {t} out of range")
            out.extend(zigzag(t))
            kids[t](out, row)
        return w
    return _leaf_writer(s, arr)


def serialize_chunk(schema: S.AvroSchema, sa: pa.Array) -> pa.Array:
    """fast_encode.rs:27-53."""
    if schema.kind != "record":
        raise EncodeError("fast_encode: top-level schema must be a Record")
    w = _record_writer(schema, sa)
    rows = []
    for i in range(len(sa)):
        out = bytearray()
        w(out, i)
        rows.append(bytes(out))
    return pa.array(rows, type=pa.binary())


def serialize_record_batch(rb: pa.RecordBatch, schema_json: str, num_chunks: int, extended: bool = False) -> List[pa.Array]:
    """serialize.rs:38-67: k = clamp(num_chunks, 1, max(rows, 1)); chunk i = rows [i*sz, (i+1)*sz), last takes the rest."""
    s = S.parse_schema(schema_json, resolve_refs=extended)
    if not (S.is_supported_extended(s) if extended else S.is_supported(s)):
        raise EncodeError("schema is outside the fast encode path (fast_encode::is_supported == false)")
    sa = rb.to_struct_array()
    n = len(sa)
    k = min(max(int(num_chunks), 1), max(n, 1))
    sz = n // k
    out = []
    for i in range(k):
        part = sa.slice(i * sz, n - i * sz) if i == k - 1 else sa.slice(i * sz, sz)
        out.append(serialize_chunk(s, part))
    return out
