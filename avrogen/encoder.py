"""Avro 1.11 binary encoder for Python values (test-only; stands in for
``apache_avro::to_avro_datum`` / ``fastavro.schemaless_writer`` that the
reference's tests and scripts use to build their inputs:
ruhvro/src/fast_decode.rs:935-943, scripts/generate_avro.py:64-70).

Values: None, bool, int, float, str, dict (record / map), list (array / list
of (key, value) pairs for maps with a chosen wire order).  A union value may be
given explicitly as ``Branch(idx, value)``; otherwise the first variant that
fits the Python type is used (fastavro's rule).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Any

from oracle.avro_schema import AvroSchema  # schema *model* only (a plain dataclass tree)


@dataclass
class Branch:
    idx: int
    value: Any = None


@dataclass
class Blocks:
    """Explicit array/map block structure: list of (items, with_byte_size)."""
    blocks: list


@dataclass
class Unscaled:
    """A decimal's unscaled integer value (the wire carries its big-endian two's complement bytes).  `pad` extra sign
    bytes make a non-minimal encoding (legal on a bytes base)."""
    value: int
    pad: int = 0


@dataclass
class Dur:
    """An Avro duration as written: months, days, milliseconds (three little-endian u32 on a fixed(12))."""
    months: int
    days: int
    millis: int


@dataclass
class UuidText:
    """A uuid as the writer puts it on the wire: text for a string base, 16 raw bytes for a fixed(16) base."""
    text: object


def zigzag(n: int) -> bytes:
    u = ((n << 1) ^ (n >> 63)) & 0xFFFFFFFFFFFFFFFF
    out = bytearray()
    while True:
        b = u & 0x7F
        u >>= 7
        if u:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fits(s: AvroSchema, v) -> bool:
    k = s.kind
    if k == "null":
        return v is None
    if k == "boolean":
        return isinstance(v, bool)
    if k in ("int", "long", "date", "timestamp-millis", "timestamp-micros", "time-millis", "time-micros"):
        return isinstance(v, int) and not isinstance(v, bool)
    if k in ("bytes", "fixed"):
        return isinstance(v, (bytes, bytearray))
    if k == "decimal":
        return isinstance(v, Unscaled)
    if k == "uuid":
        return isinstance(v, UuidText)
    if k == "duration":
        return isinstance(v, Dur)
    if k in ("float", "double"):
        return isinstance(v, float)
    if k in ("string", "enum"):
        return isinstance(v, str)
    if k == "record":
        return isinstance(v, dict)
    if k == "map":
        return isinstance(v, (dict, Blocks)) or (isinstance(v, list) and all(isinstance(x, tuple) for x in v))
    if k == "array":
        return isinstance(v, (list, Blocks))
    return False


def encode(s: AvroSchema, v, out: bytearray) -> None:
    k = s.kind
    if k == "null":
        return
    if k == "boolean":
        out.append(1 if v else 0)
    elif k in ("int", "long", "date", "timestamp-millis", "timestamp-micros", "time-millis", "time-micros"):
        out += zigzag(int(v))
    elif k == "bytes":
        out += zigzag(len(v))
        out += bytes(v)
    elif k == "fixed":
        assert len(v) == s.size, (len(v), s.size)
        out += bytes(v)
    elif k == "decimal":
        u = v.value
        if s.items.kind == "fixed":
            out += u.to_bytes(s.items.size, "big", signed=True)
        else:
            n = 1
            while True:
                try:
                    raw = u.to_bytes(n, "big", signed=True)
                    break
                except OverflowError:
                    n += 1
            raw = (b"\xff" if u < 0 else b"\x00") * v.pad + raw
            out += zigzag(len(raw))
            out += raw
    elif k == "duration":
        out += struct.pack("<III", v.months, v.days, v.millis)
    elif k == "uuid":
        if s.items.kind == "fixed":
            assert isinstance(v.text, (bytes, bytearray)) and len(v.text) == 16
            out += bytes(v.text)
        else:
            b = v.text if isinstance(v.text, bytes) else v.text.encode()
            out += zigzag(len(b))
            out += b
    elif k == "float":
        out += struct.pack("<f", v)
    elif k == "double":
        out += struct.pack("<d", v)
    elif k == "string":
        b = v if isinstance(v, bytes) else v.encode()
        out += zigzag(len(b))
        out += b
    elif k == "enum":
        out += zigzag(v if isinstance(v, int) else s.symbols.index(v))
    elif k == "record":
        for f in s.fields:
            encode(f.schema, v[f.name], out)
    elif k == "union":
        if isinstance(v, Branch):
            out += zigzag(v.idx)
            if 0 <= v.idx < len(s.variants):
                encode(s.variants[v.idx], v.value, out)
            return
        for i, var in enumerate(s.variants):
            if _fits(var, v):
                out += zigzag(i)
                encode(var, v, out)
                return
        raise ValueError(f"value {v!r} fits no union variant")
    elif k in ("array", "map"):
        if isinstance(v, Blocks):
            blocks = v.blocks
        else:
            items = list(v.items()) if isinstance(v, dict) else list(v)
            blocks = [(items, False)] if items else []
        for items, with_size in blocks:
            body = bytearray()
            for it in items:
                if k == "map":
                    kb = it[0].encode()
                    body += zigzag(len(kb))
                    body += kb
                    encode(s.items, it[1], body)
                else:
                    encode(s.items, it, body)
            if with_size:
                out += zigzag(-len(items))
                out += zigzag(len(body))
            else:
                out += zigzag(len(items))
            out += body
        out += b"\x00"
    else:
        raise ValueError(f"cannot encode kind {k}")


def to_datum(s: AvroSchema, v) -> bytes:
    out = bytearray()
    encode(s, v, out)
    return bytes(out)
