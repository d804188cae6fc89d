"""ORACLE (test infrastructure, not product code) -- pure-Python walker.

Literal small-case restatement of the reference's direct-decode walker,
ruhvro/src/fast_decode.rs:420-922, with arrow-rs 58.3.0 builder semantics
(Cargo.lock:86-87; call sites fast_decode.rs:424-480,505-533,538-565) spelled
out as Python lists.  Slow by design: use it for records counts in the
hundreds; ``c_walker`` is the same algorithm in C for large inputs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
from __future__ import annotations

import struct
from typing import List

from . import avro_schema as S
from .assemble import Buffers, assemble_batch


class DecodeError(ValueError):
    pass


class _Cur:
    __slots__ = ("b", "p", "e")

    def __init__(self, b: bytes):
        self.b = b
        self.p = 0
        self.e = len(b)


def read_byte(c: _Cur) -> int:
    """fast_decode.rs:845-852."""
    if c.p >= c.e:
        raise DecodeError("unexpected end of buffer")
    v = c.b[c.p]
    c.p += 1
    return v


def read_zigzag_long(c: _Cur) -> int:
    """fast_decode.rs:854-869 (bits shifted past 63 are dropped, u64 wrap)."""
    result = 0
    shift = 0
    while True:
        byte = read_byte(c)
        result |= ((byte & 0x7F) << shift) & 0xFFFFFFFFFFFFFFFF
        if byte & 0x80 == 0:
            v = (result >> 1) ^ (-(result & 1) & 0xFFFFFFFFFFFFFFFF)
            return v - (1 << 64) if v >= (1 << 63) else v
        shift += 7
        if shift >= 64:
            raise DecodeError("zigzag varint too long")


def _as_i32(v: int) -> int:
    """`as i32` truncating cast (fast_decode.rs:424,430)."""
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


def read_f32_bits(c: _Cur) -> int:
    """fast_decode.rs:871-879 -- kept as raw bits so NaN payloads survive."""
    if c.e - c.p < 4:
        raise DecodeError("unexpected end of buffer (f32)")
    v = struct.unpack_from("<I", c.b, c.p)[0]
    c.p += 4
    return v


def read_f64_bits(c: _Cur) -> int:
    """fast_decode.rs:881-891."""
    if c.e - c.p < 8:
        raise DecodeError("unexpected end of buffer (f64)")
    v = struct.unpack_from("<Q", c.b, c.p)[0]
    c.p += 8
    return v


def read_bool(c: _Cur) -> bool:
    """fast_decode.rs:893-900."""
    b = read_byte(c)
    if b == 0:
        return False
    if b == 1:
        return True
    raise DecodeError(f"invalid boolean byte: {b}")


def read_string(c: _Cur) -> bytes:
    """fast_decode.rs:902-922 (no UTF-8 validation)."""
    n = read_zigzag_long(c)
    if n < 0:
        raise DecodeError("negative string length")
    if c.e - c.p < n:
        raise DecodeError("unexpected end of buffer (string)")
    s = c.b[c.p:c.p + n]
    c.p += n
    return s


def union_branch(c: _Cur, null_first: bool) -> bool:
    """fast_decode.rs:585-593.  True = Value, False = Null."""
    idx = read_zigzag_long(c)
    if (idx == 0 and null_first) or (idx == 1 and not null_first):
        return False
    if (idx == 1 and null_first) or (idx == 0 and not null_first):
        return True
    raise DecodeError(f"invalid union branch index: {idx}")


def read_block_count(c: _Cur) -> int:
    """fast_decode.rs:689-700."""
    n = read_zigzag_long(c)
    if n < 0:
        read_zigzag_long(c)  # block byte size, ignored
        # `n = -n` on an i64: i64::MIN negates to itself in the release build (wrapping), and `for _ in 0..n` over a
        # negative n is empty -- the block carries no items and is NOT the terminator (the loop only stops at n == 0)
        return n if n == -(1 << 63) else -n
    return n


class _B:
    """Builder state for one node (the arrow-rs builder(s) a FieldDecoder owns)."""

    def __init__(self, node: S.Node):
        self.node = node
        self.length = 0
        self.valid: List[bool] = []     # only meaningful where the node keeps a null buffer
        self.values: List[int] = []     # ints / raw float bits / bools / type_ids
        self.offsets: List[int] = [0]
        self.data = bytearray()
        self.keys = None                # map keys: a STRING-like builder
        self.kids = [_B(ch) for ch in node.children]
        if node.kind == S.K_MAP:
            self.keys = _B(S.Node(kind=S.K_STRING, field=None))

    # ---- leaf appends ------------------------------------------------
    def _append_value(self, v):
        self.valid.append(True)
        self.values.append(v)
        self.length += 1

    def _append_str(self, s: bytes):
        self.valid.append(True)
        self.data += s
        self.offsets.append(len(self.data))
        self.length += 1

    def _decode_value(self, c: _Cur):
        k = self.node.kind
        if k in (S.K_INT, S.K_DATE):
            self._append_value(_as_i32(read_zigzag_long(c)))
        elif k in (S.K_LONG, S.K_TSMILLI, S.K_TSMICRO):
            self._append_value(read_zigzag_long(c))
        elif k == S.K_FLOAT:
            self._append_value(read_f32_bits(c))
        elif k == S.K_DOUBLE:
            self._append_value(read_f64_bits(c))
        elif k == S.K_BOOL:
            self._append_value(read_bool(c))
        elif k == S.K_STRING:
            self._append_str(read_string(c))
        elif k in (S.K_BYTES, S.K_FIXED, S.K_DECIMAL, S.K_UUID, S.K_TIMEMILLI, S.K_TIMEMICRO, S.K_DURATION):
            self._decode_n4(c)
        elif k == S.K_ENUM:
            # append_enum, fast_decode.rs:570-578: `as usize` then symbols.get
            idx = read_zigzag_long(c) & 0xFFFFFFFFFFFFFFFF
            if idx >= len(self.node.symbols):
                raise DecodeError(f"enum index {idx} out of range")
            self._append_str(self.node.symbols[idx].encode())
        else:
            raise AssertionError(k)

    # ---- beyond the reference (Avro 1.11 specification wire forms) -------------------
    def _decode_n4(self, c: _Cur):
        n = self.node
        k = n.kind
        if k == S.K_TIMEMILLI:                   # int
            self._append_value(_as_i32(read_zigzag_long(c)))
        elif k == S.K_TIMEMICRO:                 # long
            self._append_value(read_zigzag_long(c))
        elif k == S.K_BYTES:                     # like string, no character set
            self._append_str(read_string(c))
        else:
            if n.wire_size >= 0:                 # fixed: exactly `size` bytes, no prefix
                if c.e - c.p < n.wire_size:
                    raise DecodeError("unexpected end of buffer (fixed)")
                raw = bytes(c.b[c.p:c.p + n.wire_size])
                c.p += n.wire_size
            else:
                raw = read_string(c)
            if k == S.K_FIXED:
                self._append_value(raw)
            elif k == S.K_DURATION:
                # Avro 1.11 "Duration": months, days, milliseconds as three little-endian u32.  The reference maps the
                # type to Duration(Millisecond) (schema_translate.rs:143), ONE count of milliseconds: days and
                # milliseconds add up to one; a months component has no length in milliseconds, so it is an error.
                mo, dy, ms = (int.from_bytes(raw[i:i + 4], "little") for i in (0, 4, 8))
                if mo != 0:
                    raise DecodeError(f"duration with {mo} months has no value in Duration(ms)")
                self._append_value(dy * 86_400_000 + ms)
            elif k == S.K_DECIMAL:               # big-endian two's complement unscaled value -> i128
                if len(raw) > 16:
                    raise DecodeError(f"decimal value of {len(raw)} bytes does not fit Decimal128")
                self._append_value(int.from_bytes(raw, "big", signed=True) if raw else 0)
            else:                                # uuid: hex text (8-4-4-4-12 or 32 digits) or 16 raw bytes
                if n.wire_size >= 0:
                    self._append_value(raw)
                else:
                    txt = raw
                    if len(txt) == 36:
                        if any(txt[i] != 0x2D for i in (8, 13, 18, 23)):
                            raise DecodeError("invalid uuid string")
                        txt = txt[0:8] + txt[9:13] + txt[14:18] + txt[19:23] + txt[24:36]
                    if len(txt) != 32 or any(ch not in b"0123456789abcdefABCDEF" for ch in txt):
                        raise DecodeError("invalid uuid string")
                    self._append_value(bytes.fromhex(txt.decode()))

    # ---- FieldDecoder::decode (fast_decode.rs:421-499) ------------------
    def decode(self, c: _Cur):
        n = self.node
        k = n.kind
        if k == S.K_NULL:
            self.length += 1
            return
        if n.nullable:
            if not union_branch(c, n.null_first):
                self.append_null()
                return
        if k == S.K_RECORD:
            self._record_present(c)
        elif k == S.K_UNION:
            self._union_decode(c)
        elif k == S.K_LIST:
            self._list_present(c)
        elif k == S.K_MAP:
            self._map_present(c)
        else:
            self._decode_value(c)

    # ---- FieldDecoder::append_null (fast_decode.rs:503-534) --------------
    def append_null(self):
        k = self.node.kind
        self.length += 1
        if k == S.K_NULL:
            return
        if k == S.K_RECORD:                      # 608-616
            self.valid.append(False)
            for ch in self.kids:
                ch.append_null()
        elif k == S.K_UNION:                     # 660-668
            for ch in self.kids:
                ch.append_null()
            self.values.append(0)
        elif k in (S.K_LIST, S.K_MAP):           # 721-727, 764-770
            self.offsets.append(self.offsets[-1])
            self.valid.append(False)
        elif k in (S.K_STRING, S.K_ENUM, S.K_BYTES):
            self.valid.append(False)
            self.offsets.append(len(self.data))
        else:
            self.valid.append(False)
            self.values.append(0)

    def _record_present(self, c: _Cur):          # 595-606
        self.valid.append(True)
        self.length += 1
        for ch in self.kids:
            ch.decode(c)

    def _union_decode(self, c: _Cur):            # 643-658
        idx = read_zigzag_long(c)
        if idx < 0 or idx >= len(self.kids):
            raise DecodeError(f"union branch index out of range: {idx}")
        for i, ch in enumerate(self.kids):
            if i == idx:
                ch.decode(c)
            else:
                ch.append_null()
        self.values.append(idx)
        self.length += 1

    def _list_present(self, c: _Cur):            # 703-719
        cur = self.offsets[-1]
        while True:
            n = read_block_count(c)
            if n == 0:
                break
            for _ in range(n):
                self.kids[0].decode(c)
                cur += 1
        self.offsets.append(cur)
        self.valid.append(True)
        self.length += 1

    def _map_present(self, c: _Cur):             # 745-762
        cur = self.offsets[-1]
        while True:
            n = read_block_count(c)
            if n == 0:
                break
            for _ in range(n):
                self.keys._append_str(read_string(c))
                self.kids[0].decode(c)
                cur += 1
        self.offsets.append(cur)
        self.valid.append(True)
        self.length += 1

    # ---- finish(): hand raw builder state to the shared assembler ----------
    def buffers(self) -> Buffers:
        k = self.node.kind
        b = Buffers(length=self.length)
        if k != S.K_NULL and k != S.K_UNION:
            b.valid = list(self.valid)
        if k in (S.K_INT, S.K_DATE, S.K_LONG, S.K_TSMILLI, S.K_TSMICRO, S.K_FLOAT, S.K_DOUBLE,
                 S.K_BOOL, S.K_UNION):
            b.values = list(self.values)
        if k in (S.K_STRING, S.K_ENUM, S.K_BYTES):
            b.offsets = list(self.offsets)
            b.data = bytes(self.data)
        if k in (S.K_FIXED, S.K_DECIMAL, S.K_UUID, S.K_TIMEMILLI, S.K_TIMEMICRO, S.K_DURATION):
            b.values = list(self.values)
        if k in (S.K_LIST, S.K_MAP):
            b.offsets = list(self.offsets)
        b.children = [ch.buffers() for ch in self.kids]
        if k == S.K_MAP:
            b.keys = self.keys.buffers()
        return b


def decode(records: List[bytes], schema_json: str, extended: bool = False):
    """fast_decode::decode (fast_decode.rs:806-835) -> pyarrow.RecordBatch.  ``extended``: also the N4 leaf types the
    reference's gate rejects (avro_schema.is_supported_extended)."""
    avro = S.parse_schema(schema_json, resolve_refs=extended)
    arrow_schema, root = S.build_tree(avro, extended)
    top = _B(root)
    for rec in records:
        c = _Cur(bytes(rec))
        top._record_present(c)      # leftover bytes are ignored (fast_decode.rs:825-828)
    return assemble_batch(arrow_schema, root, top.buffers())
