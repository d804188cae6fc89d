"""ORACLE (test infrastructure, not product code) -- ctypes front for oracle_walk.c.

``decode_threaded`` mirrors ruhvro/src/deserialize.rs:76-121 (k batches, chunk
boundaries of deserialize.rs:53-68); ``decode`` mirrors deserialize.rs:25-30.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence

import numpy as np

from . import avro_schema as S
from .assemble import Buffers, assemble_batch
from .build import LIB, build


class OrcNode(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("kind", "nullable", "null_first", "nchildren", "first_child", "nsymbols", "first_symbol")]


class OrcView(C.Structure):
    _fields_ = [
        ("length", C.c_uint64), ("null_count", C.c_uint64),
        ("valid", C.c_void_p), ("valid_bits", C.c_uint64),
        ("values", C.c_void_p), ("values_len", C.c_uint64),
        ("bvalues", C.c_void_p), ("bvalues_bits", C.c_uint64),
        ("offsets", C.c_void_p), ("offsets_len", C.c_uint64),
        ("type_ids", C.c_void_p), ("type_ids_len", C.c_uint64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.orc_decode.restype = C.c_void_p
        _lib.orc_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int,
                                    C.c_char_p, C.c_size_t]
        _lib.orc_num_chunks.argtypes = [C.c_void_p]
        _lib.orc_free.argtypes = [C.c_void_p]
        _lib.orc_view_node.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(OrcView)]
    return _lib


class CompiledSchema:
    """Flat node table for oracle_walk.c."""

    def __init__(self, schema_json: str):
        avro = S.parse_schema(schema_json)
        self.arrow_schema, self.root = S.build_tree(avro)
        self.nodes = S.flatten(self.root)
        child_idx: List[int] = []
        sym_off: List[int] = []
        sym_data = bytearray()
        arr = (OrcNode * len(self.nodes))()
        for i, n in enumerate(self.nodes):
            arr[i].kind = n.kind
            arr[i].nullable = int(n.nullable)
            arr[i].null_first = int(n.null_first)
            arr[i].nchildren = len(n.children)
            arr[i].first_child = len(child_idx)
            child_idx.extend(c.idx for c in n.children)
            arr[i].nsymbols = len(n.symbols)
            arr[i].first_symbol = len(sym_off)
            for s in n.symbols:
                sym_off.append(len(sym_data))
                sym_data += s.encode()
            sym_off.append(len(sym_data))
        self.c_nodes = arr
        self.c_child = np.asarray(child_idx + [0], dtype=np.int32)
        self.c_symoff = np.asarray(sym_off + [0], dtype=np.int32)
        self.c_symdata = np.frombuffer(bytes(sym_data) + b"\0", dtype=np.uint8).copy()


def pack(records: Sequence[bytes]):
    lens = np.fromiter((len(r) for r in records), dtype=np.uint64, count=len(records))
    offsets = np.zeros(len(records) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    data = np.frombuffer(b"".join(records) + b"\0", dtype=np.uint8)
    return data, offsets


def _np_from(ptr, nbytes, dtype):
    if not ptr or nbytes == 0:
        return np.zeros(0, dtype=dtype)
    return np.frombuffer(C.string_at(ptr, int(nbytes)), dtype=dtype)


def _bits(ptr, nbits):
    if not ptr:
        return None
    raw = np.frombuffer(C.string_at(ptr, (int(nbits) + 7) // 8), dtype=np.uint8)
    return np.unpackbits(raw, bitorder="little")[: int(nbits)].astype(bool)


def _collect(h, chunk: int, node: S.Node, keys: bool = False) -> Buffers:
    v = OrcView()
    if lib().orc_view_node(h, chunk, node.idx, int(keys), C.byref(v)):
        raise RuntimeError("oracle: node not found")
    k = S.K_STRING if keys else node.kind
    b = Buffers(length=int(v.length))
    if k not in (S.K_NULL, S.K_UNION):
        vb = _bits(v.valid, v.valid_bits)
        b.valid = vb if vb is not None else np.ones(b.length, dtype=bool)
    if k in (S.K_INT, S.K_DATE):
        b.values = _np_from(v.values, v.values_len, np.int32)
    elif k == S.K_FLOAT:
        b.values = _np_from(v.values, v.values_len, np.uint32)
    elif k in (S.K_LONG, S.K_TSMILLI, S.K_TSMICRO):
        b.values = _np_from(v.values, v.values_len, np.int64)
    elif k == S.K_DOUBLE:
        b.values = _np_from(v.values, v.values_len, np.uint64)
    elif k == S.K_BOOL:
        bv = _bits(v.bvalues, v.bvalues_bits)
        b.values = bv if bv is not None else np.zeros(0, dtype=bool)
    elif k == S.K_UNION:
        b.values = _np_from(v.type_ids, v.type_ids_len, np.int8)
    elif k in (S.K_STRING, S.K_ENUM):
        b.offsets = _np_from(v.offsets, v.offsets_len, np.int32)
        b.data = C.string_at(v.values, int(v.values_len)) if v.values_len else b""
    elif k in (S.K_LIST, S.K_MAP):
        b.offsets = _np_from(v.offsets, v.offsets_len, np.int32)
    if not keys:
        b.children = [_collect(h, chunk, c) for c in node.children]
        if k == S.K_MAP:
            b.keys = _collect(h, chunk, node, keys=True)
    return b


def decode_packed(cs: CompiledSchema, data: np.ndarray, offsets: np.ndarray, num_chunks: int,
                  threaded: bool, materialize: bool = True):
    """Run the C walker.  ``materialize=False`` only times the decode (bench cpu_baseline)."""
    n = len(offsets) - 1
    err = C.create_string_buffer(256)
    h = lib().orc_decode(C.addressof(cs.c_nodes), len(cs.nodes), cs.c_child.ctypes.data,
                         cs.c_symdata.ctypes.data, cs.c_symoff.ctypes.data,
                         data.ctypes.data, offsets.ctypes.data, n, max(int(num_chunks), 0), int(threaded),
                         err, 256)
    if not h:
        raise ValueError(err.value.decode())
    try:
        if not materialize:
            return lib().orc_num_chunks(h)
        out = []
        for ci in range(lib().orc_num_chunks(h)):
            top = _collect(h, ci, cs.root)
            out.append(assemble_batch(cs.arrow_schema, cs.root, top))
        return out
    finally:
        lib().orc_free(h)


def decode(records: Sequence[bytes], schema_json: str):
    """per_datum_deserialize (deserialize.rs:25-30): one batch."""
    cs = CompiledSchema(schema_json)
    data, offsets = pack(records)
    return decode_packed(cs, data, offsets, 1, threaded=False)[0]


def decode_threaded(records: Sequence[bytes], schema_json: str, num_chunks: int):
    """per_datum_deserialize_threaded (deserialize.rs:76-121): k batches."""
    cs = CompiledSchema(schema_json)
    data, offsets = pack(records)
    return decode_packed(cs, data, offsets, num_chunks, threaded=True)
