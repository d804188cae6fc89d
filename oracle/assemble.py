"""ORACLE (test infrastructure, not product code) -- builder state -> Arrow arrays.

Restates the ``finish()`` half of the reference walker
(ruhvro/src/fast_decode.rs:536-567, 618-639, 670-683, 729-741, 772-798) and
the arrow-rs 58.3.0 builder ``finish`` conventions it relies on:

  * leaf builders (Primitive/Boolean/GenericString) keep a lazy null buffer:
    the validity bitmap exists iff at least one null was appended;
  * values under a null slot are zero, string offsets repeat;
  * nullable struct / list / map always carry a validity buffer
    (fast_decode.rs:629,739,790); non-nullable ones never do;
  * sparse union = i8 type_ids, no offsets, no validity (680-681);
  * map entries struct has no validity, keys_sorted = false (782-797).

Both oracle walkers (py_walker, c_walker) feed this module.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import pyarrow as pa

from . import avro_schema as S


@dataclass
class Buffers:
    """Raw builder contents of one node.  ``valid``/``values``/``offsets`` may be
    Python lists (py_walker) or numpy arrays (c_walker)."""
    length: int = 0
    valid: Optional[object] = None      # bool per row
    values: Optional[object] = None     # ints / float bit patterns / bools / type ids
    offsets: Optional[object] = None    # int32[n+1]
    data: Optional[bytes] = None        # string bytes
    children: List["Buffers"] = field(default_factory=list)
    keys: Optional["Buffers"] = None    # map keys


def _bitmap(bits) -> pa.Buffer:
    a = np.asarray(bits, dtype=np.uint8)
    return pa.py_buffer(np.packbits(a, bitorder="little").tobytes())


def _validity(b: Buffers, always: bool):
    """-> (buffer or None, null_count)."""
    if b.valid is None:
        return None, 0
    v = np.asarray(b.valid, dtype=bool)
    nulls = int(v.size - np.count_nonzero(v))
    if nulls > 0:
        return _bitmap(v), nulls
    if always:
        # Arrow C++ drops a validity buffer handed over with null_count == 0 (ArrayData::Make);
        # null_count = -1 ("unknown") keeps the always-present bitmap of nullable struct/list/map,
        # which is what pyarrow holds after importing arrow-rs' arrays over the C Data Interface.
        return _bitmap(v), -1
    return None, 0


_NP = {
    S.K_TIMEMILLI: np.int32, S.K_TIMEMICRO: np.int64, S.K_DURATION: np.int64,          # N4 (beyond the reference)
    S.K_INT: np.int32, S.K_DATE: np.int32,
    S.K_LONG: np.int64, S.K_TSMILLI: np.int64, S.K_TSMICRO: np.int64,
    S.K_FLOAT: np.uint32, S.K_DOUBLE: np.uint64,
}


def _string_array(b: Buffers, dtype: pa.DataType) -> pa.Array:
    vb, nulls = _validity(b, always=False)
    offs = np.asarray(b.offsets, dtype=np.int32)
    assert offs.size == b.length + 1
    return pa.Array.from_buffers(dtype, b.length, [vb, pa.py_buffer(offs.tobytes()), pa.py_buffer(bytes(b.data))],
                                 null_count=nulls)


def assemble(node: S.Node, b: Buffers) -> pa.Array:
    """FieldDecoder::finish (fast_decode.rs:536-567)."""
    k = node.kind
    dt = node.field.type
    n = b.length
    if k == S.K_NULL:
        return pa.nulls(n, pa.null())
    if k in _NP:
        vb, nulls = _validity(b, always=False)
        vals = np.asarray(b.values, dtype=np.int64 if _NP[k] in (np.int32, np.int64) else np.uint64).astype(_NP[k])
        assert vals.size == n
        return pa.Array.from_buffers(dt, n, [vb, pa.py_buffer(vals.tobytes())], null_count=nulls)
    if k == S.K_BOOL:
        vb, nulls = _validity(b, always=False)
        return pa.Array.from_buffers(dt, n, [vb, _bitmap(np.asarray(b.values, dtype=bool))], null_count=nulls)
    if k in (S.K_STRING, S.K_ENUM, S.K_BYTES):
        return _string_array(b, dt)
    if k in (S.K_FIXED, S.K_UUID, S.K_DECIMAL):           # N4: fixed-width values, zero under nulls, lazy validity
        vb, nulls = _validity(b, always=False)
        w = 16 if k != S.K_FIXED else dt.byte_width
        raw = bytearray()
        for v in b.values:
            if k == S.K_DECIMAL:
                raw += int(v).to_bytes(16, "little", signed=True)
            else:
                raw += v if isinstance(v, (bytes, bytearray)) and len(v) == w else bytes(w)
        assert len(raw) == w * n
        return pa.Array.from_buffers(dt, n, [vb, pa.py_buffer(bytes(raw))], null_count=nulls)
    if k == S.K_RECORD:                                   # fast_decode.rs:618-639
        kids = [assemble(cn, cb) for cn, cb in zip(node.children, b.children)]
        vb, nulls = _validity(b, always=True) if node.nullable else (None, 0)
        return pa.Array.from_buffers(dt, n, [vb], null_count=nulls, children=kids)
    if k == S.K_UNION:                                    # fast_decode.rs:670-683
        kids = [assemble(cn, cb) for cn, cb in zip(node.children, b.children)]
        tids = np.asarray(b.values, dtype=np.int8)
        assert tids.size == n
        return pa.Array.from_buffers(dt, n, [None, pa.py_buffer(tids.tobytes())], null_count=0, children=kids)
    if k == S.K_LIST:                                     # fast_decode.rs:729-741
        values = assemble(node.children[0], b.children[0])
        vb, nulls = _validity(b, always=True) if node.nullable else (None, 0)
        offs = np.asarray(b.offsets, dtype=np.int32)
        assert offs.size == n + 1
        return pa.Array.from_buffers(dt, n, [vb, pa.py_buffer(offs.tobytes())], null_count=nulls, children=[values])
    if k == S.K_MAP:                                      # fast_decode.rs:772-798
        keys = _string_array(b.keys, pa.string())
        values = assemble(node.children[0], b.children[0])
        entries_t = pa.struct([dt.key_field, dt.item_field])
        entries = pa.Array.from_buffers(entries_t, len(keys), [None], null_count=0, children=[keys, values])
        vb, nulls = _validity(b, always=True) if node.nullable else (None, 0)
        offs = np.asarray(b.offsets, dtype=np.int32)
        assert offs.size == n + 1
        return pa.Array.from_buffers(dt, n, [vb, pa.py_buffer(offs.tobytes())], null_count=nulls, children=[entries])
    raise AssertionError(k)


def assemble_batch(arrow_schema: pa.Schema, root: S.Node, top: Buffers) -> pa.RecordBatch:
    """RecordBatch::try_new (fast_decode.rs:829-834)."""
    cols = [assemble(cn, cb) for cn, cb in zip(root.children, top.children)]
    if not cols:
        raise ValueError("RecordDecoder produced a record with 0 fields")
    return pa.RecordBatch.from_arrays(cols, schema=arrow_schema)
