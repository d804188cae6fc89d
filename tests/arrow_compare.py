"""Buffer-for-buffer comparison of two Arrow arrays / batches (test helper).

``RecordBatch.equals`` is logical equality (it ignores e.g. an all-valid validity
buffer vs an absent one); the parity bar for the GPU path is stricter: same
buffers present, same meaningful bytes, same null counts -- the canonical form
of arrow-rs builders described in oracle/assemble.py.
"""
from __future__ import annotations

import numpy as np
import pyarrow as pa


class _View:
    """The first n bytes of an Arrow buffer as a zero-copy numpy view; == compares contents (vectorised, so the
    10M-record configuration gets the same bar as the small ones)."""

    def __init__(self, buf, n):
        if n == 0:
            self.a = np.zeros(0, dtype=np.uint8)
            return
        assert buf is not None, "missing buffer"
        assert buf.size >= n, f"buffer too small: {buf.size} < {n}"
        self.a = np.frombuffer(buf, dtype=np.uint8, count=n)

    def __eq__(self, other):
        return self.a.shape == other.a.shape and bool(np.array_equal(self.a, other.a))

    def i32(self):
        return self.a.view(np.int32)


def _bytes(buf, n):
    return _View(buf, n)


def _bits(buf, nbits):
    """Validity / boolean bitmap restricted to its nbits meaningful bits, as bytes with the padding bits masked."""
    nb = (nbits + 7) // 8
    raw = _View(buf, nb).a
    if nbits & 7 and nb:
        raw = raw.copy()
        raw[-1] &= (1 << (nbits & 7)) - 1
    return raw


def assert_identical(a: pa.Array, b: pa.Array, path: str = "") -> None:
    assert a.type == b.type, f"{path}: type {a.type} != {b.type}"
    assert len(a) == len(b), f"{path}: length {len(a)} != {len(b)}"
    assert a.offset == 0 and b.offset == 0, f"{path}: non-zero offset"
    assert a.null_count == b.null_count, f"{path}: null_count {a.null_count} != {b.null_count}"
    n = len(a)
    t = a.type
    ba, bb = a.buffers(), b.buffers()
    if pa.types.is_null(t):
        return
    if not pa.types.is_union(t):
        assert (ba[0] is None) == (bb[0] is None), f"{path}: validity presence {ba[0] is not None} != {bb[0] is not None}"
        if ba[0] is not None:
            assert np.array_equal(_bits(ba[0], n), _bits(bb[0], n)), f"{path}: validity bits differ"
    if pa.types.is_boolean(t):
        assert np.array_equal(_bits(ba[1], n), _bits(bb[1], n)), f"{path}: boolean values differ"
    elif pa.types.is_string(t) or pa.types.is_binary(t):
        assert _bytes(ba[1], 4 * (n + 1)) == _bytes(bb[1], 4 * (n + 1)), f"{path}: string offsets differ"
        last = int(_bytes(ba[1], 4 * (n + 1)).i32()[-1])
        assert _bytes(ba[2], last) == _bytes(bb[2], last), f"{path}: string data differ"
    elif pa.types.is_struct(t):
        for i in range(t.num_fields):
            assert_identical(a.field(i), b.field(i), f"{path}.{t.field(i).name}")
    elif pa.types.is_union(t):
        assert _bytes(ba[1], n) == _bytes(bb[1], n), f"{path}: type_ids differ"
        for i in range(t.num_fields):
            assert_identical(a.field(i), b.field(i), f"{path}<{t.field(i).name}>")
    elif pa.types.is_map(t):
        assert _bytes(ba[1], 4 * (n + 1)) == _bytes(bb[1], 4 * (n + 1)), f"{path}: map offsets differ"
        assert_identical(a.keys, b.keys, f"{path}.keys")
        assert_identical(a.items, b.items, f"{path}.values")
    elif pa.types.is_list(t):
        assert _bytes(ba[1], 4 * (n + 1)) == _bytes(bb[1], 4 * (n + 1)), f"{path}: list offsets differ"
        assert_identical(a.values, b.values, f"{path}[]")
    else:
        w = t.bit_width // 8
        assert _bytes(ba[1], w * n) == _bytes(bb[1], w * n), f"{path}: values differ"


def assert_batches_identical(a: pa.RecordBatch, b: pa.RecordBatch) -> None:
    assert a.schema.equals(b.schema, check_metadata=True), f"schema differs:\n{a.schema}\nvs\n{b.schema}"
    assert a.num_rows == b.num_rows
    for i, f in enumerate(a.schema):
        assert_identical(a.column(i), b.column(i), f.name)
    # (no logical .equals() here: NaN != NaN there, and buffer identity above is the stricter bar)
