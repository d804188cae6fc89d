"""Device-resident output for consumers that keep the data on the GPU (torch / cuDF-style).

    dec = pyruhvro_amd.deserialize_to_device(records, schema, num_chunks)
    col = dec.batches[0].column("created_at")              # DeviceColumn: Arrow layout, buffers in HBM
    t = torch.from_dlpack(col.values)                      # int64 tensor over the engine's buffer: no copy
    total = int(t.sum())

The decode is the reference's chunked direct decode (``deserialize_array_threaded``, ``src/lib.rs:73-89``) with the same
chunk boundaries and the same Arrow buffers -- they are just left where the kernels wrote them.  Every buffer of every column is a
``DeviceBuffer`` that implements the DLPack protocol (``__dlpack__`` / ``__dlpack_device__``, device type kDLROCM): a 1-D array
of the buffer's element type (values: the column's physical type; offsets: int32; validity / boolean bitmaps and string data:
uint8).  A DLPack capsule keeps the whole decode result alive until its consumer drops the tensor, so views may outlive
``dec``.  Nothing here depends on PyTorch; it is one consumer of the protocol.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import pyarrow as pa

from . import cabi

KDLROCM = 10
_DL_INT, _DL_UINT, _DL_FLOAT = 0, 1, 2


class _DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int32), ("device_id", C.c_int32)]


class _DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class _DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", _DLDevice), ("ndim", C.c_int32), ("dtype", _DLDataType),
                ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class _DLManagedTensor(C.Structure):
    pass


_DELETER = C.CFUNCTYPE(None, C.POINTER(_DLManagedTensor))
_DLManagedTensor._fields_ = [("dl_tensor", _DLTensor), ("manager_ctx", C.c_void_p), ("deleter", _DELETER)]

# what every exported DLManagedTensor needs to stay alive until its consumer calls the deleter: the struct itself, its shape
# array, and the decode result that owns the device memory
_live: Dict[int, tuple] = {}


@_DELETER
def _dl_deleter(mt):
    _live.pop(C.addressof(mt.contents), None)


_CAPSULE_DESTRUCTOR = C.CFUNCTYPE(None, C.c_void_p)
_api = C.pythonapi
_api.PyCapsule_New.restype = C.py_object
_api.PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, _CAPSULE_DESTRUCTOR]
_api.PyCapsule_IsValid.restype = C.c_int
_api.PyCapsule_IsValid.argtypes = [C.c_void_p, C.c_char_p]
_api.PyCapsule_GetPointer.restype = C.c_void_p
_api.PyCapsule_GetPointer.argtypes = [C.c_void_p, C.c_char_p]


@_CAPSULE_DESTRUCTOR
def _capsule_destructor(cap):
    # a capsule nobody consumed is still called "dltensor": its tensor is ours to delete
    if _api.PyCapsule_IsValid(cap, b"dltensor"):
        p = _api.PyCapsule_GetPointer(cap, b"dltensor")
        _live.pop(p, None)


class DeviceBuffer:
    """One Arrow buffer in HBM: ``ptr``, ``nbytes``, element ``dtype`` (numpy), ``count`` elements.  DLPack producer."""

    __slots__ = ("ptr", "nbytes", "device", "dtype", "count", "_owner")

    def __init__(self, ptr: int, nbytes: int, device: int, dtype, count: int, owner):
        self.ptr, self.nbytes, self.device, self.dtype, self.count, self._owner = ptr, nbytes, device, np.dtype(dtype), count, owner

    def __dlpack_device__(self):
        return (KDLROCM, self.device)

    def __dlpack__(self, stream=None, **_):
        # (the decode call is settled before its buffers are handed out: nothing to order against `stream`)
        mt = _DLManagedTensor()
        shape = (C.c_int64 * 1)(self.count)
        kind = self.dtype.kind
        code = _DL_FLOAT if kind == "f" else _DL_INT if kind == "i" else _DL_UINT
        mt.dl_tensor = _DLTensor(self.ptr, _DLDevice(KDLROCM, self.device), 1, _DLDataType(code, self.dtype.itemsize * 8, 1),
                                 shape, None, 0)
        mt.manager_ctx = None
        mt.deleter = _dl_deleter
        _live[C.addressof(mt)] = (mt, shape, self._owner)
        return _api.PyCapsule_New(C.addressof(mt), b"dltensor", _capsule_destructor)

    def __repr__(self):
        return f"DeviceBuffer({self.count} x {self.dtype}, {self.nbytes} B @ 0x{self.ptr:x}, rocm:{self.device})"


class DeviceColumn:
    """One Arrow array whose buffers live in HBM.  ``buffers``: role -> DeviceBuffer ("validity", "values", "offsets", "data",
    "type_ids"; a role the Arrow layout of ``type`` does not have is absent, a validity bitmap of a column without nulls is None, as
    in Arrow).  ``values`` / ``offsets`` / ``data`` / ``validity`` are shortcuts; ``children`` the child columns (struct fields, list
    items, map entries, union variants)."""

    def __init__(self, name: str, typ: pa.DataType, length: int, null_count: int, buffers: dict, children: list):
        self.name, self.type, self.length, self.null_count, self.buffers, self.children = name, typ, length, null_count, buffers, children

    values = property(lambda self: self.buffers.get("values"))
    offsets = property(lambda self: self.buffers.get("offsets"))
    data = property(lambda self: self.buffers.get("data"))
    validity = property(lambda self: self.buffers.get("validity"))

    def child(self, name: str) -> "DeviceColumn":
        for c in self.children:
            if c.name == name:
                return c
        raise KeyError(name)

    def __repr__(self):
        return f"DeviceColumn({self.name!r}: {self.type}, {self.length} rows, {self.null_count} nulls)"


class DeviceBatch:
    """One chunk (what would be one RecordBatch): ``schema``, ``num_rows``, ``columns``."""

    def __init__(self, schema: pa.Schema, num_rows: int, columns: List[DeviceColumn]):
        self.schema, self.num_rows, self.columns = schema, num_rows, columns

    def column(self, key) -> DeviceColumn:
        return self.columns[key] if isinstance(key, int) else self.columns[self.schema.get_field_index(key)]


_FIXED = {pa.int32(): np.int32, pa.int64(): np.int64, pa.float32(): np.float32, pa.float64(): np.float64, pa.date32(): np.int32}


def _value_dtype(t: pa.DataType):
    if t in _FIXED:
        return _FIXED[t]
    if pa.types.is_timestamp(t) or pa.types.is_time64(t) or pa.types.is_duration(t):
        return np.int64
    if pa.types.is_time32(t):
        return np.int32
    return None


def _column(name: str, t: pa.DataType, a: "cabi.ArrowArray", sizes: Dict[int, int], device: int, owner) -> DeviceColumn:
    def buf(i: int, dtype, count: Optional[int] = None):
        p = a.buffers[i] if i < a.n_buffers else None
        if not p:
            return None
        nbytes = sizes.get(p)
        if nbytes is None:
            raise RuntimeError(f"column {name!r}: buffer {i} is not in the result's buffer table")
        isz = np.dtype(dtype).itemsize
        return DeviceBuffer(p, nbytes, device, dtype, nbytes // isz if count is None else min(count, nbytes // isz), owner)

    n = int(a.length)
    bufs: dict = {}
    kids: List[DeviceColumn] = []
    vd = _value_dtype(t)
    if pa.types.is_null(t):
        pass
    elif vd is not None:
        bufs = {"validity": buf(0, np.uint8), "values": buf(1, vd, n)}
    elif pa.types.is_boolean(t):
        bufs = {"validity": buf(0, np.uint8), "values": buf(1, np.uint8)}           # a bitmap, like the validity
    elif pa.types.is_string(t) or pa.types.is_binary(t):
        bufs = {"validity": buf(0, np.uint8), "offsets": buf(1, np.int32, n + 1), "data": buf(2, np.uint8)}
    elif pa.types.is_fixed_size_binary(t) or pa.types.is_decimal(t):
        bufs = {"validity": buf(0, np.uint8), "values": buf(1, np.uint8)}
    elif pa.types.is_struct(t):
        bufs = {"validity": buf(0, np.uint8)}
        kids = [_column(t.field(i).name, t.field(i).type, a.children[i].contents, sizes, device, owner) for i in range(t.num_fields)]
    elif pa.types.is_map(t):
        bufs = {"validity": buf(0, np.uint8), "offsets": buf(1, np.int32, n + 1)}
        entries = pa.struct([pa.field("keys", t.key_type, nullable=False), t.item_field])
        kids = [_column("entries", entries, a.children[0].contents, sizes, device, owner)]
    elif pa.types.is_list(t):
        bufs = {"validity": buf(0, np.uint8), "offsets": buf(1, np.int32, n + 1)}
        kids = [_column(t.value_field.name, t.value_type, a.children[0].contents, sizes, device, owner)]
    elif pa.types.is_union(t):
        bufs = {"type_ids": buf(0, np.int8, n)}
        kids = [_column(t.field(i).name, t.field(i).type, a.children[i].contents, sizes, device, owner) for i in range(t.num_fields)]
    else:
        raise NotImplementedError(f"column {name!r}: no device view for Arrow type {t}")
    return DeviceColumn(name, t, n, int(a.null_count), bufs, kids)


class _Hip:
    """The few HIP runtime calls the upload of host records needs, on the runtime the engine itself is bound to."""

    _lib = None

    @classmethod
    def get(cls):
        if cls._lib is None:
            cabi.lib()
            L = None
            for so in ("libamdhip64.so.7", "libamdhip64.so.6", "libamdhip64.so"):      # (the soname list of rtc_compile.cpp)
                try:
                    L = C.CDLL(so)
                    break
                except OSError:
                    continue
            if L is None:
                raise RuntimeError("the HIP runtime (libamdhip64.so) is not loadable")
            L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            L.hipFree.argtypes = [C.c_void_p]
            L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            L.hipSetDevice.argtypes = [C.c_int]
            cls._lib = L
        return cls._lib


class _DevMem:
    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        if _Hip.get().hipMalloc(C.byref(self.ptr), max(int(nbytes), 64)) != 0:
            raise RuntimeError(f"hipMalloc of {nbytes} bytes failed")

    def upload(self, arr: np.ndarray):
        if arr.nbytes and _Hip.get().hipMemcpy(self.ptr, arr.ctypes.data, arr.nbytes, 1) != 0:
            raise RuntimeError("hipMemcpy (host to device) failed")
        return self

    def free(self):
        if self.ptr:
            _Hip.get().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    __del__ = free


class DeviceDecode:
    """Result of ``deserialize_to_device``: ``batches`` (one DeviceBatch per chunk, reference chunk boundaries), ``stats``.  Owns the
    device memory; DLPack views keep it alive on their own, ``free()`` lets go of it early (views exported before stay valid)."""

    def __init__(self, result: "cabi.DeviceResult", schema: pa.Schema):
        self._result = result
        self.schema = schema
        self.stats = result.stats
        self.device = 0
        self.batches: List[DeviceBatch] = []
        L = cabi.lib()
        nbuf = L.rh_device_result_buffers(result.handle, 0, None, None, 0)
        for c in range(result.chunks):
            ptrs, sizes = (C.c_uint64 * max(nbuf, 1))(), (C.c_uint64 * max(nbuf, 1))()
            if L.rh_device_result_buffers(result.handle, c, ptrs, sizes, nbuf) != nbuf:
                raise RuntimeError("rh_device_result_buffers failed")
            table = {int(ptrs[i]): int(sizes[i]) for i in range(nbuf)}
            view = result.export(c)
            try:
                self.device = int(view.device_id)
                top = view.array
                cols = [_column(schema.field(i).name, schema.field(i).type, top.children[i].contents, table, self.device, result)
                        for i in range(len(schema))]
                self.batches.append(DeviceBatch(schema, int(top.length), cols))
            finally:
                if view.array.release:
                    C.CFUNCTYPE(None, C.POINTER(cabi.ArrowArray))(view.array.release)(C.byref(view.array))

    def to_host(self) -> List[pa.RecordBatch]:
        """The same batches copied to host memory (what deserialize_array_threaded returns)."""
        return self._result.to_host()

    def free(self):
        self.batches = []
        self._result = None           # (DLPack views hold their own reference)


def _pack(records) -> tuple:
    """(payload uint8[...], offsets uint64[n + 1]) from a list of bytes-like objects or a pyarrow (Large)BinaryArray."""
    if isinstance(records, pa.ChunkedArray):
        records = records.combine_chunks() if records.num_chunks != 1 else records.chunk(0)
    if isinstance(records, (pa.BinaryArray, pa.LargeBinaryArray)):
        if records.null_count:
            raise ValueError("null records cannot be decoded")
        n = len(records)
        bufs = records.buffers()
        odt = np.int64 if isinstance(records, pa.LargeBinaryArray) else np.int32
        offs = np.frombuffer(bufs[1], dtype=odt, count=records.offset + n + 1)[records.offset:].astype(np.uint64)
        base, end = (int(offs[0]), int(offs[-1])) if n else (0, 0)
        data = np.frombuffer(bufs[2], dtype=np.uint8, count=end)[base:] if bufs[2] is not None and end > base else np.zeros(0, np.uint8)
        return data, offs - np.uint64(base)
    if not isinstance(records, (list, tuple)):
        raise TypeError("argument 'records': expected a list of bytes or a pyarrow BinaryArray")
    lens = np.fromiter((len(r) for r in records), dtype=np.uint64, count=len(records))
    offs = np.zeros(len(records) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    return np.frombuffer(b"".join(records), dtype=np.uint8), offs


def deserialize_to_device(records, schema: str, num_chunks: int, device: int = -1, stream: int = 0, kernel: int = 0) -> DeviceDecode:
    """Extension.  Decode like ``deserialize_array_threaded(records, schema, num_chunks)`` but leave the Arrow
    buffers in HBM.  ``records``: a list of ``bytes``, a pyarrow ``BinaryArray`` / ``LargeBinaryArray`` (packed on the host and
    uploaded once), or a pair ``(data, offsets)`` of device arrays that already hold the packed payload and its n + 1 uint64
    offsets (anything with ``data_ptr()`` -- torch tensors -- or raw integer addresses; ``data`` 16-byte aligned with 64 readable
    bytes of slack).  ``stream``: the HIP stream to launch on (e.g. ``torch.cuda.current_stream().cuda_stream``).
    Errors are the reference's: ``ValueError`` with its message for a malformed datum or an unsupported schema."""
    if not isinstance(num_chunks, int) or isinstance(num_chunks, bool):
        raise TypeError("argument 'num_chunks': expected int")
    if num_chunks < 0:
        raise OverflowError("can't convert negative int to unsigned")
    s = cabi.Schema.get(schema)
    keep = []
    if isinstance(records, tuple) and len(records) == 2 and not isinstance(records[0], (bytes, bytearray)):
        d_data, d_off = records
        ptr = lambda x: int(x.data_ptr()) if hasattr(x, "data_ptr") else int(x)      # noqa: E731
        n = (int(d_off.numel()) if hasattr(d_off, "numel") else int(records[1].shape[0])) - 1
        if hasattr(d_off, "__getitem__") and hasattr(d_off, "data_ptr"):
            data_len = int(d_off[-1])
        else:
            raise TypeError("device input: offsets must be a device array with data_ptr() (e.g. a torch int64 / uint64 tensor)")
        p_data, p_off = ptr(d_data), ptr(d_off)
        keep = [d_data, d_off]
    else:
        data, offs = _pack(records)
        n, data_len = len(offs) - 1, int(offs[-1])
        if device >= 0:
            _Hip.get().hipSetDevice(device)
        md = _DevMem(len(data) + 64).upload(np.ascontiguousarray(data))
        mo = _DevMem(8 * len(offs)).upload(np.ascontiguousarray(offs))
        p_data, p_off = md.ptr.value, mo.ptr.value
        keep = [md, mo]
    try:
        r = cabi.decode_device(p_data, p_off, data_len, n, schema, num_chunks, device=device, stream=stream, kernel=kernel)
    finally:
        for k in keep:
            if isinstance(k, _DevMem):
                k.free()                # (a synchronous call: the kernels are done with the input)
    return DeviceDecode(r, s.arrow_schema)
