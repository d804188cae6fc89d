"""ctypes view of the C ABI in include/ruhvro_hip.h (libruhvro_hip.so).

Used by bench.py (device-resident decode on torch-owned memory), by the
packed-input Python entry point, and by the tests that exercise the ABI
directly.  Nothing here decodes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import numpy as np
import pyarrow as pa

_HERE = os.path.dirname(os.path.abspath(__file__))
# RUHVRO_HIP_LIB: another build of the same library (the sanitizer build of _build.build_sanitized); never a fallback
LIB_PATH = os.environ.get("RUHVRO_HIP_LIB") or os.path.join(_HERE, "libruhvro_hip.so")

RH_OK, RH_ERR_SCHEMA, RH_ERR_DECODE, RH_ERR_RUNTIME, RH_ERR_ARGUMENT = range(5)


class RhStats(C.Structure):
    _fields_ = [("records", C.c_uint64), ("input_bytes", C.c_uint64), ("output_bytes", C.c_uint64),
                ("chunks", C.c_uint32), ("blocks", C.c_uint32), ("pack_ms", C.c_float), ("h2d_ms", C.c_float),
                ("size_kernel_ms", C.c_float), ("scan_kernel_ms", C.c_float), ("emit_kernel_ms", C.c_float),
                ("d2h_ms", C.c_float), ("total_ms", C.c_float), ("specialized", C.c_uint32), ("lds_bytes", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RhOpts(C.Structure):
    """rh_opts (ABI version 5: `struct_size` took the place of ABI 3's must-be-zero `reserved0`).  Build with make_opts()."""
    _fields_ = [("device", C.c_int32), ("flags", C.c_int32), ("stream", C.c_void_p),
                ("devices", C.POINTER(C.c_int32)), ("n_devices", C.c_uint32), ("struct_size", C.c_uint32),
                ("chunk_rows", C.c_uint64), ("device_stats", C.POINTER(RhStats)),
                ("ready", C.POINTER(C.c_uint64)), ("gathered", C.POINTER(C.c_uint64))]   # streaming hand-over (rh_decode): NULL = off


def make_opts(device: int = -1, kernel: int = 0, stream=None, devices=None, chunk_rows: int = 0):
    """-> (RhOpts, keepalive).  `devices`: sequence of HIP ordinals to shard the chunks over (repeats allowed);
    per-shard stats land in keepalive["device_stats"]."""
    o = RhOpts()
    o.struct_size = C.sizeof(RhOpts)
    o.device, o.flags, o.stream = device, kernel, stream or None
    o.chunk_rows = chunk_rows
    keep = {}
    if devices:
        arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        st = (RhStats * len(devices))()
        o.devices = C.cast(arr, C.POINTER(C.c_int32))
        o.n_devices = len(devices)
        o.device_stats = C.cast(st, C.POINTER(RhStats))
        keep["devices"], keep["device_stats"] = arr, st
    return o, keep


class ArrowArray(C.Structure):
    pass


ArrowArray._fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64),
                       ("n_buffers", C.c_int64), ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)),
                       ("children", C.POINTER(C.POINTER(ArrowArray))), ("dictionary", C.POINTER(ArrowArray)),
                       ("release", C.c_void_p), ("private_data", C.c_void_p)]


class ArrowSchema(C.Structure):
    pass


ArrowSchema._fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_void_p), ("flags", C.c_int64),
                        ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(ArrowSchema))),
                        ("dictionary", C.POINTER(ArrowSchema)), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class ArrowDeviceArray(C.Structure):
    _fields_ = [("array", ArrowArray), ("device_id", C.c_int64), ("device_type", C.c_int32),
                ("sync_event", C.c_void_p), ("reserved", C.c_int64 * 3)]


_lib = None


def lib():
    """Load libruhvro_hip.so (raises OSError if it was not built -- there is no fallback)."""
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.rh_schema_compile.restype = C.c_void_p
        L.rh_schema_compile.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p)]
        L.rh_schema_free.argtypes = [C.c_void_p]
        L.rh_schema_export.argtypes = [C.c_void_p, C.POINTER(ArrowSchema)]
        L.rh_clamp_chunks.restype = C.c_uint32
        L.rh_clamp_chunks.argtypes = [C.c_uint64, C.c_uint64]
        L.rh_shard_chunks.restype = None
        L.rh_shard_chunks.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.rh_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(RhOpts),
                                C.POINTER(ArrowArray), C.POINTER(C.c_uint32), C.POINTER(RhStats), C.POINTER(C.c_char_p)]
        L.rh_decode_packed.argtypes = L.rh_decode.argtypes
        L.rh_decode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                       C.POINTER(RhOpts), C.POINTER(C.c_void_p), C.POINTER(RhStats),
                                       C.POINTER(C.c_char_p)]
        L.rh_device_result_wait.argtypes = [C.c_void_p, C.POINTER(RhStats), C.POINTER(C.c_char_p)]
        L.rh_device_result_chunks.restype = C.c_uint32
        L.rh_device_result_chunks.argtypes = [C.c_void_p]
        L.rh_device_result_output_bytes.restype = C.c_uint64
        L.rh_device_result_output_bytes.argtypes = [C.c_void_p]
        L.rh_device_result_export.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(ArrowDeviceArray)]
        L.rh_device_result_to_host.argtypes = [C.c_void_p, C.POINTER(ArrowArray), C.POINTER(C.c_char_p)]
        L.rh_device_result_free.argtypes = [C.c_void_p]
        L.rh_device_result_buffers.restype = C.c_uint32
        L.rh_device_result_buffers.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32]
        L.rh_free_string.argtypes = [C.c_void_p]
        L.rh_schema_kernel_source.restype = C.c_void_p
        L.rh_schema_kernel_source.argtypes = [C.c_void_p]
        L.rh_schema_encode_kernel_source.restype = C.c_void_p
        L.rh_schema_encode_kernel_source.argtypes = [C.c_void_p]
        L.rh_schema_kernel_key.restype = C.c_void_p
        L.rh_schema_kernel_key.argtypes = [C.c_void_p, C.c_int]
        L.rh_schema_prebuild.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_char_p)]
        L.rh_schema_kernels_ready.restype = C.c_int
        L.rh_schema_kernels_ready.argtypes = [C.c_void_p, C.c_int, C.c_long, C.POINTER(C.c_char_p)]
        L.rh_abi_version.restype = C.c_int
        L.rh_current_device.restype = C.c_int
        L.rh_device_count.restype = C.c_int
        L.rh_encode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(RhOpts), C.POINTER(C.c_void_p),
                                       C.POINTER(RhStats), C.POINTER(C.c_char_p)]
        L.rh_device_encoded_chunks.restype = C.c_uint32
        L.rh_device_encoded_chunks.argtypes = [C.c_void_p]
        L.rh_device_encoded_output_bytes.restype = C.c_uint64
        L.rh_device_encoded_output_bytes.argtypes = [C.c_void_p]
        L.rh_device_encoded_export.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(ArrowDeviceArray)]
        L.rh_device_encoded_to_host.argtypes = [C.c_void_p, C.POINTER(ArrowArray), C.POINTER(C.c_char_p)]
        L.rh_device_encoded_free.argtypes = [C.c_void_p]
        L.rh_engine_counters.restype = C.c_uint32
        L.rh_engine_counters.argtypes = [C.POINTER(C.c_uint64), C.c_uint32]
        _lib = L
    return _lib


ENGINE_COUNTERS = ("fused_calls", "two_sync_calls", "capacity_retries", "wide_fallbacks", "offset32_errors", "split_calls", "single_pass_calls", "single_pass_failovers", "background_compiles",
                   "tiles", "careful_tiles", "over_window_tiles", "rewalked_waves", "subtiled_tiles", "ranged_retries")


def engine_counters() -> dict:
    """Process-wide counters of the engine's rarely taken branches (rh_engine_counters), by name."""
    buf = (C.c_uint64 * len(ENGINE_COUNTERS))()
    n = lib().rh_engine_counters(buf, len(ENGINE_COUNTERS))
    assert n == len(ENGINE_COUNTERS)
    return {name: int(buf[i]) for i, name in enumerate(ENGINE_COUNTERS)}


def _take_err(err: C.c_char_p) -> str:
    msg = err.value.decode("utf-8", "replace") if err.value else "ruhvro_hip error"
    if err.value is not None:
        lib().rh_free_string(C.cast(err, C.c_void_p))
    return msg


def _raise(rc: int, err: C.c_char_p):
    msg = _take_err(err)
    if rc in (RH_ERR_SCHEMA, RH_ERR_DECODE, RH_ERR_ARGUMENT):
        raise ValueError(msg)
    raise RuntimeError(msg)


class Schema:
    """Compiled schema handle (rh_schema*) + the pyarrow schema of its batches."""

    _cache: dict = {}

    def __init__(self, schema_json: str):
        L = lib()
        raw = schema_json.encode()
        err = C.c_char_p()
        self.handle = L.rh_schema_compile(raw, len(raw), C.byref(err))
        if not self.handle:
            _raise(RH_ERR_SCHEMA, err)
        cs = ArrowSchema()
        if L.rh_schema_export(self.handle, C.byref(cs)) != RH_OK:
            raise RuntimeError("rh_schema_export failed")
        st = pa.DataType._import_from_c(C.addressof(cs))
        self.arrow_schema = pa.schema(list(st))

    @classmethod
    def get(cls, schema_json: str) -> "Schema":
        s = cls._cache.get(schema_json)
        if s is None:
            s = cls._cache[schema_json] = cls(schema_json)
        return s


def _import_chunks(arr, k: int, schema: pa.Schema) -> List[pa.RecordBatch]:
    return [pa.RecordBatch._import_from_c(C.addressof(arr[i]), schema) for i in range(k)]


KERNEL_AUTO, KERNEL_GENERIC, KERNEL_SPECIALIZED = 0, 1, 2
RH_TWO_PASS = 16  # rh_opts.flags: rh_decode_device takes the two-pass form (size pass, scan, emit pass) even where the single-pass form would run
RH_SINGLE_PASS = 32  # rh_opts.flags: rh_decode_device prefers the single-pass form (one kernel sizes, scans across tiles, emits)
RH_ASYNC = 8      # rh_opts.flags: rh_decode_device returns once the call is on the stream (rh_device_result_wait settles it)


def kernel_source(schema_json: str) -> str:
    """HIP source of the schema-specialised kernels (rh_schema_kernel_source)."""
    L = lib()
    p = L.rh_schema_kernel_source(Schema.get(schema_json).handle)
    if not p:
        raise RuntimeError("rh_schema_kernel_source failed")
    try:
        return C.string_at(p).decode()
    finally:
        L.rh_free_string(p)


def kernel_key(schema_json: str, encode: bool = False) -> str:
    """Content hash of the schema's specialised kernel pair (rh_schema_kernel_key): the kernel-cache key."""
    L = lib()
    p = L.rh_schema_kernel_key(Schema.get(schema_json).handle, 1 if encode else 0)
    if not p:
        raise RuntimeError("rh_schema_kernel_key failed")
    try:
        return C.string_at(p).decode()
    finally:
        L.rh_free_string(p)


def encode_kernel_source(schema_json: str) -> str:
    """HIP source of the schema-specialised Arrow -> Avro kernels (rh_schema_encode_kernel_source)."""
    L = lib()
    p = L.rh_schema_encode_kernel_source(Schema.get(schema_json).handle)
    if not p:
        raise RuntimeError("rh_schema_encode_kernel_source failed")
    try:
        return C.string_at(p).decode()
    finally:
        L.rh_free_string(p)


def prebuild(schema_json: str) -> bool:
    """Compile the schema-specialised kernels into the on-disk kernel cache (hiprtc; no GPU needed).
    Returns True when the code object was already cached."""
    L = lib()
    cached = C.c_int()
    err = C.c_char_p()
    rc = L.rh_schema_prebuild(Schema.get(schema_json).handle, C.byref(cached), C.byref(err))
    if rc != RH_OK:
        _raise(rc, err)
    return bool(cached.value)


def kernels_ready(schema_json: str, encode: bool = False, timeout_ms: int = 0) -> bool:
    """rh_schema_kernels_ready: True when the schema's specialised kernels are there (the next call runs on them), False while
    their compile jobs are still running after `timeout_ms` (or nobody asked for them); raises when the compile failed."""
    err = C.c_char_p()
    rc = lib().rh_schema_kernels_ready(Schema.get(schema_json).handle, 1 if encode else 0, int(timeout_ms), C.byref(err))
    if rc < 0:
        _raise(RH_ERR_RUNTIME, err)
    return rc == 1


def shard_chunks(n: int, num_chunks: int, n_shards: int, shard: int):
    """rh_shard_chunks -> (chunk_lo, chunk_hi, row_lo, row_hi): the multi-GPU deal of whole reference chunks."""
    c0, c1, r0, r1 = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_uint64()
    lib().rh_shard_chunks(n, num_chunks, n_shards, shard, C.byref(c0), C.byref(c1), C.byref(r0), C.byref(r1))
    return c0.value, c1.value, r0.value, r1.value


def decode_packed(data: np.ndarray, offsets: np.ndarray, schema_json: str, num_chunks: int,
                  device: int = -1, want_stats: bool = False, kernel: int = KERNEL_AUTO, devices=None):
    """rh_decode_packed: one contiguous payload + u64 offsets (host memory) -> list[RecordBatch].
    `devices`: shard the chunks over these HIP devices (rh_opts.devices); with want_stats the stats dict then carries
    the per-shard stats under "device_stats"."""
    L = lib()
    s = Schema.get(schema_json)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    k = L.rh_clamp_chunks(n, num_chunks)
    arr = (ArrowArray * k)()
    out_k = C.c_uint32()
    st = RhStats()
    err = C.c_char_p()
    opts, keep = make_opts(device, kernel, None, devices)
    rc = L.rh_decode_packed(s.handle, data.ctypes.data, offsets.ctypes.data, n, num_chunks, C.byref(opts), arr,
                            C.byref(out_k), C.byref(st), C.byref(err))
    if rc != RH_OK:
        _raise(rc, err)
    out = _import_chunks(arr, out_k.value, s.arrow_schema)
    if not want_stats:
        return out
    d = st.as_dict()
    if devices:
        d["device_stats"] = [x.as_dict() for x in keep["device_stats"]]
    return out, d


def decode_slices(ptrs: np.ndarray, lens: np.ndarray, schema_json: str, num_chunks: int,
                  device: int = -1, want_stats: bool = False, kernel: int = KERNEL_AUTO, devices=None):
    """rh_decode: one (pointer, length) pair per record, the form the CPython boundary extracts from list[bytes]
    (u64 addresses / u64 lengths; the caller keeps the pointed-to memory alive) -> list[RecordBatch]."""
    L = lib()
    s = Schema.get(schema_json)
    ptrs = np.ascontiguousarray(ptrs, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint64)
    n = len(ptrs)
    k = L.rh_clamp_chunks(n, num_chunks)
    arr = (ArrowArray * k)()
    out_k = C.c_uint32()
    st = RhStats()
    err = C.c_char_p()
    opts, keep = make_opts(device, kernel, None, devices)
    rc = L.rh_decode(s.handle, ptrs.ctypes.data, lens.ctypes.data, n, num_chunks, C.byref(opts), arr,
                     C.byref(out_k), C.byref(st), C.byref(err))
    if rc != RH_OK:
        _raise(rc, err)
    out = _import_chunks(arr, out_k.value, s.arrow_schema)
    return (out, st.as_dict()) if want_stats else out


class DeviceResult:
    """Owns an rh_device_result (Arrow buffers resident in HBM)."""

    def __init__(self, handle, schema: Schema, stats: dict):
        self.handle = handle
        self.schema = schema
        self.stats = stats

    @property
    def chunks(self) -> int:
        return lib().rh_device_result_chunks(self.handle)

    @property
    def output_bytes(self) -> int:
        return lib().rh_device_result_output_bytes(self.handle)

    def export(self, chunk: int) -> "ArrowDeviceArray":
        """Chunk `chunk` as an ArrowDeviceArray view (device pointers); release it through array.release."""
        out = ArrowDeviceArray()
        if lib().rh_device_result_export(self.handle, chunk, C.byref(out)) != RH_OK:
            raise RuntimeError("rh_device_result_export failed")
        return out

    def wait(self) -> "DeviceResult":
        """rh_device_result_wait: settle an asynchronous call (raises what the synchronous call would have raised)."""
        st = RhStats()
        err = C.c_char_p()
        rc = lib().rh_device_result_wait(self.handle, C.byref(st), C.byref(err))
        if rc != RH_OK:
            _raise(rc, err)
        if getattr(self, "_want_stats", False) and st.records:
            self.stats = st.as_dict()
        return self

    def to_host(self) -> List[pa.RecordBatch]:
        k = self.chunks
        arr = (ArrowArray * k)()
        err = C.c_char_p()
        rc = lib().rh_device_result_to_host(self.handle, arr, C.byref(err))
        if rc != RH_OK:
            _raise(rc, err)
        return _import_chunks(arr, k, self.schema.arrow_schema)

    def free(self):
        if self.handle:
            lib().rh_device_result_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceEncoded:
    """Owns an rh_device_encoded (k BinaryArrays of Avro datums resident in HBM)."""

    def __init__(self, handle, stats: dict):
        self.handle = handle
        self.stats = stats

    @property
    def chunks(self) -> int:
        return lib().rh_device_encoded_chunks(self.handle)

    @property
    def output_bytes(self) -> int:
        return lib().rh_device_encoded_output_bytes(self.handle)

    def export(self, chunk: int) -> "ArrowDeviceArray":
        out = ArrowDeviceArray()
        if lib().rh_device_encoded_export(self.handle, chunk, C.byref(out)) != RH_OK:
            raise RuntimeError("rh_device_encoded_export failed")
        return out

    def to_host(self) -> List[pa.Array]:
        k = self.chunks
        arr = (ArrowArray * k)()
        err = C.c_char_p()
        rc = lib().rh_device_encoded_to_host(self.handle, arr, C.byref(err))
        if rc != RH_OK:
            _raise(rc, err)
        return [pa.Array._import_from_c(C.addressof(arr[i]), pa.binary()) for i in range(k)]

    def free(self):
        if self.handle:
            lib().rh_device_encoded_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def schema_struct(schema_json: str) -> "ArrowSchema":
    """The schema's batches as an Arrow C ArrowSchema ("+s" struct; the caller releases it)."""
    cs = ArrowSchema()
    if lib().rh_schema_export(Schema.get(schema_json).handle, C.byref(cs)) != RH_OK:
        raise RuntimeError("rh_schema_export failed")
    return cs


def encode_device(batch_array_addr: int, batch_schema_addr: int, schema_json: str, num_chunks: int, device: int = -1,
                  stream: int = 0, want_stats: bool = True, kernel: int = KERNEL_AUTO) -> DeviceEncoded:
    """rh_encode_device: `batch_array_addr` is the address of a struct ArrowArray whose buffer pointers are DEVICE
    pointers (e.g. ctypes.addressof(device_result_export.array)); `batch_schema_addr` of its ArrowSchema."""
    L = lib()
    s = Schema.get(schema_json)
    out = C.c_void_p()
    st = RhStats()
    err = C.c_char_p()
    opts, _keep = make_opts(device, kernel, stream, None, 0)
    rc = L.rh_encode_device(s.handle, batch_array_addr, batch_schema_addr, num_chunks, C.byref(opts), C.byref(out),
                            C.byref(st) if want_stats else None, C.byref(err))
    if rc != RH_OK:
        _raise(rc, err)
    return DeviceEncoded(out.value, st.as_dict())


def decode_device(d_data: int, d_offsets: int, data_len: int, n: int, schema_json: str, num_chunks: int,
                  device: int = -1, stream: int = 0, want_stats: bool = True, kernel: int = KERNEL_AUTO,
                  chunk_rows: int = 0, asynchronous: bool = False, two_pass: bool = False, single_pass: bool = False) -> DeviceResult:
    """rh_decode_device on raw device pointers (e.g. torch tensors' data_ptr()).  chunk_rows: explicit geometry
    for a range of a larger call's chunks (rh_opts.chunk_rows).  asynchronous: RH_ASYNC -- the result is returned
    unsettled (DeviceResult.wait() settles it and fills .stats; every accessor settles implicitly)."""
    L = lib()
    s = Schema.get(schema_json)
    out = C.c_void_p()
    st = RhStats()
    err = C.c_char_p()
    opts, _keep = make_opts(device, kernel | (RH_ASYNC if asynchronous else 0) | (RH_TWO_PASS if two_pass else 0) |
                            (RH_SINGLE_PASS if single_pass else 0), stream, None, chunk_rows)
    rc = L.rh_decode_device(s.handle, d_data, d_offsets, data_len, n, num_chunks, C.byref(opts), C.byref(out),
                            C.byref(st) if want_stats else None, C.byref(err))
    if rc != RH_OK:
        _raise(rc, err)
    r = DeviceResult(out.value, s, st.as_dict())
    r._want_stats = want_stats
    return r


class PreparedDeviceDecode:
    """One rh_decode_device call with every ctypes argument built ONCE: a loop that repeats the call pays the C entry
    point only (what a C caller pays), not ~15 us of Python argument marshalling per call -- which is the size of a
    whole 1M-record call's fixed cost.  run() -> opaque result handle (free it with free()); `stats` is refilled by
    every run(want_stats=True)."""

    def __init__(self, d_data: int, d_offsets: int, data_len: int, n: int, schema_json: str, num_chunks: int,
                 device: int = -1, stream: int = 0, kernel: int = KERNEL_AUTO, chunk_rows: int = 0, asynchronous: bool = False,
                 two_pass: bool = False, single_pass: bool = False):
        self._L = lib()
        self._schema = Schema.get(schema_json)
        self._opts, self._keep = make_opts(device, kernel | (RH_ASYNC if asynchronous else 0) | (RH_TWO_PASS if two_pass else 0) |
                                           (RH_SINGLE_PASS if single_pass else 0), stream, None, chunk_rows)
        self._wait = self._L.rh_device_result_wait
        self.stats = RhStats()
        self._out = C.c_void_p()
        self._err = C.c_char_p()
        self._args = (self._schema.handle, C.c_void_p(d_data), C.c_void_p(d_offsets), C.c_uint64(data_len), C.c_uint64(n),
                      C.c_uint64(num_chunks), C.byref(self._opts), C.byref(self._out))
        self._st = C.byref(self.stats)
        self._e = C.byref(self._err)
        self._fn = self._L.rh_decode_device
        self._free = self._L.rh_device_result_free

    def run(self, want_stats: bool = False) -> int:
        rc = self._fn(*self._args, self._st if want_stats else None, self._e)
        if rc != RH_OK:
            _raise(rc, self._err)
        return self._out.value

    def wait(self, handle: int, want_stats: bool = False) -> None:
        """Settle an asynchronous run() (rh_device_result_wait): raises what the synchronous call would have raised;
        `stats` is refilled when the run asked for them."""
        rc = self._wait(handle, self._st if want_stats else None, self._e)
        if rc != RH_OK:
            _raise(rc, self._err)

    def output_bytes(self, handle: int) -> int:
        return int(self._L.rh_device_result_output_bytes(handle))

    def free(self, handle: int) -> None:
        self._free(handle)


    def to_host(self, handle: int) -> List[pa.RecordBatch]:
        """rh_device_result_to_host of a run()'s result (settles an asynchronous one first); the handle stays the caller's
        to free."""
        k = int(self._L.rh_device_result_chunks(handle))
        arr = (ArrowArray * k)()
        rc = self._L.rh_device_result_to_host(handle, arr, self._e)
        if rc != RH_OK:
            _raise(rc, self._err)
        return _import_chunks(arr, k, self._schema.arrow_schema)
