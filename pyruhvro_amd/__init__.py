"""pyruhvro_amd -- MI355X-native Avro -> Arrow direct decode behind pyruhvro's Python surface.

Same function names, positional signatures, return types and error behaviour
as the reference's PyO3 module (``/src/lib.rs:56-158`` of Tyler-Sch/pyruhvro):

    deserialize_array(list, schema) -> pyarrow.RecordBatch
    deserialize_array_threaded(list, schema, num_chunks) -> list[pyarrow.RecordBatch]
    deserialize_array_threaded_spawn(list, schema, num_chunks) -> list[pyarrow.RecordBatch]
    serialize_record_batch(batch, schema, num_chunks) -> list[pyarrow.BinaryArray]   (Arrow -> Avro, also on the GPU)
    serialize_record_batch_spawn(...)                                                 same

Several GPUs: ``set_devices([0, 1, ...])`` (or ``PYRUHVRO_DEVICES=0,1,...`` in the environment) deals the
``num_chunks`` output chunks to those devices in contiguous runs, so every returned batch is produced by one GPU and
the result does not depend on how many there are (the reference deals its chunks to a thread pool the same way,
``ruhvro/src/deserialize.rs:92-120``).

The decode runs in hand-written HIP kernels on the GPU through the C ABI in
``include/ruhvro_hip.h`` (``libruhvro_hip.so``).  There is no CPU decode path in
this package: without the native library or without a HIP device the calls
raise ``RuntimeError``.

If the process also uses PyTorch-ROCm, import torch BEFORE this package so both
share torch's bundled HIP runtime (same SONAME, first one loaded wins).
"""
from __future__ import annotations

import os
import threading
from typing import List

import pyarrow as pa

_HERE = os.path.dirname(os.path.abspath(__file__))

try:
    from . import _pyruhvro as _native  # noqa: F401
except ImportError as _e:  # pragma: no cover - exercised only on an unbuilt tree
    _native = None
    _native_error = _e


def _require_native():
    if _native is None:
        raise RuntimeError(
            "pyruhvro_amd: native extension not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `python pyruhvro_amd/_build.py`): {_native_error}")
    return _native


# Schema cache keyed by the exact schema string, unbounded -- src/lib.rs:35-54.
_cache_lock = threading.Lock()
_cache: dict = {}


class _Compiled:
    __slots__ = ("capsule", "arrow_schema")

    def __init__(self, capsule, arrow_schema):
        self.capsule = capsule
        self.arrow_schema = arrow_schema


def _get_schema(schema: str) -> _Compiled:
    if not isinstance(schema, str):
        raise TypeError("argument 'schema': expected str")
    with _cache_lock:
        hit = _cache.get(schema)
    if hit is not None:
        return hit
    nat = _require_native()
    cap = nat.compile_schema(schema)          # ValueError on a bad / unsupported schema (src/lib.rs:52)
    addr = nat.export_schema(cap)
    try:
        st = pa.DataType._import_from_c(addr)  # "+s" struct whose fields are the batch columns
    finally:
        nat.free_struct(addr)
    comp = _Compiled(cap, pa.schema(list(st)))
    with _cache_lock:
        return _cache.setdefault(schema, comp)


def arrow_schema(schema: str) -> pa.Schema:
    """Arrow schema the decode produces for this Avro schema (schema_translate.rs:19-37)."""
    return _get_schema(schema).arrow_schema


KERNEL_AUTO, KERNEL_GENERIC, KERNEL_SPECIALIZED = 0, 1, 2
_kernel_mode = KERNEL_AUTO


def set_kernel_mode(mode) -> int:
    """Which HIP kernel form decodes: "auto" (default: schema-specialised kernels for large calls or when the
    code object is cached, generic schema-program interpreter otherwise), "generic" or "specialized".
    Both run on the GPU and produce identical buffers.  Returns the previous mode."""
    global _kernel_mode
    names = {"auto": KERNEL_AUTO, "generic": KERNEL_GENERIC, "specialized": KERNEL_SPECIALIZED}
    new = names[mode] if isinstance(mode, str) else int(mode)
    if new not in (0, 1, 2):
        raise ValueError("kernel mode must be auto, generic or specialized")
    old, _kernel_mode = _kernel_mode, new
    return old


_devices = None      # None: the current HIP device; else the ordinals the chunks are dealt to


def _env_devices():
    e = os.environ.get("PYRUHVRO_DEVICES", "").strip()
    if not e:
        return None
    try:
        return [int(x) for x in e.split(",") if x.strip() != ""]
    except ValueError:
        raise ValueError(f"PYRUHVRO_DEVICES: expected comma-separated device ordinals, got {e!r}") from None


def set_devices(devices):
    """Deal the chunks of every later decode call to these HIP devices (a sequence of ordinals; an ordinal may repeat
    to run several logical shards on one GPU; ``None`` = back to the current device / ``PYRUHVRO_DEVICES``).
    Returns the previous setting."""
    global _devices
    old = _devices
    if devices is None:
        _devices = None
    else:
        devs = [int(d) for d in devices]
        if not devs:
            raise ValueError("set_devices: empty device list")
        _devices = devs
    return old


def _current_devices():
    return _devices if _devices is not None else _env_devices()


def _decode(list_, schema: str, num_chunks: int, want_stats: bool = False, device: int = -1, stream: int = 0):
    comp = _get_schema(schema)
    nat = _require_native()
    if not isinstance(num_chunks, int) or isinstance(num_chunks, bool):
        raise TypeError("argument 'num_chunks': expected int")
    if num_chunks < 0:
        raise OverflowError("can't convert negative int to unsigned")  # usize extraction in PyO3
    devs = _current_devices() if device < 0 and not stream else None
    addrs, stats = nat.decode(comp.capsule, list_, num_chunks, device, stream, want_stats, _kernel_mode, devs)
    out: List[pa.RecordBatch] = []
    try:
        for i, a in enumerate(addrs):
            out.append(pa.RecordBatch._import_from_c(a, comp.arrow_schema))
            nat.free_struct(a)     # content was moved into pyarrow, only the shell is left
            addrs[i] = 0
    finally:
        for a in addrs:
            if a:
                nat.release_array(a)
    return out, stats


def deserialize_array(list, schema):  # noqa: A002 - the reference's parameter name
    """src/lib.rs:56-71 -> ruhvro::deserialize::per_datum_deserialize (deserialize.rs:25-30)."""
    return _decode(list, schema, 1)[0][0]


def deserialize_array_threaded(list, schema, num_chunks):  # noqa: A002
    """src/lib.rs:73-89 -> per_datum_deserialize_threaded (deserialize.rs:76-121):
    ``clamp(num_chunks, 1, max(len(list), 1))`` batches, chunk order preserved."""
    return _decode(list, schema, num_chunks)[0]


def deserialize_array_threaded_spawn(list, schema, num_chunks):  # noqa: A002
    """src/lib.rs:108-128 -- same results as deserialize_array_threaded (deserialize.rs:127-170)."""
    return _decode(list, schema, num_chunks)[0]


def deserialize_array_threaded_with_stats(list, schema, num_chunks, device: int = -1):  # noqa: A002
    """Extension: also returns the engine's per-stage timings (rh_stats)."""
    return _decode(list, schema, num_chunks, want_stats=True, device=device)


def last_decode_profile():
    """Extension: where this thread's most recent ``deserialize_array*`` call spent its time at the CPython boundary --
    set-up, list extraction, the rest of the engine call, the total, and how long the call held the GIL (milliseconds)."""
    _require_native()
    return _native.last_decode_profile()


def deserialize_binary_array(array, schema, num_chunks):
    """Extension: the same decode for records that already sit in an Arrow
    ``BinaryArray`` / ``LargeBinaryArray`` (one record per element) -- the form the reference packs its
    list into internally (deserialize.rs:90).  Zero-copy on the payload: no per-``bytes`` extraction
    under the GIL.  Returns ``list[pyarrow.RecordBatch]`` with the chunking of deserialize_array_threaded."""
    import numpy as np
    from . import cabi
    if isinstance(array, pa.ChunkedArray):
        array = array.combine_chunks() if array.num_chunks != 1 else array.chunk(0)
    if not isinstance(array, (pa.BinaryArray, pa.LargeBinaryArray)):
        raise TypeError("argument 'array': expected a pyarrow BinaryArray or LargeBinaryArray")
    if array.null_count:
        raise ValueError("null records cannot be decoded")
    if not isinstance(num_chunks, int) or isinstance(num_chunks, bool):
        raise TypeError("argument 'num_chunks': expected int")
    if num_chunks < 0:
        raise OverflowError("can't convert negative int to unsigned")
    _get_schema(schema)                                  # ValueError on a bad / unsupported schema, like every entry point
    n = len(array)
    bufs = array.buffers()
    odt = np.int64 if isinstance(array, pa.LargeBinaryArray) else np.int32
    offs = np.frombuffer(bufs[1], dtype=odt, count=array.offset + n + 1)[array.offset:]
    base = int(offs[0]) if n else 0
    end = int(offs[-1]) if n else 0
    data = (np.frombuffer(bufs[2], dtype=np.uint8, count=end)[base:] if bufs[2] is not None and end > base
            else np.zeros(1, dtype=np.uint8))
    offsets = (offs.astype(np.uint64) - np.uint64(base)) if n else np.zeros(1, dtype=np.uint64)
    return cabi.decode_packed(data, offsets, schema, num_chunks, kernel=_kernel_mode, devices=_current_devices())


def _encode(data, schema: str, num_chunks: int, want_stats: bool = False, device: int = -1, stream: int = 0):
    import ctypes
    comp = _get_schema(schema)
    nat = _require_native()
    if isinstance(data, pa.RecordBatch):
        sa = data.to_struct_array()
    elif isinstance(data, pa.StructArray):
        sa = data
    else:
        raise TypeError("argument 'data': expected a pyarrow RecordBatch")
    if not isinstance(num_chunks, int) or isinstance(num_chunks, bool):
        raise TypeError("argument 'num_chunks': expected int")
    if num_chunks < 0:
        raise OverflowError("can't convert negative int to unsigned")
    c_arr = ctypes.create_string_buffer(80)        # struct ArrowArray
    c_sch = ctypes.create_string_buffer(72)        # struct ArrowSchema
    sa._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_sch))
    addrs, stats = nat.encode(comp.capsule, ctypes.addressof(c_arr), ctypes.addressof(c_sch), num_chunks,
                              device, stream, want_stats, _kernel_mode)   # releases the two exported structs
    out = []
    try:
        for i, a in enumerate(addrs):
            out.append(pa.Array._import_from_c(a, pa.binary()))
            nat.free_struct(a)
            addrs[i] = 0
    finally:
        for a in addrs:
            if a:
                nat.release_array(a)
    return out, stats


def serialize_record_batch(data, schema, num_chunks):
    """src/lib.rs:91-106 -> ruhvro::serialize::serialize_record_batch (serialize.rs:38-67) with the fast encoder
    (fast_encode.rs:27-53): ``clamp(num_chunks, 1, max(rows, 1))`` BinaryArrays, one Avro datum per row, columns
    matched to the schema's fields by name.  Runs on the GPU (rh_encode)."""
    return _encode(data, schema, num_chunks)[0]


def serialize_record_batch_spawn(data, schema, num_chunks):
    """src/lib.rs:130-148 -- same results as serialize_record_batch (serialize.rs:70-99)."""
    return _encode(data, schema, num_chunks)[0]


def serialize_record_batch_with_stats(data, schema, num_chunks, device: int = -1):
    """Extension: also returns the engine's per-stage timings (rh_stats)."""
    return _encode(data, schema, num_chunks, want_stats=True, device=device)


def device_count() -> int:
    return _require_native().device_count()


def deserialize_to_device(records, schema, num_chunks, device: int = -1, stream: int = 0):
    """Extension: the same decode with the Arrow buffers left in HBM, every buffer a DLPack producer
    (``torch.from_dlpack(dec.batches[0].column("created_at").values)`` is an int64 tensor over the engine's memory, no copy).
    See ``pyruhvro_amd.device``."""
    from .device import deserialize_to_device as f
    return f(records, schema, num_chunks, device=device, stream=stream, kernel=_kernel_mode)


def kernels_ready(schema: str, encode: bool = False, timeout_ms: int = 0) -> bool:
    """Extension.  A schema this process has not met is decoded by the generic kernels at once while the kernels specialised
    to it compile in the background (the reference's cost of a new schema is a JSON parse, ``src/lib.rs:39-54``; a hiprtc
    compile is seconds).  True when they are there -- the next call runs on them; waits up to ``timeout_ms`` for running
    compile jobs.  Raises ``RuntimeError`` if the compile failed (calls keep working on the generic kernels)."""
    return bool(_require_native().kernels_ready(_get_schema(schema).capsule, bool(encode), int(timeout_ms)))


def prebuild(schema: str) -> bool:
    """Extension.  Compile this schema's specialised kernels now (all of them, side by side) and wait: for services that
    want their first batch at full speed.  The code objects land in the kernel cache (``RUHVRO_HIP_KERNEL_CACHE`` or
    ``pyruhvro_amd/_kcache``), where later processes find them.  True when nothing had to be compiled."""
    return bool(_require_native().prebuild(_get_schema(schema).capsule))


__all__ = [
    "deserialize_array", "deserialize_array_threaded", "deserialize_array_threaded_spawn",
    "serialize_record_batch_with_stats",
    "serialize_record_batch", "serialize_record_batch_spawn", "arrow_schema", "device_count", "set_kernel_mode",
    "deserialize_binary_array", "set_devices", "kernels_ready", "prebuild", "deserialize_to_device",
]
