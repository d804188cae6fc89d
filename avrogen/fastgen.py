"""ctypes front for fastgen.c: (config, seed, start, n) -> packed bytes + u64 offsets."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libfastgen.so")
CFG = {"full": 0, "flat4": 1, "cfg3": 2, "full_realistic": 3, "full_realistic_heavy": 4, "full_skewed": 5, "full_realistic_nogiant": 6,
       "wide97": 100 + 97, "wide200": 100 + 200, "wide400": 100 + 400}
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "fastgen.c")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-shared", "-fPIC", "-pthread", "-Wall", src, "-o", LIB + ".tmp"])
        os.replace(LIB + ".tmp", LIB)
    return LIB


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.fg_generate.restype = C.c_void_p
        _lib.fg_generate.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
        _lib.fg_total_bytes.restype = C.c_uint64
        _lib.fg_total_bytes.argtypes = [C.c_void_p]
        _lib.fg_copy_out.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.fg_free.argtypes = [C.c_void_p]
    return _lib


def generate(name: str, n: int, seed: int = 20260921, start: int = 0, nthreads: int = 0):
    """-> (data: np.uint8[total], offsets: np.uint64[n+1])."""
    lib = _load()
    if nthreads <= 0:
        nthreads = min(os.cpu_count() or 1, 32)
    h = lib.fg_generate(CFG[name], seed, start, n, nthreads)
    try:
        total = lib.fg_total_bytes(h)
        data = np.empty(max(int(total), 1), dtype=np.uint8)
        offsets = np.empty(n + 1, dtype=np.uint64)
        lib.fg_copy_out(h, data.ctypes.data, offsets.ctypes.data)
    finally:
        lib.fg_free(h)
    return data[:total], offsets


def split(data: np.ndarray, offsets: np.ndarray):
    """Packed form -> list[bytes] (what the Python surface takes)."""
    b = data.tobytes()
    o = offsets.tolist()
    return [b[o[i]:o[i + 1]] for i in range(len(o) - 1)]
