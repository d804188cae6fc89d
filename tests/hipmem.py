"""Device memory for the GPU tests without PyTorch: ctypes on the HIP runtime the engine itself is bound to
(libruhvro_hip.so is loaded first, so dlopen by SONAME returns that same runtime)."""
import ctypes as C

import numpy as np

from pyruhvro_amd import cabi

_hip = None


def hip():
    global _hip
    if _hip is None:
        cabi.lib()
        L = C.CDLL("libamdhip64.so.7")
        L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.hipFree.argtypes = [C.c_void_p]
        L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.hipPointerGetAttributes.argtypes = [C.c_void_p, C.c_void_p]
        L.hipSetDevice.argtypes = [C.c_int]
        _hip = L
    return _hip


class DevBuf:
    """hipMalloc'd bytes (zero-filled), freed with the object."""

    def __init__(self, nbytes: int, device: int = 0):
        L = hip()
        assert L.hipSetDevice(device) == 0
        self.n = max(int(nbytes), 16)
        p = C.c_void_p()
        assert L.hipMalloc(C.byref(p), self.n) == 0
        self.ptr = p.value
        assert L.hipMemset(self.ptr, 0, self.n) == 0

    def upload(self, arr: np.ndarray, offset: int = 0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.n
        if arr.nbytes:
            assert hip().hipMemcpy(self.ptr + offset, arr.ctypes.data, arr.nbytes, 1) == 0      # HostToDevice
        return self

    def __del__(self):
        try:
            if self.ptr:
                hip().hipFree(self.ptr)
                self.ptr = None
        except Exception:
            pass


def d2h(dptr: int, n: int) -> np.ndarray:
    out = np.zeros(max(n, 1), dtype=np.uint8)
    if n:
        assert dptr, "null device buffer"
        assert hip().hipMemcpy(out.ctypes.data, dptr, n, 2) == 0                                  # DeviceToHost
    return out[:n]


def upload_packed(data: np.ndarray, offsets: np.ndarray, device: int = 0):
    """(payload DevBuf with 64 bytes of slack, offsets DevBuf) for rh_decode_device."""
    d = DevBuf(len(data) + 64, device).upload(np.asarray(data, dtype=np.uint8))
    o = DevBuf(8 * len(offsets), device).upload(np.asarray(offsets, dtype=np.uint64))
    return d, o
