"""Minimal device-memory helper over libamdhip64 (ctypes), for callers that
need device-resident buffers without importing torch (tests, small tools).
bench.py and the multi-GPU path use torch tensors instead."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        # the HIP runtime libzudsmi runs on (one copy per process, see _lib._one_hip_runtime):
        # asking for the SONAME returns the copy that is already loaded
        from . import _lib
        _lib.lib()
        try:
            _hip = C.CDLL('libamdhip64.so.7')
        except OSError:
            _hip = C.CDLL('libamdhip64.so')
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipDeviceSynchronize.argtypes = []
    return _hip


class DeviceBuffer(object):

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        rc = hip().hipMalloc(C.byref(p), max(self.nbytes, 16))
        if rc != 0:
            raise MemoryError(f'hipMalloc({self.nbytes}) failed with {rc}')
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        rc = hip().hipMemcpy(self.ptr, arr.ctypes.data, arr.nbytes, 1)
        if rc != 0:
            raise RuntimeError(f'hipMemcpy H2D failed with {rc}')

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        rc = hip().hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2)
        if rc != 0:
            raise RuntimeError(f'hipMemcpy D2H failed with {rc}')
        return out

    def __del__(self):
        try:
            if self.ptr:
                hip().hipFree(self.ptr)
                self.ptr = None
        except Exception:
            pass
