"""The few HIP runtime calls the GPU tests need for device buffers and streams of their own, through ctypes on the
runtime the product library is linked against (so that a test process holds one HIP runtime, not PyTorch's as well)."""
import ctypes as C

import numpy as np

_hip = C.CDLL("libamdhip64.so.7")
for _name, _args in (("hipMalloc", [C.POINTER(C.c_void_p), C.c_size_t]), ("hipFree", [C.c_void_p]),
                     ("hipMemset", [C.c_void_p, C.c_int, C.c_size_t]),
                     ("hipMemcpy", [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
                     ("hipDeviceSynchronize", []), ("hipStreamCreate", [C.POINTER(C.c_void_p)]),
                     ("hipStreamDestroy", [C.c_void_p]), ("hipStreamSynchronize", [C.c_void_p])):
    _fn = getattr(_hip, _name)
    _fn.restype, _fn.argtypes = C.c_int, _args


def _ok(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed with HIP error {status}")


def synchronize():
    _ok(_hip.hipDeviceSynchronize(), "hipDeviceSynchronize")


class DeviceArray:
    """n zeroed doubles in device memory."""

    def __init__(self, n):
        self.n = int(n)
        self.ptr = C.c_void_p()
        _ok(_hip.hipMalloc(C.byref(self.ptr), self.n * 8), "hipMalloc")
        self.zero()

    def zero(self):
        _ok(_hip.hipMemset(self.ptr, 0, self.n * 8), "hipMemset")

    def numpy(self):
        synchronize()
        out = np.empty(self.n)
        _ok(_hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, self.n * 8, 2), "hipMemcpy")   # 2 = device to host
        return out

    def free(self):
        if self.ptr:
            synchronize()
            _hip.hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream:
    def __init__(self):
        self.handle = C.c_void_p()
        _ok(_hip.hipStreamCreate(C.byref(self.handle)), "hipStreamCreate")

    def close(self):
        if self.handle:
            _hip.hipStreamSynchronize(self.handle)
            _hip.hipStreamDestroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
