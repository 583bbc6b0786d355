"""The few HIP runtime calls the GPU tests need for device buffers and streams of their own, through ctypes on the
runtime the product library is linked against (so that a test process holds one HIP runtime, not PyTorch's as well)."""
import ctypes as C

import numpy as np



def _runtime_path():
    """The HIP runtime the product library brought into this process (whatever its version), else the plain soname."""
    import raytracing_rust_amd  # noqa: F401  (loads librtx_amd.so and with it the runtime it is linked against)
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split()[-1]
                if "/libamdhip64.so" in path:
                    return path
    except OSError:
        pass
    return "libamdhip64.so"


_hip = C.CDLL(_runtime_path())
for _name, _args in (("hipMalloc", [C.POINTER(C.c_void_p), C.c_size_t]), ("hipFree", [C.c_void_p]),
                     ("hipStreamBeginCapture", [C.c_void_p, C.c_int]), ("hipStreamEndCapture", [C.c_void_p, C.POINTER(C.c_void_p)]),
                     ("hipGraphInstantiate", [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
                     ("hipGraphLaunch", [C.c_void_p, C.c_void_p]), ("hipGraphExecDestroy", [C.c_void_p]), ("hipGraphDestroy", [C.c_void_p]),
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


class CapturedGraph:
    """Work enqueued on `stream` inside the `with` block is captured into a HIP graph instead of running; launch()
    then runs the whole graph on a stream."""

    def __init__(self, stream):
        self.stream, self.graph, self.exe = stream, C.c_void_p(), C.c_void_p()

    def __enter__(self):
        _ok(_hip.hipStreamBeginCapture(self.stream.handle, 1), "hipStreamBeginCapture")   # 1 = thread-local mode
        return self

    def __exit__(self, exc_type, exc, tb):
        st = _hip.hipStreamEndCapture(self.stream.handle, C.byref(self.graph))
        if exc_type is None:
            _ok(st, "hipStreamEndCapture")
            _ok(_hip.hipGraphInstantiate(C.byref(self.exe), self.graph, None, None, 0), "hipGraphInstantiate")
        return False

    def launch(self, stream=None):
        _ok(_hip.hipGraphLaunch(self.exe, (stream or self.stream).handle), "hipGraphLaunch")

    def close(self):
        if self.exe:
            _hip.hipGraphExecDestroy(self.exe)
            self.exe = C.c_void_p()
        if self.graph:
            _hip.hipGraphDestroy(self.graph)
            self.graph = C.c_void_p()
