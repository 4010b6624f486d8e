"""roctx ranges around the phases of the training loop: `rocprofv3 --kernel-trace --marker-trace` then attributes
kernels to collect / GAE / update instead of by kernel name only.  libroctx64.so is loaded through ctypes when it is there
(/opt/rocm/lib); without it every call is a no-op -- nothing here is needed for results."""
import contextlib
import ctypes as C
import os

_lib = None
_tried = False


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("INVERSUS_NO_ROCTX") == "1":
        return None
    for name in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so"):
        try:
            lib = C.CDLL(name)
            lib.roctxRangePushA.argtypes = [C.c_char_p]
            lib.roctxRangePushA.restype = C.c_int
            lib.roctxRangePop.restype = C.c_int
            _lib = lib
            break
        except (OSError, AttributeError):
            continue
    return _lib


def available():
    return _load() is not None


def push(name):
    lib = _load()
    if lib is not None:
        lib.roctxRangePushA(name.encode())


def pop():
    lib = _load()
    if lib is not None:
        lib.roctxRangePop()


@contextlib.contextmanager
def range(name):                                           # noqa: A001 (mirrors roctxRange)
    push(name)
    try:
        yield
    finally:
        pop()
