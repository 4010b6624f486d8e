"""roctx ranges around the phases of the training loop so that a profiler can attribute kernels to collect / GAE /
update instead of by kernel name only.  libroctx64.so (roctracer's) is loaded through ctypes when it is there (/opt/rocm/lib); without
it every call is a no-op -- nothing here is needed for results.

What this image's rocprofv3 does with them (round 4, profiles/r04_incidents.md): with libroctx64.so a `--kernel-trace --marker-trace`
run completes but writes NO marker file (rocprofiler-sdk only sees ranges pushed through its own librocprofiler-sdk-roctx.so); with that
library loaded here instead -- through ctypes, after torch has initialised HIP -- the one profiled run of tools/agent_update_profile.py
stopped producing output and was killed by the 7-minute watchdog.  So the default stays libroctx64.so; INVERSUS_ROCTX_LIB names another
library for whoever wants to try a different tool, and INVERSUS_NO_ROCTX=1 switches the ranges off."""
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
    names = ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so")
    if os.environ.get("INVERSUS_ROCTX_LIB"):
        names = (os.environ["INVERSUS_ROCTX_LIB"],)
    for name in names:
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
