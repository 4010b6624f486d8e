"""roctx ranges around the phases of the training loop so that a profiler can attribute kernels to collect / GAE /
update instead of by kernel name only.  Nothing here is needed for results; without a ROCTx library every call is a no-op.

Which library (round 5; the cause of round 4's empty marker files, profiles/r05_roctx.md): `rocprofv3 --marker-trace` LD_PRELOADs
/opt/rocm/lib/librocprofiler-sdk-roctx.so into the profiled program (the launcher script says so: "to override the roctx symbols of an
app linked to the old roctracer roctx"), i.e. the profiler's own roctxRangePushA / roctxRangePop are ALREADY in the process's global
symbol namespace, registered with the profiler by the library's constructor before Python starts.  Round 4 loaded roctracer's
libroctx64.so through ctypes and called the functions of THAT handle: a dlsym on a handle returns the handle's own symbol, the
preloaded override is bypassed, rocprofv3 never sees a range and writes no marker file.  So the order here is:
  1. the global namespace (ctypes.CDLL(None) = dlsym(RTLD_DEFAULT)): no dlopen at all, finds the profiler's preloaded library when
     there is one (checked on the CPU box: `rocprofv3 --marker-trace --stats --output-format csv -- python3 x.py` writes
     *_marker_api_stats.csv with the pushed names, before and after `import torch`);
  2. roctracer's libroctx64.so by name (tools that intercept that one); INVERSUS_ROCTX_LIB names another library;
  INVERSUS_NO_ROCTX=1 switches the ranges off.
Resolution happens once, at the import of this module (a symbol lookup; nothing is loaded into a process that a profiler has
already set up)."""
import contextlib
import ctypes as C
import os

_lib = None
_tried = False
source = None          # where the functions came from (for the profile scripts' logs)


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("INVERSUS_NO_ROCTX") == "1":
        return None
    names = (None, "libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so")   # None = the global namespace (a preloaded profiler library)
    if os.environ.get("INVERSUS_ROCTX_LIB"):
        names = (os.environ["INVERSUS_ROCTX_LIB"],)
    for name in names:
        try:
            lib = C.CDLL(name)
            lib.roctxRangePushA.argtypes = [C.c_char_p]
            lib.roctxRangePushA.restype = C.c_int
            lib.roctxRangePop.restype = C.c_int
            _lib = lib
            global source
            source = "global namespace (preloaded)" if name is None else name
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


_load()
