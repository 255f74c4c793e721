"""Stage ranges for rocprofv3 (`--marker-trace`): roctx push / pop around the stages of the hot path (tracing).

Off unless LANEMAP_ROCTX=1 (read once): the ranges cost two library calls per stage, which is nothing next to ~350 kernel launches per
batch, but the default command stays exactly what the bench times.  The library is librocprofiler-sdk-roctx.so (what rocprofv3
intercepts), libroctx64.so as the fallback; both are in /opt/rocm/lib of the ROCm image.  `tools/r5/stage_stats.py` turns a
`rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace` run into per-stage kernel time restricted to the `timed_steps` range.
"""
import contextlib
import ctypes
import os

ENABLED = os.environ.get('LANEMAP_ROCTX', '0') != '0'
_lib = None


def _load():
    global _lib
    if _lib is None:
        for name in ('librocprofiler-sdk-roctx.so', 'libroctx64.so'):
            for prefix in ('', '/opt/rocm/lib/'):
                try:
                    _lib = ctypes.CDLL(prefix + name)
                    _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    _lib.roctxRangePushA.restype = ctypes.c_int
                    _lib.roctxRangePop.restype = ctypes.c_int
                    _lib.roctxMarkA.argtypes = [ctypes.c_char_p]
                    return _lib
                except (OSError, AttributeError):
                    _lib = None
        raise OSError('LANEMAP_ROCTX=1 but neither librocprofiler-sdk-roctx.so nor libroctx64.so could be loaded')
    return _lib


def push(name):
    if ENABLED:
        _load().roctxRangePushA(name.encode())


def pop():
    if ENABLED:
        _load().roctxRangePop()


def mark(name):
    if ENABLED:
        _load().roctxMarkA(name.encode())


@contextlib.contextmanager
def stage(name):
    """with trace.stage('fpn.layer3'): ...  (a no-op unless LANEMAP_ROCTX=1)"""
    if not ENABLED:
        yield
        return
    _load().roctxRangePushA(name.encode())
    try:
        yield
    finally:
        _lib.roctxRangePop()
