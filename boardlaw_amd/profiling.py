"""Range markers on the hot path's host functions (rebar/profiling.py:15-28: `@profiling.nvtx` on MCTS.descend / backup /
simulate / root and MCTSAgent.__call__, switched on by EMIT_NVTX=1).  On ROCm the markers are roctx ranges
(libroctx64.so: roctxRangePushA / roctxRangePop), which `rocprofv3 --marker-trace` records next to the kernel trace:

    EMIT_ROCTX=1 rocprofv3 --kernel-trace --marker-trace -- python bench.py --eager --steps 2

(EMIT_NVTX=1 is honoured too, so the reference's switch keeps working.)  Off by default: the decorator then returns the
function itself, so the search pays nothing.  A captured move (graph=True) is one replay call on the host: its range
wraps the replay, the per-simulation ranges exist in eager mode only."""
import ctypes
import functools
import os

_lib = None


def enabled():
    return os.environ.get('EMIT_ROCTX') == '1' or os.environ.get('EMIT_NVTX') == '1'


def _roctx():
    global _lib
    if _lib is None:
        for name in ('libroctx64.so', 'libroctx64.so.4', '/opt/rocm/lib/libroctx64.so'):
            try:
                _lib = ctypes.CDLL(name)
                break
            except OSError:
                continue
        else:
            raise RuntimeError('EMIT_ROCTX=1 but libroctx64.so could not be loaded')
        _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
        _lib.roctxRangePushA.restype = ctypes.c_int
        _lib.roctxRangePop.restype = ctypes.c_int
    return _lib


def push(name):
    return _roctx().roctxRangePushA(name.encode())


def pop():
    return _roctx().roctxRangePop()


def roctx(f):
    """Decorator: a roctx range named module.qualname around every call, when EMIT_ROCTX=1 (or EMIT_NVTX=1) was set at
    import time -- the reference's `nvtx` decorator."""
    if not enabled():
        return f
    name = f'{f.__module__}.{f.__qualname__}'

    @functools.wraps(f)
    def g(*args, **kwargs):
        push(name)
        try:
            return f(*args, **kwargs)
        finally:
            pop()
    return g


nvtx = roctx      # the reference's name for it
