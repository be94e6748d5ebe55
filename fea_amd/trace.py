"""roctx ranges for rocprofv3 (`rocprofv3 --marker-trace --kernel-trace --stats -- python3 bench.py ...`).

Off by default and then free: `traced()` returns the function it is given.  FEA_AMD_ROCTX=1 brackets the phases of a
training step (forward, backward, optimiser step, gradient buckets); FEA_AMD_ROCTX=2 also brackets every C-ABI call with
its entry-point name, so that a kernel trace can be grouped by the call that launched it.  The reference has no tracing
(SURVEY.md section 5); the range names follow its loop (train_direct.py:96-109)."""
import contextlib
import ctypes
import functools
import os

LEVEL = int(os.environ.get("FEA_AMD_ROCTX", "0") or 0)
_push = _pop = None
if LEVEL:
    for _name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        try:
            _l = ctypes.CDLL(_name)
            _push, _pop = _l.roctxRangePushA, _l.roctxRangePop
            _push.argtypes, _push.restype, _pop.restype = [ctypes.c_char_p], ctypes.c_int, ctypes.c_int
            break
        except (OSError, AttributeError):
            continue
    if _push is None:
        raise ImportError("FEA_AMD_ROCTX is set but neither librocprofiler-sdk-roctx.so nor libroctx64.so could be loaded")

CALLS = LEVEL >= 2


def push(name: str):
    if _push is not None:
        _push(name.encode())


def pop():
    if _pop is not None:
        _pop()


@contextlib.contextmanager
def span(name: str):
    push(name)
    try:
        yield
    finally:
        pop()


def traced(name: str):
    """Decorator: the call runs inside a roctx range `name` (identity when tracing is off)."""
    def deco(fn):
        if not LEVEL:
            return fn

        @functools.wraps(fn)
        def wrapper(*a, **k):
            push(name)
            try:
                return fn(*a, **k)
            finally:
                pop()
        return wrapper
    return deco
