"""ctypes binding of libjodo_hip.so (the C ABI declared in include/jodo_hip.h).

The product path has no CPU fallback: if the shared library is missing or a symbol cannot be
resolved, importing callers get a RuntimeError that says how to build it (`python -c "import
__graft_entry__ as g; g.build()"`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libjodo_hip.so')
_lib = None


class JodoHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise JodoHipError(
                "libjodo_hip.so not found at %s — the HIP extension is required (no CPU fallback). "
                "Build it with: python -c 'import __graft_entry__ as g; g.build()'" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.jodo_last_error.restype = ctypes.c_char_p
    return _lib


def check(code, what=''):
    if code != 0:
        msg = lib().jodo_last_error()
        raise JodoHipError("%s failed (%d): %s" % (what, code, msg.decode() if msg else ''))


def ptr(t):
    """device/host pointer of a torch tensor (or None) as c_void_p"""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def current_stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
