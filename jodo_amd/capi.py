"""ctypes binding of libjodo_hip.so (the C ABI declared in include/jodo_hip.h).

The product path has no CPU fallback: if the shared library is missing or a symbol cannot be
resolved, importing callers get a RuntimeError that says how to build it (`python -c "import
__graft_entry__ as g; g.build()"`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# JODO_HIP_LIB: A/B experiments of tools/ load another build of the same library (csrc/Makefile LIB=...); tests and the
# driver never set it
LIB_PATH = os.environ.get('JODO_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libjodo_hip.so')
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


class JodoTensor(ctypes.Structure):             # jodo_tensor (include/jodo_hip.h)
    _fields_ = [('name', ctypes.c_char_p), ('data', ctypes.c_void_p), ('shape', ctypes.POINTER(ctypes.c_int64)),
                ('ndim', ctypes.c_int32)]


_PACKED_SIZE = {}


def pack_weights(cfg_struct, state_dict, device=None):
    """state_dict (reference key names; values on any device) -> (blob, woff ctypes int64 array, n_woff) through the
    C packer jodo_dgt_pack_weights[_host].  device=None: blob is a CPU torch tensor; otherwise it is packed straight
    into a device tensor (one host->device copy of the blob inside the call)."""
    import numpy as np
    import torch
    L = lib()
    keep, arr = [], (JodoTensor * len(state_dict))()
    for i, (k, v) in enumerate(state_dict.items()):
        t = np.ascontiguousarray(v.detach().float().cpu().numpy())
        shp = (ctypes.c_int64 * max(t.ndim, 1))(*t.shape)
        name = k.encode()
        keep.append((t, shp, name))
        arr[i] = JodoTensor(name, t.ctypes.data_as(ctypes.c_void_p), shp, t.ndim)
    # the packed size depends on the configuration only; asking for it runs the whole packer (QR factorisations included), so it
    # is asked once per configuration — a missing or mis-sized tensor is still a named error of the packing call below
    key = bytes(cfg_struct)
    n_floats, n_woff = ctypes.c_size_t(), ctypes.c_int()
    if key in _PACKED_SIZE:
        n_floats.value, n_woff.value = _PACKED_SIZE[key]
    else:
        check(L.jodo_dgt_packed_size(ctypes.byref(cfg_struct), arr, len(keep), ctypes.byref(n_floats), ctypes.byref(n_woff)),
              'jodo_dgt_packed_size')
        _PACKED_SIZE[key] = (n_floats.value, n_woff.value)
    woff = (ctypes.c_int64 * n_woff.value)()
    if device is None:
        blob = torch.empty(n_floats.value, dtype=torch.float32)
        check(L.jodo_dgt_pack_weights_host(ctypes.byref(cfg_struct), arr, len(keep), ctypes.c_void_p(blob.data_ptr()),
                                           ctypes.c_size_t(n_floats.value), woff, n_woff.value), 'jodo_dgt_pack_weights_host')
    else:
        blob = torch.empty(n_floats.value, dtype=torch.float32, device=device)
        check(L.jodo_dgt_pack_weights(ctypes.byref(cfg_struct), arr, len(keep), ctypes.c_void_p(blob.data_ptr()),
                                      ctypes.c_size_t(n_floats.value), woff, n_woff.value, current_stream_ptr()),
              'jodo_dgt_pack_weights')
    return blob, woff, n_woff.value


def pack_split_tape(cfg_struct, state_dict, device=None):
    """The static weight tape of the OPT-IN split-bf16 pair update (jodo_dgt_pack_split_host; JODO_OPT_SPLIT_BF16): a uint8 tensor
    (CPU, or uploaded to `device`).  Raises JodoHipError for configurations the split form is not built for (nf != 256, conditional)."""
    import numpy as np
    import torch
    L = lib()
    keep, arr = [], (JodoTensor * len(state_dict))()
    for i, (k, v) in enumerate(state_dict.items()):
        t = np.ascontiguousarray(v.detach().float().cpu().numpy())
        shp = (ctypes.c_int64 * max(t.ndim, 1))(*t.shape)
        name = k.encode()
        keep.append((t, shp, name))
        arr[i] = JodoTensor(name, t.ctypes.data_as(ctypes.c_void_p), shp, t.ndim)
    total, per_block, node_block, attn_block = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    check(L.jodo_dgt_split_size(ctypes.byref(cfg_struct), ctypes.byref(total), ctypes.byref(per_block), ctypes.byref(node_block), ctypes.byref(attn_block)),
          'jodo_dgt_split_size')
    tape = torch.empty(total.value, dtype=torch.uint8)
    check(L.jodo_dgt_pack_split_host(ctypes.byref(cfg_struct), arr, len(keep), ctypes.c_void_p(tape.data_ptr()), ctypes.c_size_t(total.value)),
          'jodo_dgt_pack_split_host')
    return tape if device is None else tape.to(device)
