"""The opt-in split-bf16 (three-term) MFMA form of a projection (csrc/dgt_split.h): packing on the CPU, arithmetic on the GPU.

w = hi + mid + lo in bf16 is exact for every finite float; the six retained products accumulated in fp32 must be as good as the
exact-fp32 MFMA chain the product path runs (gate of the round-5 review: not worse than 2 x its error against float64)."""
import ctypes

import numpy as np
import pytest
import torch

from jodo_amd import capi


def _pack(W):
    n_out, n_in = W.shape
    f = np.zeros(n_out * n_in, dtype=np.float32)
    s = np.zeros(n_out * n_in * 3, dtype=np.uint16)
    Wc = np.ascontiguousarray(W, dtype=np.float32)
    capi.check(capi.lib().jodo_debug_pack_split(Wc.ctypes.data_as(ctypes.c_void_p), n_out, n_in, f.ctypes.data_as(ctypes.c_void_p),
                                                s.ctypes.data_as(ctypes.c_void_p)), 'pack_split')
    return f, s


def test_split_packing_is_exact_and_a_permutation_of_the_f32_packing():
    rng = np.random.default_rng(3)
    W = (rng.standard_normal((64, 96)) * 10.0 ** rng.uniform(-6, 6, size=(64, 96))).astype(np.float32)
    f, s = _pack(W)
    b2f = lambda u: (u.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    t = s.reshape(2, 6, 3, 64, 8)                                # [out block][K16 step][term][lane][j]
    rec = b2f(t[:, :, 0]) + b2f(t[:, :, 1]) + b2f(t[:, :, 2])    # hi + mid + lo, summed in float64: exact
    # lane l of step G, element j reads the f32 packing's k-step 8 G + j of the same lane: [block][quad = kstep / 4][lane][kstep % 4]
    f4 = f.reshape(2, 12, 64, 4)
    want = np.empty_like(rec)
    for G in range(6):
        for j in range(8):
            k = 8 * G + j
            want[:, G, :, j] = f4[:, k // 4, :, k % 4]
    assert np.array_equal(rec, want.astype(np.float64))
    assert capi.lib().jodo_debug_pack_split(W.ctypes.data_as(ctypes.c_void_p), 60, 96, None, None) != 0      # sizes are checked


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 33, 4096])
def test_split_projection_is_as_good_as_the_fp32_chain(rows):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 256, generator=g)
    W = (torch.rand(256, 256, generator=g) * 2 - 1) / 16
    y64 = x.double() @ W.double().t()
    f, s = _pack(W.numpy())
    wf, ws = torch.from_numpy(f).cuda(), torch.from_numpy(s.view(np.int16)).cuda()
    xd = x.cuda()
    err = {}
    for mode in (0, 1, 2):
        y = torch.zeros(rows, 256, device='cuda')
        capi.check(capi.lib().jodo_debug_chain(mode, 256, 1, capi.ptr(xd), rows, capi.ptr(wf), capi.ptr(ws), capi.ptr(y), 1, None,
                                               capi.current_stream_ptr()), 'debug_chain')
        torch.cuda.synchronize()
        err[mode] = float((y.cpu().double() - y64).abs().max())
    assert err[0] < 5e-6                                          # the product form: a K = 256 fp32 fma chain
    assert err[1] <= 2.0 * err[0] and err[2] <= 2.0 * err[0], err
