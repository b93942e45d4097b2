"""MFMA lane-map / packer validation on real hardware: the strip-model MLP chain in
jodo_amd/csrc/dgt_debug.hip must reproduce a plain torch fp32 reference."""
import numpy as np
import pytest
import torch

import py_packing as P


def _ref(x, W1, b1, W2, b2):
    h = torch.nn.functional.silu(x @ W1.t() + b1)
    y = h @ W2.t() + b2
    y = torch.nn.functional.layer_norm(y, (y.shape[-1],), None, None, 1e-6)
    return torch.tanh(y)


def test_packer_emulation_matches_dense():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((20, 64))
    W1 = rng.standard_normal((128, 64))
    W2 = rng.standard_normal((252, 128))
    im, om = P.natural_in_map(64), P.natural_out_map(128)
    acc = P.emulate_projection(P.pack_projection(W1, im, om), P.to_slots(x, im))
    y1 = P.from_slots(acc, om, 128)[:20]
    assert np.abs(y1 - x @ W1.T).max() < 1e-4
    qm = P.qk_out_map(14, 18)
    acc2 = P.emulate_projection(P.pack_projection(W2, P.natural_in_map(128), qm), P.acc_as_act(acc))
    y2 = P.from_slots(acc2, qm, 252)[:20]
    assert np.abs(y2 - (x @ W1.T) @ W2.T).max() < 1e-3
    # every one of the 252 features appears exactly once in the qk arrangement
    flat = qm.reshape(-1)
    assert sorted(flat[flat >= 0].tolist()) == list(range(252))


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 31, 32, 33, 1000])
def test_debug_mlp_matches_torch(rows):
    from jodo_amd import capi
    torch.manual_seed(rows)
    dev = torch.device('cuda:0')
    x = torch.randn(rows, 64)
    W1 = torch.randn(128, 64) / 8
    b1 = torch.randn(128)
    W2 = torch.randn(64, 128) / 11      # asymmetric, non-square: catches transposed maps
    b2 = torch.randn(64)
    w1p = torch.from_numpy(P.pack_projection(W1.numpy(), P.natural_in_map(64), P.natural_out_map(128))).to(dev)
    w2p = torch.from_numpy(P.pack_projection(W2.numpy(), P.natural_in_map(128), P.natural_out_map(64))).to(dev)
    xd, b1d, b2d = x.to(dev), b1.to(dev), b2.to(dev)
    y = torch.empty(rows, 64, device=dev)
    L = capi.lib()
    capi.check(L.jodo_debug_mlp(capi.ptr(xd), rows, capi.ptr(w1p), capi.ptr(b1d), capi.ptr(w2p), capi.ptr(b2d),
                                capi.ptr(y), capi.current_stream_ptr()), 'jodo_debug_mlp')
    torch.cuda.synchronize()
    ref = _ref(x, W1, b1, W2, b2)
    err = (y.cpu() - ref).abs().max().item()
    assert err < 2e-5, err
