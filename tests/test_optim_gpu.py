"""The optimiser side of the training step on flat buffers (csrc/train_optim.hip, jodo_amd/optim.py) against torch's own optimisers,
the reference's host-side gradient clipping (losses.py:29-72) and the per-tensor EMA update (models/ema.py:38-40)."""
import copy

import numpy as np
import pytest
import torch

from jodo_amd import optim as JO
from jodo_amd import losses as L
from jodo_amd.models.ema import ExponentialMovingAverage

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
SHAPES = [(7, 5), (3,), (64, 33), (1,), (129, 17), (2, 3, 5)]          # 35 + 3 + 2112 + 1 + 2193 + 30 = 4374 floats: slices off 16-byte boundaries


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(s, generator=g) * 0.3).to(DEV)) for s in SHAPES]


def _grads(step, seed=5):
    g = torch.Generator().manual_seed(1000 * seed + step)
    return [(torch.randn(s, generator=g) * (10.0 ** float(torch.randint(-3, 2, (1,), generator=g)))).to(DEV) for s in SHAPES]


@pytest.mark.parametrize("kind", ["AdamW", "Adam"])
def test_flat_adam_follows_torch(kind):
    """Eight steps with a moving learning rate (the reference's warm-up rewrites it every step): parameters and moments stay within float32
    rounding of torch.optim.AdamW(amsgrad=True, weight_decay=1e-12) / Adam(weight_decay=0.01) — the optimisers get_optimizer builds
    (losses.py:14-26)."""
    pa, pb = _params(1), _params(1)
    if kind == "AdamW":
        ref = torch.optim.AdamW(pa, lr=2e-4, amsgrad=True, weight_decay=1e-12)
        opt = JO.FlatAdam(pb, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-12, amsgrad=True, decoupled=True)
    else:
        ref = torch.optim.Adam(pa, lr=2e-4, betas=(0.8, 0.999), eps=1e-6, weight_decay=0.01)
        opt = JO.FlatAdam(pb, lr=2e-4, betas=(0.8, 0.999), eps=1e-6, weight_decay=0.01, amsgrad=False, decoupled=False)
    flat = JO.flat_view([p.data for p in pb])
    assert flat is not None and flat.numel() == JO.slice_offsets([int(np.prod(s)) for s in SHAPES])[1]     # the parameters now tile one buffer
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)                                                           # ... with their values
    versions = [p._version for p in pb]
    for step in range(1, 9):
        lr = 2e-4 * min(step / 5.0, 1.0) * (10.0 if step == 7 else 1.0)
        for o in (ref, opt):
            o.param_groups[0]['lr'] = lr
        gs = _grads(step)
        for a, b, g in zip(pa, pb, gs):
            a.grad, b.grad = g.clone(), g.clone()                                          # separate gradient tensors: FlatAdam gathers them
        ref.step(); opt.step()
        for a, b in zip(pa, pb):
            torch.testing.assert_close(b.data, a.data, rtol=2e-6, atol=1e-9)
    assert all(p._version > v for p, v in zip(pb, versions))                               # the raw-pointer update is visible to version checks
    for a, b in zip(pa, pb):
        sa, sb = ref.state[a], opt.state[b]
        torch.testing.assert_close(sb['exp_avg'], sa['exp_avg'], rtol=2e-6, atol=1e-12)
        torch.testing.assert_close(sb['exp_avg_sq'], sa['exp_avg_sq'], rtol=2e-6, atol=1e-20)
        if kind == "AdamW":
            torch.testing.assert_close(sb['max_exp_avg_sq'], sa['max_exp_avg_sq'], rtol=2e-6, atol=1e-20)
        assert float(sb['step']) == float(sa['step']) == 8.0


def test_flat_adam_takes_the_flat_gradient_and_resumes_from_a_state_dict():
    """Gradients that are slices of one buffer (what jodo_train_backward hands to autograd) are used in place; a state_dict taken after
    three steps and loaded into a new optimiser over copies of the parameters continues bit for bit."""
    pa = _params(2)
    opt = JO.FlatAdam(pa, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-12, amsgrad=True, decoupled=True)
    offs, total = JO.slice_offsets([p.numel() for p in pa])

    def set_flat_grads(ps, step):
        flat = torch.zeros(total, device=DEV)
        for p, piece, g in zip(ps, JO.carve(flat, [tuple(p.shape) for p in ps], offs), _grads(step, seed=9)):
            piece.copy_(g)
            p.grad = piece
        return flat

    for step in range(1, 4):
        flat = set_flat_grads(pa, step)
        got = JO.flat_gradient(pa, total)
        assert got is not None and got.data_ptr() == flat.data_ptr()
        opt.step()
    import io
    buf = io.BytesIO()
    torch.save(opt.state_dict(), buf)                                                      # the way utils.save_checkpoint stores it (utils.py:23-30)
    assert buf.tell() < 4 * 4 * total * 4                                                   # the shared moment buffers are written once, not once per view
    buf.seek(0)
    sd = torch.load(buf, map_location=DEV)
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    opt2 = JO.FlatAdam(pb, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-12, amsgrad=True, decoupled=True)
    opt2.load_state_dict(sd)
    assert JO.flat_view([opt2.state[p]['exp_avg'] for p in pb]) is not None                # still views of the flat moment buffer
    for step in range(4, 7):
        set_flat_grads(pa, step); set_flat_grads(pb, step)
        opt.step(); opt2.step()
    for a, b in zip(pa, pb):
        assert torch.equal(a.data, b.data)
    with pytest.raises(Exception):
        JO.FlatAdam([torch.nn.Parameter(torch.zeros(3))], lr=1e-3)                          # CPU parameters: refused, no fallback
    # a MIDDLE gradient replaced by a separate tensor (a hook, a manual assignment) is noticed every step: the flat view is refused and
    # the step gathers the real gradients instead of reading stale bytes of the old flat buffer (round-5 advisor finding)
    flat = set_flat_grads(pa, 7); set_flat_grads(pb, 7)
    mid = len(pa) // 2
    fresh = torch.full_like(pa[mid], 0.125)
    pa[mid].grad = fresh.clone(); pb[mid].grad = fresh.clone()
    assert JO.flat_gradient(pa, total) is None
    flat[offs[mid]:offs[mid] + pa[mid].numel()] = 1e6            # what a stale read would pick up
    before = pa[mid].detach().clone()
    opt.step()
    ref = [torch.nn.Parameter(p.detach().clone()) for p in pb]
    for r, p in zip(ref, pb):
        r.grad = p.grad.detach().clone()
    opt2.step()
    assert torch.equal(pa[mid].data, pb[mid].data) and float((pa[mid] - before).abs().max()) < 1e-2
    # load_state_dict validates what it is given
    sd2 = opt.state_dict()
    bad = dict(state=sd2['state'], param_groups=[dict(sd2['param_groups'][0], params=list(range(len(pa) - 1)))])
    with pytest.raises(ValueError, match='parameters'):
        opt2.load_state_dict(bad)
    part = dict(state={0: sd2['state'][0]}, param_groups=sd2['param_groups'])
    with pytest.raises(ValueError, match='all or none'):
        opt2.load_state_dict(part)
    with pytest.raises(ValueError, match='maximize'):
        opt2.load_state_dict(dict(state=sd2['state'], param_groups=[dict(sd2['param_groups'][0], maximize=True)]))
    opt2.load_state_dict(dict(state=sd2['state'], param_groups=[dict(sd2['param_groups'][0], foreach=None, fused=None, capturable=False)]))
    assert 'capturable' not in opt2.param_groups[0] and 'fused' not in opt2.param_groups[0]


def test_device_gradnorm_queue_follows_the_host_clipping():
    """gradient_clipping (losses.py:29-50) over 80 steps of gradient norms with spikes: the device-side history (jodo_gradnorm_clip) makes
    the same decisions as the host Queue — same clipped gradients, same history — without reading the norm back."""
    g = torch.Generator().manual_seed(11)
    host_q = L.Queue(); host_q.add(3000)
    dev_q = JO.DeviceGradNormQueue(DEV, first=3000.0)
    n = 5000
    clipped = 0
    for step in range(80):
        scale = float(torch.rand((), generator=g)) * 3.0 + 0.2
        if step in (20, 41, 42, 70):
            scale *= 40.0                                                                   # spikes: clipped against the history
        if step < 3:
            scale *= 500.0                                                                  # early large norms: clipped by max_grad
        base = (torch.randn(n, generator=g) * scale).to(DEV)
        p = torch.nn.Parameter(torch.zeros(n, device=DEV))
        p.grad = base.clone()
        norm_before = float(torch.linalg.vector_norm(base))
        L.gradient_clipping([p], host_q, 2000.0, True)
        flat = base.clone()
        total = dev_q.clip_(flat, 2000.0)
        assert abs(float(total) - norm_before) <= 1e-6 * norm_before
        torch.testing.assert_close(flat, p.grad, rtol=2e-6, atol=0)
        clipped += int(float(dev_q.coef) < 1.0)
        np.testing.assert_allclose(dev_q.items(), host_q.items, rtol=1e-6)
    assert 4 <= clipped <= 60 and len(host_q) == 50                                        # some steps clipped, most not


def test_flat_ema_update_is_the_per_tensor_update_bit_for_bit():
    pa = _params(3)
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    ema_a = ExponentialMovingAverage(pa, 0.999)
    ema_b = ExponentialMovingAverage(pb, 0.999)
    JO.flatten_parameters(pb)                                                               # b: parameters on one buffer -> flat EMA path
    for step in range(1, 6):
        for a, b, gr in zip(pa, pb, _grads(step, seed=3)):
            with torch.no_grad():
                a.add_(gr * 0.01); b.add_(gr * 0.01)
        ema_a.update(pa); ema_b.update(pb)
    assert getattr(ema_b, '_flat_shadow', None) is not None and getattr(ema_a, '_flat_shadow', None) is None
    for sa, sb in zip(ema_a.shadow_params, ema_b.shadow_params):
        assert torch.equal(sa, sb)
    # a loaded state replaces the shadow list: the flat copy follows
    ema_b.load_state_dict(copy.deepcopy(ema_a.state_dict()))
    ema_a.update(pa); ema_b.update(pb)
    for sa, sb in zip(ema_a.shadow_params, ema_b.shadow_params):
        assert torch.equal(sa, sb)


def test_kabsch_rotations_match_the_svd_form():
    """kabsch_batch (losses.py:424-434) on the device runs jodo_kabsch_rotations instead of torch.linalg.svd (which synchronises): the
    same rotation U diag(1, 1, sign det A) V^T — proper and improper covariances, near-aligned point sets, and what it is used for
    (get_align_position)."""
    g = torch.Generator().manual_seed(5)
    B, N = 300, 19
    P = torch.randn(B, N, 3, generator=g)
    Q = torch.randn(B, N, 3, generator=g)
    rot = L.kabsch_batch(torch.randn(B, 4, 3, generator=g), torch.randn(B, 4, 3, generator=g))           # CPU: the reference's SVD form
    Q[:100] = torch.einsum('bij,bnj->bni', rot[:100], P[:100]) + 0.01 * torch.randn(100, N, 3, generator=g)   # nearly a rotated copy
    Q[100:150] = -Q[100:150]
    want = L.kabsch_batch(P.double(), Q.double())                                                         # CPU float64, SVD
    got = L.kabsch_batch(P.to(DEV), Q.to(DEV))
    assert got.is_cuda and got.dtype == torch.float32
    A = torch.einsum('...ki,...kj->...ij', P.double(), Q.double())
    assert int((torch.det(A) < 0).sum()) > 20 and int((torch.det(A) > 0).sum()) > 20
    torch.testing.assert_close(got.cpu().double(), want, rtol=0, atol=5e-6)
    z = torch.cat([P, torch.zeros(B, N, 2)], -1).to(DEV)
    x = torch.cat([Q, torch.zeros(B, N, 2)], -1).to(DEV)
    torch.testing.assert_close(L.get_align_position(z, x).cpu().double(), L.get_align_position(z.cpu().double(), x.cpu().double()), rtol=0, atol=2e-5)


def test_flat_adam_leaves_parameters_without_a_gradient_alone():
    """Frozen layers: torch's optimisers skip parameters whose .grad is None; FlatAdam updates the runs of consecutive parameters that
    have one and leaves the others bit for bit where they were (moments too)."""
    pa, pb = _params(4), _params(4)
    ref = torch.optim.AdamW(pa, lr=1e-3, amsgrad=True, weight_decay=1e-12)
    opt = JO.FlatAdam(pb, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-12, amsgrad=True, decoupled=True)
    frozen = (1, 4)                                                                       # the 3-float and the [129, 17] tensor
    before = [pb[i].detach().clone() for i in frozen]
    for step in range(1, 4):
        for i, (a, b, g) in enumerate(zip(pa, pb, _grads(step, seed=6))):
            a.grad, b.grad = (None, None) if i in frozen else (g.clone(), g.clone())
        ref.step(); opt.step()
    for i, (a, b) in enumerate(zip(pa, pb)):
        torch.testing.assert_close(b.data, a.data, rtol=2e-6, atol=1e-9)
    for i, w in zip(frozen, before):
        assert torch.equal(pb[i].data, w) and float(opt.state[pb[i]]['exp_avg'].abs().max()) == 0.0
