"""jodo_amd/losses.py (host mirror of /root/reference/losses.py) against the reference's recorded training step
(tests/golden/grad_qm9.npz, written by oracle/make_golden.py from the reference's own get_sde_graph_loss_fn + loss.backward()):
same batch, same torch / python seeds -> the same diffused inputs, aligned targets, loss and gradients, with the CPU oracle
standing in for the score network."""
import random

import numpy as np
import torch

from jodo_amd import losses as L
from jodo_amd.diffusion.noise_schedule import NoiseScheduleVP
from jodo_amd.utils import get_data_scaler
from oracle import dgt_oracle as O

from helpers import grad_fixture_batch, load_fixture, make_config, make_model


class OracleScoreNet:
    """Differentiable CPU stand-in with the model call signature (autograd through oracle.forward_dense)."""

    def __init__(self, model, hp):
        self.sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        self.hp, self.calls, self.mode = hp, [], None

    def train(self):
        self.mode = 'train'

    def eval(self):
        self.mode = 'eval'

    def __call__(self, t, xh, node_mask, edge_mask, context=None, **kw):
        self.calls.append(dict(t=t, z_t=xh, grad=torch.is_grad_enabled(), **kw))
        return O.forward_dense(self.sd, self.hp, xh, node_mask, edge_mask, kw['edge_x'], kw.get('cond_x'), kw.get('cond_edge_x'),
                               kw['noise_level'], context)


def test_loss_fn_reproduces_the_reference_training_step():
    fx = load_fixture('grad_qm9.npz')
    cfg = make_config(str(fx['cfg_name']))
    cfg.device = torch.device('cpu')
    seed = int(fx['seed'])
    batch, pyseed = grad_fixture_batch(cfg, fx['n_nodes'].tolist(), seed)
    ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
    net = OracleScoreNet(make_model(cfg, seed), O.Hyper.from_config(cfg))
    loss_fn = L.get_sde_graph_loss_fn(ns, False, get_data_scaler(cfg), cfg)
    random.seed(pyseed)
    torch.manual_seed(seed)
    loss = loss_fn(net, batch)
    assert net.mode == 'eval' and [c['grad'] for c in net.calls] == [False, True]          # self-conditioning double forward
    first, second = net.calls
    assert first['cond_x'] is None and second['cond_x'] is not None and not second['cond_x'].requires_grad
    t = lambda k: torch.from_numpy(fx[k])
    assert torch.equal(second['t'], t('t')) and torch.equal(second['z_t'], t('z_t')) and torch.equal(second['edge_x'], t('edge_z_t'))
    assert torch.equal(second['noise_level'], t('noise_level'))
    assert (second['cond_x'] - t('cond_x')).abs().max() < 2e-5 and (second['cond_edge_x'] - t('cond_edge_x')).abs().max() < 2e-5
    assert abs(loss.item() - float(fx['loss'])) < 2e-5 * float(fx['loss'])
    loss.backward()
    for i, k in enumerate(fx['grad_names'].tolist()):
        want = t('grad_%d' % i)
        rel = (net.sd[k].grad - want).abs().max().item() / (want.abs().max().item() + 1e-12)
        assert rel < 5e-4, "%s: %g" % (k, rel)


def test_kabsch_alignment_and_scaler():
    g = torch.Generator().manual_seed(0)
    P = torch.randn(5, 7, 3, generator=g)
    q, _ = torch.linalg.qr(torch.randn(5, 3, 3, generator=g))
    q = q * torch.sign(torch.det(q)).reshape(5, 1, 1)                # proper rotations
    Q = torch.einsum('bij,bnj->bni', q, P)                           # Q_n = R P_n
    rot = L.kabsch_batch(Q, P)                                       # rotation taking the target P onto the prediction Q
    assert (rot - q).abs().max() < 1e-5
    z = torch.cat([Q, torch.zeros(5, 7, 2)], -1)
    x = torch.cat([P, torch.zeros(5, 7, 2)], -1)
    assert (L.get_align_position(z, x) - Q).abs().max() < 1e-5
    cfg = make_config('vpsde_qm9_uncond_jodo')
    from jodo_amd.utils import get_data_inverse_scaler
    sc, inv = get_data_scaler(cfg), get_data_inverse_scaler(cfg)
    nm = torch.ones(2, 3, 1)
    em = torch.ones(2, 9, 1)
    pos, at, fc, et = torch.randn(2, 3, 3, generator=g), torch.rand(2, 3, 5, generator=g), torch.randn(2, 3, 1, generator=g), torch.rand(2, 3, 3, 2, generator=g)
    back = inv(*sc(pos, at, fc, nm, et, em)[:3], nm, sc(pos, at, fc, nm, et, em)[3], em)
    for a, b in zip(back, (pos, at, fc, et)):
        assert (a - b).abs().max() < 1e-5


def test_optimization_manager_warmup_and_adaptive_clipping():
    cfg = make_config('vpsde_qm9_uncond_jodo')
    p = torch.nn.Parameter(torch.ones(4))
    opt = L.get_optimizer(cfg, [p])
    assert isinstance(opt, torch.optim.AdamW) and opt.defaults['amsgrad']
    fn = L.optimization_manager(cfg)
    p.grad = torch.full((4,), 100.0)                                  # norm 200: above grad_clip 10 -> clipped to 10
    fn(opt, [p], step=cfg.optim.warmup // 2)
    assert abs(opt.param_groups[0]['lr'] - cfg.optim.lr * 0.5) < 1e-12
    assert abs(float(p.grad.norm()) - cfg.optim.grad_clip) < 1e-4
    q = L.Queue(max_len=3)
    for v in (1, 2, 3, 4):
        q.add(v)
    assert q.items == [4, 3, 2] and q.mean() == 3.0 and abs(q.std() - np.std([4, 3, 2])) < 1e-12


def test_ema_update_is_the_reference_recurrence_bit_for_bit():
    """models/ema.py:31-40: decay warmed up as min(decay, (1 + k) / (10 + k)); shadow <- shadow - (1 - d)(shadow - p) evaluated with the
    reference's three roundings per element (the multi-tensor form must not change a bit)."""
    from jodo_amd.models.ema import ExponentialMovingAverage
    g = torch.Generator().manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in [(7, 5), (33,), (2, 3, 4)]]
    frozen = torch.nn.Parameter(torch.randn(4, generator=g), requires_grad=False)
    ema = ExponentialMovingAverage(params + [frozen], decay=0.999)
    want = [p.detach().clone() for p in params]
    for k in range(1, 6):
        with torch.no_grad():
            for p in params:
                p.add_(torch.randn(p.shape, generator=g) * 0.1)
        ema.update(params + [frozen])
        d = min(0.999, (1 + k) / (10 + k))
        for w, p in zip(want, params):
            w.sub_((1.0 - d) * (w - p.detach()))
        assert ema.num_updates == k and len(ema.shadow_params) == 3
        assert all(torch.equal(s, w) for s, w in zip(ema.shadow_params, want))


def test_clipping_on_the_flat_gradient_buffer_is_clip_grad_norm():
    """jodo_train_backward hands out gradients that are slices of one buffer; gradient_clipping then takes the norm of and scales the
    buffer itself.  Same norm and same clipped gradients as torch.nn.utils.clip_grad_norm_ on separate tensors (losses.py:29-50)."""
    from jodo_amd import losses as L
    g = torch.Generator().manual_seed(3)
    shapes = [(5, 7), (11,), (2, 3, 2)]
    from jodo_amd.optim import slice_offsets
    offs, total = slice_offsets([torch.Size(s).numel() for s in shapes])   # every slice on a 16-byte boundary: 0, 36, 48 of 60 floats
    assert offs == [0, 36, 48] and total == 60
    flat = torch.zeros(total)
    a = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    b = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    for pa, pb, off in zip(a, b, offs):
        n = pa.numel()
        flat[off:off + n] = torch.randn(n, generator=g) * 10
        pa.grad = flat[off:off + n].view(pa.shape).detach()          # what autograd leaves in p.grad: an alias without a base
        pb.grad = flat[off:off + n].view(pb.shape).clone()
    fa = L._flat_gradient(a)
    assert fa is not None and fa.data_ptr() == flat.data_ptr() and fa.shape == flat.shape and L._flat_gradient(b) is None
    want = torch.nn.utils.clip_grad_norm_(b, max_norm=2.5, norm_type=2.0)
    got = L._clip_grad_norm(a, 2.5)
    assert abs(float(got) - float(want)) <= 1e-6 * float(want)
    for pa, pb in zip(a, b):
        assert torch.allclose(pa.grad, pb.grad, rtol=1e-6, atol=0)
    # a partial cover of the buffer (another tensor lives in it) is not taken for the flat case
    assert L._flat_gradient([a[0], a[1]]) is None
    q = L.Queue(); q.add(3000)
    assert float(L.gradient_clipping(a, q, 1000.0, True)) > 0
