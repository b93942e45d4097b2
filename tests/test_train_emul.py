"""CPU checks of the training path: the SAME kernel sources that libjodo_hip.so runs on the GPU
(jodo_amd/csrc/dgt_train.hip + train_ops.h), compiled for the host by tests/emul/Makefile against a sequential stand-in for the HIP
runtime (tests/emul/hip/hip_runtime.h; the MFMA GEMM is replaced by plain loops), driven with host pointers and compared with
torch.autograd through the oracle and with the reference's own loss.backward() (tests/golden/grad_qm9.npz).  This checks the
calculus and the index arithmetic of every training kernel where there is no GPU; tests/test_train_gpu.py repeats it on the device."""
import ctypes
import os
import subprocess

import pytest
import torch

from oracle import dgt_oracle as O
from oracle import train_ref as T

from helpers import load_fixture, make_config, make_model, masks, random_inputs

EMUL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul')


@pytest.fixture(scope='module')
def emul():
    subprocess.run(['make', '-C', EMUL_DIR], check=True, capture_output=True)
    return ctypes.CDLL(os.path.join(EMUL_DIR, 'libjodo_train_emul.so'))


def engine_for(emul, model, n_nodes):
    from jodo_amd.train import TrainEngine
    named = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    return TrainEngine(model._cfg(), n_nodes, max(n_nodes), named, 'cpu', lib=emul, stream_ptr=lambda: ctypes.c_void_p(0)), [k for k, _ in named]


def oracle_grads(model, hp, xh, nm, em, ex, cx, cex, nl, ctx, d_out_x, d_out_e, dtype=torch.float64):
    """d <d_out, outputs> / d parameters by autograd through oracle.forward_dense."""
    sd = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in model.state_dict().items()}
    c = lambda t: None if t is None else t.to(dtype)
    px, pe = O.forward_dense(sd, hp, c(xh), c(nm), c(em), c(ex), c(cx), c(cex), c(nl), c(ctx))
    ((px * d_out_x.to(dtype)).sum() + (pe * d_out_e.to(dtype)).sum()).backward()
    return px.detach(), pe.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}


def compare_all(names, grads, want, rel_tol):
    bad = []
    for k, g in zip(names, grads):
        w = want[k].double()
        scale = float(w.abs().max())
        err = float((g.double() - w).abs().max())
        if not err <= rel_tol * max(scale, 1e-12) + 1e-12:
            bad.append("%s: err %.3e scale %.3e" % (k, err, scale))
    assert not bad, "%d of %d parameter gradients differ:\n  %s" % (len(bad), len(names), "\n  ".join(bad[:40]))


def test_training_step_reproduces_the_reference_backward(emul):
    """The reference's recorded training step (grad_qm9: QM9 model, self-conditioned branch, eval-mode dropout): forward ==
    the reference's prediction, loss == its loss, and after loss.backward() through the kernels the 17 recorded parameter
    gradients == the reference's, every other parameter == autograd through the oracle."""
    fx = load_fixture('grad_qm9.npz')
    cfg = make_config(str(fx['cfg_name']))
    model = make_model(cfg, int(fx['seed']))
    hp = O.Hyper.from_config(cfg)
    n_nodes = fx['n_nodes'].tolist()
    nm, em = masks(n_nodes)
    t = lambda k: torch.from_numpy(fx[k]).contiguous()
    eng, names = engine_for(emul, model, n_nodes)
    params = [v.detach().float().contiguous() for v in model.state_dict().values()]
    out_x, out_e = eng.forward(params, t('z_t'), t('edge_z_t'), t('cond_x'), t('cond_edge_x'), t('noise_level'), None, 0.0, 0)
    assert eng.flags.tolist()[0] == 0 and eng.flags.tolist()[3] == 1
    assert (out_x - t('pred')).abs().max() < 2e-5 and (out_e - t('edge_pred')).abs().max() < 2e-5
    px, pe = out_x.clone().requires_grad_(True), out_e.clone().requires_grad_(True)
    lw = [float(w) for w in cfg.model.loss_weights.split(',')]
    loss = T.sde_graph_loss(px, pe, t('xh'), t('edge_x'), t('align_pos'), nm, em, t('alpha_t'), t('sigma_t'), lw, cfg.training.reduce_mean)
    assert abs(loss.item() - float(fx['loss'])) < 1e-5 * float(fx['loss'])
    loss.backward()
    grads = eng.backward(params, t('noise_level'), px.grad.contiguous(), pe.grad.contiguous(), 0.0, 0)
    by_name = dict(zip(names, grads))
    for i, k in enumerate(fx['grad_names'].tolist()):
        want = t('grad_%d' % i)
        rel = (by_name[k] - want).abs().max().item() / (want.abs().max().item() + 1e-12)
        assert rel < 2e-4, "%s: %g" % (k, rel)
    _, _, want = oracle_grads(model, hp, t('z_t'), nm, em, t('edge_z_t'), t('cond_x'), t('cond_edge_x'), t('noise_level'), None, px.grad, pe.grad)
    compare_all(names, grads, want, 2e-4)


@pytest.mark.parametrize("cfg_name,n_nodes,over,selfcond", [
    ('vpsde_qm9_uncond_jodo', [4, 1, 2, 6], dict(nf=128, n_layers=2), False),       # first-step branch, n = 1 and 2
    ('vpsde_qm9_cond_jodo', [3, 5], dict(nf=128, n_layers=2), True),                # conditional model: cond_mlp / cond_lin
    ('vpsde_geom_uncond_jodo', [7, 3], dict(nf=128, n_layers=2), True),             # edge_ch 3, mlp_ratio 4, nd 17
])
def test_all_parameter_gradients_match_autograd_through_the_oracle(emul, cfg_name, n_nodes, over, selfcond):
    cfg = make_config(cfg_name, **over)
    model = make_model(cfg, 3, gain=1.5, coord_scale=0.05)
    hp = O.Hyper.from_config(cfg)
    xh, ex, nl, ctx, nm, em = random_inputs(hp, n_nodes, seed=5)
    g = torch.Generator().manual_seed(9)
    cx = cex = None
    if selfcond:
        cx = torch.randn(xh.shape, generator=g) * nm
        cex = torch.randn(ex.shape, generator=g)
        cex = (cex + cex.transpose(1, 2)) * em.reshape(ex.shape[0], ex.shape[1], ex.shape[1], 1)
    d_x = torch.randn(xh.shape, generator=g)
    d_e = torch.randn(ex.shape, generator=g)                   # not symmetric, not masked: the kernels must mask and symmetrise
    eng, names = engine_for(emul, model, n_nodes)
    params = [v.detach().float().contiguous() for v in model.state_dict().values()]
    out_x, out_e = eng.forward(params, xh, ex, cx, cex, nl, ctx, 0.0, 0)
    px, pe, want = oracle_grads(model, hp, xh, nm, em, ex, cx, cex, nl, ctx, d_x, d_e)
    assert (out_x.double() - px).abs().max() < 2e-5 and (out_e.double() - pe).abs().max() < 2e-5
    assert eng.flags.tolist()[3] == (1 if selfcond else 0)
    grads = eng.backward(params, nl, d_x, d_e, 0.0, 0)
    compare_all(names, grads, want, 3e-4)


def test_dropout_masks_are_shared_by_forward_and_backward(emul):
    """With dropout on, the backward must differentiate the function the forward evaluated: directional finite differences of
    <d_out, forward(theta + eps v)> at a fixed seed against <grads, v>; another seed gives another function; p = 0 is the eval path."""
    cfg = make_config('vpsde_qm9_uncond_jodo', nf=128, n_layers=2)
    model = make_model(cfg, 4, gain=1.5, coord_scale=0.05)
    hp = O.Hyper.from_config(cfg)
    n_nodes = [5, 3]
    xh, ex, nl, ctx, nm, em = random_inputs(hp, n_nodes, seed=8)
    g = torch.Generator().manual_seed(2)
    d_x, d_e = torch.randn(xh.shape, generator=g), torch.randn(ex.shape, generator=g)
    eng, names = engine_for(emul, model, n_nodes)
    params = [v.detach().float().contiguous() for v in model.state_dict().values()]
    p, seed = 0.1, 1234
    o0 = eng.forward(params, xh, ex, None, None, nl, None, 0.0, seed)
    alpha0, hhat0 = eng.debug_fetch(1, 0), eng.debug_fetch(0, 0)
    hhat0_last = eng.debug_fetch(0, hp.n_layers - 1)
    o1 = eng.forward(params, xh, ex, None, None, nl, None, p, seed)
    # Where dropout acts.  The reference drops in the four FFN positions of a block (mol_gnn.py:262-268) and NOT on the attention
    # weights: F.dropout(alpha, p=self.dropout) at layers.py:179 runs with TransMixLayer's default dropout = 0 because the block
    # never passes its own (mol_gnn.py:230-231).  Block 0's attention precedes every live site, so its softmax weights and its
    # output must not move with p; the last block's input has been through the FFN sites and must.
    assert torch.equal(eng.debug_fetch(1, 0), alpha0) and torch.equal(eng.debug_fetch(0, 0), hhat0)
    assert float(alpha0.abs().max()) > 0 and not torch.equal(eng.debug_fetch(0, hp.n_layers - 1), hhat0_last)
    o1b = eng.forward(params, xh, ex, None, None, nl, None, p, seed)
    o2 = eng.forward(params, xh, ex, None, None, nl, None, p, seed + 1)
    assert torch.equal(o1[0], o1b[0]) and torch.equal(o1[1], o1b[1])
    assert not torch.equal(o1[0], o0[0]) and not torch.equal(o1[0], o2[0])
    eng.forward(params, xh, ex, None, None, nl, None, p, seed)
    grads = eng.backward(params, nl, d_x, d_e, p, seed)
    f = lambda ps: float(sum((a.double() * b.double()).sum() for a, b in zip(eng.forward(ps, xh, ex, None, None, nl, None, p, seed), (d_x, d_e))))
    vs = [torch.randn(q.shape, generator=g) * q.abs().mean().clamp(min=1e-3) for q in params]
    eps = 1e-3
    plus = f([(q.double() + eps * v.double()).float() for q, v in zip(params, vs)])
    minus = f([(q.double() - eps * v.double()).float() for q, v in zip(params, vs)])
    fd = (plus - minus) / (2 * eps)
    an = float(sum((gq.double() * v.double()).sum() for gq, v in zip(grads, vs)))
    assert abs(fd - an) <= 2e-2 * max(abs(an), abs(fd), 1e-6), (fd, an)
