"""GPU parity of the training-side slice: jodo_edge_ffn_backward — the backward of phase D of a DGT block
(models/mol_gnn.py:313-317) — against torch.autograd through the CPU oracle.  The oracle's gradients are pinned by the
reference's own loss.backward() (tests/golden/grad_qm9.npz, tests/test_oracle_golden.py)."""
import pytest
import torch

from oracle import dgt_oracle as O
from oracle import train_ref as T

from helpers import load_fixture, make_config, make_model, masks

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(got, want, atol, rtol=1e-4):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = (got - want).abs()
    assert bool((err <= atol + rtol * want.abs()).all()), "max err %.3e (max |want| %.3e)" % (err.max().item(), want.abs().max().item())


def _autograd_reference(e_in, ehat, mods, row_mod, W3, b3, W4, b4, d_out, dtype=torch.float32):
    """Local vector-Jacobian product of phase D by torch.autograd (CPU) through oracle/train_ref.edge_ffn_phase."""
    c = lambda t: t.detach().to(dtype).clone().requires_grad_(True)
    e_in, ehat, mods, W3, b3, W4, b4 = c(e_in), c(ehat), c(mods), c(W3), c(b3), c(W4), c(b4)
    De = e_in.shape[1]
    m = mods[row_mod.long()]
    out = T.edge_ffn_phase(e_in, ehat, m[:, 2 * De:3 * De], m[:, 3 * De:4 * De], m[:, 4 * De:5 * De], m[:, 5 * De:6 * De], W3, b3, W4, b4)
    out.backward(d_out.to(dtype))
    return dict(d_e_in=e_in.grad, d_ehat=ehat.grad, d_mods=mods.grad, dW3=W3.grad, db3=b3.grad, dW4=W4.grad, db4=b4.grad)


def test_edge_ffn_backward_on_the_reference_training_step():
    """Block 5 of the QM9 model on the reference's recorded training batch: inputs of phase D and the upstream gradient
    d loss / d e_out come from autograd through the oracle on the reference's loss (the very computation the fixture pins);
    the kernel's d e_in, d ehat, dW3, db3, dW4, db4 and the per-molecule modulation gradients must equal the oracle's local
    backward through those lines."""
    from jodo_amd.train import EdgeFFNBackward
    fx = load_fixture('grad_qm9.npz')
    cfg = make_config(str(fx['cfg_name']))
    model = make_model(cfg, int(fx['seed']))
    hp = O.Hyper.from_config(cfg)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    n_nodes = fx['n_nodes'].tolist()
    nm, em = masks(n_nodes)
    t = lambda k: torch.from_numpy(fx[k])
    px, pe, inter = O.forward_dense(sd, hp, t('z_t'), nm, em, t('edge_z_t'), t('cond_x'), t('cond_edge_x'), t('noise_level'), None,
                                    return_intermediates='graph')
    lw = [float(w) for w in cfg.model.loss_weights.split(',')]
    loss = T.sde_graph_loss(px, pe, t('xh'), t('edge_x'), t('align_pos'), nm, em, t('alpha_t'), t('sigma_t'), lw, cfg.training.reduce_mean)
    l = 5
    blks = [inter[b][l] for b in range(len(n_nodes))]
    d_outs = torch.autograd.grad(loss, [b['e_out'] for b in blks])
    De = hp.de
    e_in = torch.cat([b['e_in'].detach().reshape(-1, De) for b in blks])
    ehat = torch.cat([b['ehat'].detach().reshape(-1, De) for b in blks])
    d_out = torch.cat([g.reshape(-1, De) for g in d_outs])
    assert float(d_out.abs().max()) > 0
    d_out = d_out / d_out.abs().max()                         # the loss gradient this deep is tiny: bring it to O(1) (the map is linear in it)
    rows = [n * n for n in n_nodes]
    row_mod = torch.repeat_interleave(torch.arange(len(n_nodes)), torch.tensor(rows)).to(torch.int32)
    mod_off = torch.tensor([0] + torch.tensor(rows).cumsum(0).tolist(), dtype=torch.int32)
    p = 'e_block_%d.' % l
    mods = torch.stack([torch.cat([blks[b][k].detach()[0] for k in ('eg1', 'eg1', 'eg1', 'es2', 'ec2', 'eg2')]) for b in range(len(n_nodes))])
    assert mods.shape == (len(n_nodes), 6 * De)               # chunks 0, 1 (es1, ec1) are not read by phase D: filled with eg1
    W3, b3, W4, b4 = (sd[p + k].detach() for k in ('ff_linear3.weight', 'ff_linear3.bias', 'ff_linear4.weight', 'ff_linear4.bias'))
    with torch.no_grad():                                     # the function under differentiation reproduces the oracle's block output
        m = mods[row_mod.long()]
        again = T.edge_ffn_phase(e_in, ehat, m[:, 2 * De:3 * De], m[:, 3 * De:4 * De], m[:, 4 * De:5 * De], m[:, 5 * De:], W3, b3, W4, b4)
        assert torch.allclose(again, torch.cat([b['e_out'].detach().reshape(-1, De) for b in blks]), atol=1e-6)
    want = _autograd_reference(e_in, ehat, mods, row_mod, W3, b3, W4, b4, d_out)
    d = lambda x: x.contiguous().to(DEV)
    got = EdgeFFNBackward(W3, b3, W4, b4, DEV)(d(e_in), d(ehat), d(mods), d(row_mod), d(mod_off), d(d_out))
    torch.cuda.synchronize()
    for k in want:
        scale = float(want[k].abs().max())
        close(got[k], want[k], atol=2e-5 * max(scale, 1e-3))
    assert float(got['d_mods'][:, :2 * De].abs().max()) == 0.0           # es1 / ec1 are not part of phase D


@pytest.mark.parametrize("rows,U,seed", [(1, 1, 0), (33, 2, 1), (5000, 37, 2), (40000, 211, 3)])
def test_edge_ffn_backward_random_rows(rows, U, seed):
    """Shapes that exercise padding rows (rows not a multiple of 32), modulation rows cut in the middle of a tile, and
    persistent waves that own several tiles (40 000 rows = 1250 tiles on 512 waves); float64 autograd as the yardstick, and
    run-to-run bit-determinism (fixed-order sums, no atomics)."""
    from jodo_amd.train import EdgeFFNBackward
    g = torch.Generator().manual_seed(seed)
    De, H = 64, 128
    r = lambda *s: torch.randn(*s, generator=g)
    e_in, ehat, d_out = r(rows, De), r(rows, De) * 0.7, r(rows, De)
    mods = r(U, 6 * De) * 0.3
    W3, b3, W4, b4 = r(H, De) / 8, r(H) * 0.1, r(De, H) / 11, r(De) * 0.1
    cuts = torch.sort(torch.randperm(rows - 1, generator=g)[:U - 1] + 1).values if U > 1 else torch.zeros(0, dtype=torch.long)
    mod_off = torch.cat([torch.zeros(1, dtype=torch.long), cuts, torch.tensor([rows])]).to(torch.int32)
    row_mod = torch.repeat_interleave(torch.arange(U), (mod_off[1:] - mod_off[:-1]).long()).to(torch.int32)
    want = _autograd_reference(e_in, ehat, mods, row_mod, W3, b3, W4, b4, d_out, dtype=torch.float64)
    d = lambda x: x.contiguous().to(DEV)
    op = EdgeFFNBackward(W3, b3, W4, b4, DEV)
    got = op(d(e_in), d(ehat), d(mods), d(row_mod), d(mod_off), d(d_out))
    again = op(d(e_in), d(ehat), d(mods), d(row_mod), d(mod_off), d(d_out))
    torch.cuda.synchronize()
    for k in want:
        scale = float(want[k].abs().max())
        summed = k.startswith(('dW', 'db', 'd_mods'))
        close(got[k], want[k], atol=3e-6 * max(scale, 1.0) * (rows ** 0.5 if summed else 1.0), rtol=2e-5)
        assert torch.equal(got[k], again[k])


def test_edge_ffn_backward_rejects_other_shapes():
    from jodo_amd import capi
    from jodo_amd.train import EdgeFFNBackward
    W3, b3, W4, b4 = torch.zeros(256, 64), torch.zeros(256), torch.zeros(64, 256), torch.zeros(64)      # mlp_ratio 4: not in this slice
    op = EdgeFFNBackward(W3, b3, W4, b4, DEV)
    z = torch.zeros(32, 64, device=DEV)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    with pytest.raises(capi.JodoHipError, match="unsupported"):
        op(z, z, torch.zeros(1, 384, device=DEV), torch.zeros(32, dtype=torch.int32, device=DEV), i32([0, 32]), z)
    ok = EdgeFFNBackward(torch.zeros(128, 64), torch.zeros(128), torch.zeros(64, 128), torch.zeros(64), DEV)
    with pytest.raises(TypeError):
        ok(z.cpu(), z, torch.zeros(1, 384, device=DEV), torch.zeros(32, dtype=torch.int32, device=DEV), i32([0, 32]), z)
