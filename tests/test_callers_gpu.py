"""GPU parity tests of the caller-side kernels against what the REFERENCE recorded, one hop:

  * jodo_sampler_step on every step of the reference's 50-step trajectory: its own inputs, predictions and noise draws in,
    its own next state out (sampling.py:569-589);
  * jodo_decode on the reference's final states against the reference's decoded molecules (sampling.py:53-97);
  * the in-kernel Philox draws (jodo_sampler_step_rng / jodo_dpm_update_rng) against the numpy restatement
    oracle/philox_ref.py (itself pinned by the Random123 known-answer vectors) and their distributional properties;
  * a 50-NFE hybrid DPM-solver round of the conditional model at the per-GPU batch of BASELINE configs[4], every one of
    its score-network evaluations re-evaluated by the CPU oracle on sampled molecules;
  * two sampling rounds with different atom counts through ONE solver object;
  * checkpoint ingestion on the device: reference-format file -> load_for_sampling -> `.data.copy_` EMA overwrite.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import dgt_oracle as O
from oracle import philox_ref as PR

from helpers import close64, load_fixture, make_config, make_model, masks, oracle_32_64, state_dict_cpu

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(got, want, atol=2e-5, rtol=1e-4):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    assert bool((err <= bound).all()), "max err %.3e (bound %.1e + %.0e*|x|)" % (err.max().item(), atol, rtol)


# ---- jodo_sampler_step against the reference's own per-step tensors -------------------------------------------------
def test_fused_sampler_step_reproduces_every_reference_step():
    """traj_qm9_anc50.npz holds, for each of the 50 steps of the reference's AncestralSampler, the state it fed to the
    model (step_x, step_edge_x), the model's prediction and the noise it drew.  One jodo_sampler_step per step on exactly
    those tensors must give the reference's NEXT recorded state (and, at the last step, its returned x_mean): a
    comparison with the reference's arithmetic (sampling.py:569-589, models/utils.py:67-99), not with this repo's torch
    expressions.  fp32 products in the same order: equal to a few ulps."""
    from jodo_amd import fused
    from jodo_amd.diffusion import NoiseScheduleVP
    from jodo_amd.sampling import posterior_coefficients
    fx = load_fixture('traj_qm9_anc50.npz')
    cfg = make_config('vpsde_qm9_uncond_jodo')
    steps = int(fx['steps'])
    nm, em = masks(fx['n_nodes'].tolist(), DEV)
    ns = NoiseScheduleVP(cfg.sde.schedule)
    ts = torch.linspace(ns.T, 1e-3, steps)
    ss = torch.cat([ts[1:], torch.zeros(1)])
    t = lambda k, i: torch.from_numpy(fx[k][i]).to(DEV)
    bufs = fused.StepBuffers(t('step_x', 0), t('step_edge_x', 0))
    n_dev = fused.n_nodes_from_mask(nm)
    worst = 0.0
    for i in range(steps):
        c_x, c_pred, sigma = posterior_coefficients(ns, ts[i], ss[i])[:3]
        eps = fused.split_replayed_noise(t('node_noise', i), t('edge_noise', i))
        xn, en, xm, emn = fused.sampler_step(bufs, n_dev, float(c_x), float(c_pred), float(sigma), t('step_x', i), t('step_edge_x', i),
                                             t('step_pred', i), t('step_edge_pred', i), *eps)
        if i + 1 < steps:
            want_x, want_e = t('step_x', i + 1), t('step_edge_x', i + 1)
            close(xn, want_x, atol=2e-6, rtol=2e-6)
            close(en, want_e, atol=2e-6, rtol=2e-6)
            worst = max(worst, (xn - want_x).abs().max().item(), (en - want_e).abs().max().item())
        else:
            close(xm, torch.from_numpy(fx['x_mean']), atol=2e-6, rtol=2e-6)
            close(emn, torch.from_numpy(fx['edge_x_mean']), atol=2e-6, rtol=2e-6)
        assert torch.equal(en, en.transpose(1, 2)) and float((xn * (1 - nm)).abs().max()) == 0.0
    print("fused step vs reference next state, worst |err| over %d steps: %.2e" % (steps, worst))


# ---- jodo_decode against the reference's decoded molecules --------------------------------------------------------
@pytest.mark.parametrize("fname", ['traj_qm9_anc5.npz', 'traj_qm9_anc50.npz', 'traj_geom_anc3.npz'])
def test_fused_decode_reproduces_reference_decodes(fname):
    """The reference's final (x_mean, edge_x_mean) through jodo_decode against the reference's own post_process output
    stored in the fixture (atom type argmax, charge round, bond thresholds incl. the aromatic branch of the 3-channel GEOM
    data, positions): same fp32 operations in the same order, so discrete results are bit-identical and positions equal."""
    from jodo_amd import fused
    fx = load_fixture(fname)
    cfg = make_config(str(fx['cfg_name']) if 'cfg_name' in fx else 'vpsde_qm9_uncond_jodo')
    n_nodes = fx['n_nodes'].tolist()
    nm, em = masks(n_nodes, DEV)
    pos, at, fc, et = fused.decode(cfg, torch.from_numpy(fx['x_mean']).to(DEV), torch.from_numpy(fx['edge_x_mean']).to(DEV),
                                   fused.n_nodes_from_mask(nm))
    torch.cuda.synchronize()
    real = nm[..., 0].cpu().numpy() > 0
    assert np.array_equal(at.cpu().numpy()[real], fx['atom_type'][real])
    assert np.array_equal(fc.cpu().numpy().astype(np.int64), fx['fc'][..., 0])
    assert np.array_equal(et.cpu().numpy().astype(np.float32), fx['edge_type'])
    assert np.abs(pos.cpu().numpy() - fx['pos']).max() < 1e-6
    if 'geom' in fname:
        assert 4 in np.unique(fx['edge_type'])                   # the aromatic branch is really exercised
    mols = fused.mols_from_decoded(pos, at, fc, et, n_nodes)
    for b, n in enumerate(n_nodes):                              # tuple format of mol_process (sampling.py:12-32)
        assert mols[b][0].shape == (n, 3) and mols[b][2].shape == (n, n)
        assert np.array_equal(mols[b][1].numpy(), fx['atom_type'][b, :n])


# ---- in-kernel Philox draws -----------------------------------------------------------------------------------------
def _rng_step(n_nodes, nd, ch, seed, draw, table_step=None):
    """eps as jodo_sampler_step_rng applies it: x = pred = 0, sigma = 1 -> x_next = eps."""
    from jodo_amd import capi, fused
    B, N = len(n_nodes), max(n_nodes)
    nm, _ = masks(n_nodes, DEV)
    n_dev = fused.n_nodes_from_mask(nm)
    z = lambda *s: torch.zeros(*s, device=DEV)
    x, e = z(B, N, 3 + nd), z(B, N, N, ch)
    xn, en, xm, emn = z(B, N, 3 + nd), z(B, N, N, ch), z(B, N, 3 + nd), z(B, N, N, ch)
    tab = step = None
    if table_step is not None:
        tab = torch.zeros(table_step + 1, 4, device=DEV)
        tab[table_step, 2] = 1.0
        step = torch.tensor([table_step], dtype=torch.int32, device=DEV)
    capi.check(capi.lib().jodo_sampler_step_rng(
        B, N, 3 + nd, ch, capi.ptr(n_dev), ctypes.c_float(0.0), ctypes.c_float(0.0), ctypes.c_float(1.0), capi.ptr(tab), capi.ptr(step),
        ctypes.c_uint64(seed), ctypes.c_uint32(draw), capi.ptr(x), capi.ptr(e), capi.ptr(x), capi.ptr(e), capi.ptr(xn), capi.ptr(en),
        capi.ptr(xm), capi.ptr(emn), capi.current_stream_ptr()), 'jodo_sampler_step_rng')
    torch.cuda.synchronize()
    assert float(xm.abs().max()) == 0.0 and float(emn.abs().max()) == 0.0
    return xn.cpu(), en.cpu()


@pytest.mark.parametrize("n_nodes,nd,ch", [([9, 1, 29, 17, 2], 6, 2), ([44, 7, 61], 17, 3)])
def test_device_noise_matches_philox_restatement(n_nodes, nd, ch):
    seed, draw = 0x1234ABCD5678EF01, 7
    xn, en = _rng_step(n_nodes, nd, ch, seed, draw)
    N = max(n_nodes)
    want_x = torch.from_numpy(PR.node_noise(seed, draw, n_nodes, N, nd))
    want_e = torch.from_numpy(PR.edge_noise(seed, draw, n_nodes, N, ch))
    close(xn, want_x, atol=2e-5, rtol=1e-5)
    close(en, want_e, atol=2e-5, rtol=1e-5)
    assert torch.equal(en, en.transpose(1, 2))                                    # both triangle halves read one counter
    # table form (captured graphs): draw index = draw + *step
    xt, et = _rng_step(n_nodes, nd, ch, seed, draw - 3, table_step=3)
    assert torch.equal(xt, xn) and torch.equal(et, en)


def test_device_noise_properties():
    """N(0,1) moments, exact symmetry / masking / zero diagonal, centre-of-mass-free positions, and distinct streams for
    distinct (seed, draw) — what models/utils.py:67-99 guarantees for the reference's draws."""
    from jodo_amd import fused
    torch.manual_seed(3)
    n_nodes = torch.randint(12, 30, (600,)).tolist()
    B, N = len(n_nodes), max(n_nodes)
    nm, em = masks(n_nodes)
    xn, en = _rng_step(n_nodes, 6, 2, 99, 0)
    assert float((xn * (1 - nm)).abs().max()) == 0.0
    assert float((en * (1 - em.reshape(B, N, N, 1))).abs().max()) == 0.0
    assert torch.equal(en, en.transpose(1, 2))
    assert float(xn[:, :, :3].sum(1).abs().max()) < 2e-5                          # CoM removed per molecule
    feat = xn[:, :, 3:][nm[..., 0] > 0]
    assert abs(float(feat.mean())) < 0.01 and abs(float(feat.std()) - 1) < 0.01
    assert abs(float((feat ** 4).mean()) - 3.0) < 0.1                              # kurtosis of a normal
    tri = torch.tril(torch.ones(N, N), -1).bool()
    lower = en[:, tri][em.reshape(B, N, N)[:, tri] > 0]
    assert abs(float(lower.mean())) < 0.01 and abs(float(lower.std()) - 1) < 0.01
    pos = xn[:, :, :3][nm[..., 0] > 0]
    assert abs(float(pos.std()) - (1 - 1 / 20.0) ** 0.5) < 0.02                    # CoM removal takes 1/n of the variance
    # channels / neighbouring elements uncorrelated (about 12 000 products: sigma of their mean = 0.009, bound = 4.5 sigma)
    assert abs(float((feat[:, 0] * feat[:, 1]).mean())) < 0.04 and abs(float((feat[:-1, 0] * feat[1:, 0]).mean())) < 0.04
    # other draw index, other seed, other rank: different streams
    for seed, draw in ((99, 1), (100, 0), (fused.DeviceNoise.for_rank(99, 1).seed, 0)):
        x2, e2 = _rng_step(n_nodes, 6, 2, seed, draw)
        m = nm[..., 0] > 0
        assert abs(float((x2[:, :, 3:][m] * feat.reshape(-1, 6)).mean())) < 0.02        # 72 000 products: sigma 0.004
        assert not torch.equal(e2, en)
    assert len({fused.DeviceNoise.for_rank(s, r, k).seed for s in (1, 2) for r in range(8) for k in range(3)}) == 48


def test_dpm_update_device_noise_matches_philox_restatement():
    from jodo_amd import capi, fused
    n_nodes = [9, 1, 29, 17, 2]
    B, N, F, ch = len(n_nodes), max(n_nodes), 9, 2
    nm, _ = masks(n_nodes, DEV)
    n_dev = fused.n_nodes_from_mask(nm)
    z = lambda *s: torch.zeros(*s, device=DEV)
    xz, ez, xo, eo = z(B, N, F), z(B, N, N, ch), z(B, N, F), z(B, N, N, ch)
    c8 = (ctypes.c_float * 8)(0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0)
    seed, draw = 0xFEDCBA9876543210, 41
    capi.check(capi.lib().jodo_dpm_update_rng(B, N, F, ch, capi.ptr(n_dev), c8, None, None, 0, 0, ctypes.c_uint64(seed),
                                              ctypes.c_uint32(draw), ctypes.c_uint32(0), capi.ptr(xz), capi.ptr(xz), capi.ptr(ez),
                                              capi.ptr(xz), capi.ptr(ez), capi.ptr(xz), capi.ptr(ez), capi.ptr(xz), capi.ptr(ez),
                                              capi.ptr(xz), capi.ptr(xo), capi.ptr(eo), capi.current_stream_ptr()), 'jodo_dpm_update_rng')
    torch.cuda.synchronize()
    want = torch.from_numpy(PR.node_noise(seed, draw, n_nodes, N, 6))
    close(xo[:, :, :3].cpu(), want[:, :, :3], atol=2e-5, rtol=1e-5)
    assert float(xo[:, :, 3:].abs().max()) == 0.0 and float(eo.abs().max()) == 0.0


def test_device_noise_sampling_round_statistics():
    """A whole ancestral round with in-kernel draws (AncestralSampler(device_noise=...)) against the same round with
    torch.randn draws: different streams, same law — with a model that predicts zero the state after K steps is a known
    Gaussian: x_K = prod(c_x) x_0 + sum of scaled draws; compare the per-channel variance of both runs."""
    from jodo_amd import fused
    from jodo_amd.diffusion import NoiseScheduleVP
    from jodo_amd.models.utils import sample_combined_position_feature_noise, sample_symmetric_edge_feature_noise
    from jodo_amd.sampling import AncestralSampler
    cfg = make_config('vpsde_qm9_uncond_jodo')
    n_nodes = [20] * 512
    nm, em = masks(n_nodes, DEV)
    ns = NoiseScheduleVP(cfg.sde.schedule)

    class Zero(torch.nn.Module):
        def forward(self, t, x, node_mask, edge_mask, edge_x=None, **kw):
            return torch.zeros_like(x), torch.zeros_like(edge_x)

    out = []
    for dn in (None, fused.DeviceNoise.for_rank(5, 0)):
        torch.manual_seed(11)
        z = sample_combined_position_feature_noise(512, 20, 6, nm)
        ez = sample_symmetric_edge_feature_noise(512, 20, 2, em)
        smp = AncestralSampler(ns, torch.linspace(ns.T, 0.5, 8), True, True, True, lambda a, b: (a, b), device_noise=dn)
        st = smp.init_state(z, ez)
        for i in range(7):                                       # (the 8th step goes to s = 0: no noise, state = mean)
            st = smp.step(Zero(), i, st, nm, em)
        out.append((st['x'].clone(), st['edge_x'].clone()))
        if dn is not None:
            assert dn.draw == 7
    (xa, ea), (xb, eb) = out
    assert abs(float(xa[:, :, 3:].std()) / float(xb[:, :, 3:].std()) - 1) < 0.02
    assert abs(float(xa[:, :, :3].std()) / float(xb[:, :, :3].std()) - 1) < 0.03
    assert abs(float(ea.std()) / float(eb.std()) - 1) < 0.02
    assert torch.equal(eb, eb.transpose(1, 2)) and float(xb[:, :, :3].sum(1).abs().max()) < 1e-4


# ---- BASELINE configs[4]: conditional model + hybrid DPM-solver, 50 NFE, per-GPU batch -------------------------------
def test_dpm_50_nfe_round_teacher_forced_against_oracle():
    """One complete 50-NFE round (single-step, order 2: the configuration of BASELINE configs[4]) of the conditional model at
    B = 313 through DPM_Solver_hybrid with the fused updates.  Every one of the 50 score-network evaluations is recorded
    for 24 molecules spread over the size range (inputs incl. the self-conditioning pair and context; the kernels' output)
    and re-evaluated by the dense CPU oracle on exactly those inputs (outputs are batch-independent): teacher forcing
    along the HIP trajectory at the single-forward tolerance."""
    from jodo_amd.diffusion import NoiseScheduleVP
    from jodo_amd.mix_dpm_solver import DPM_Solver_hybrid
    from jodo_amd.models import load_dataset_info, get_node_dist
    from jodo_amd.models.utils import sample_combined_position_feature_noise, sample_symmetric_edge_feature_noise
    cfg = make_config('vpsde_qm9_cond_jodo')
    cfg.sampling.steps, cfg.sampling.method = 50, 'fast'
    cfg.sampling['dpm_solver_method'], cfg.sampling['dpm_solver_order'] = 'singlestep_fixed', 2
    B = 313
    model = make_model(cfg, 12, DEV, head_gain=10.0)
    hp = O.Hyper.from_config(cfg)
    sd = state_dict_cpu(model)
    torch.manual_seed(42)
    n_nodes = get_node_dist(load_dataset_info('qm9_second_half')).sample(B).tolist()
    N = max(n_nodes)
    nm, em = masks(n_nodes, DEV)
    by_size = sorted(range(B), key=lambda b: (n_nodes[b], b))
    sub = sorted(set(by_size[int(round(i * (B - 1) / 23.0))] for i in range(24)), key=lambda b: (n_nodes[b], b))
    sn = [n_nodes[b] for b in sub]
    Ns = max(sn)
    assert Ns == N and min(sn) == min(n_nodes)
    rec = []

    class Rec(torch.nn.Module):
        def forward(self, t, x, node_mask, edge_mask, edge_x=None, noise_level=None, cond_x=None, cond_edge_x=None, context=None):
            out = model(t, x, node_mask, edge_mask, edge_x=edge_x, noise_level=noise_level, cond_x=cond_x, cond_edge_x=cond_edge_x,
                        context=context)
            c = lambda a, k: None if a is None else (a[sub][:, :Ns] if k == 1 else a[sub][:, :Ns, :Ns]).cpu()
            rec.append((c(x, 1), c(edge_x, 2), c(cond_x, 1), c(cond_edge_x, 2), noise_level[sub].cpu(), context[sub].cpu(),
                        c(out[0], 1), c(out[1], 2)))
            return out

    torch.manual_seed(7)
    z = sample_combined_position_feature_noise(B, N, hp.in_node_dim, nm)
    ez = sample_symmetric_edge_feature_noise(B, N, hp.edge_ch, em)
    ctx = torch.randn(B, 1, device=DEV)
    solver = DPM_Solver_hybrid(NoiseScheduleVP(cfg.sde.schedule), cfg)
    x, ex = solver.sampling(Rec(), z, nm, em, ez, ctx)
    torch.cuda.synchronize()
    assert len(rec) == 50 and solver._noise_calls == 49
    assert torch.isfinite(x).all() and torch.isfinite(ex).all() and torch.equal(ex, ex.transpose(1, 2))
    assert float((x * (1 - nm)).abs().max()) == 0.0 and float(x[:, :, :3].sum(1).abs().max()) < 1e-3
    nms, ems = masks(sn)
    worst = 0.0
    with torch.no_grad():
        for k, (xs, es, cx, ce, nl, cs, ox, oe) in enumerate(rec):
            assert (cx is None) == (k == 0)
            r, r64 = oracle_32_64(sd, hp, xs, nms, ems, es, cx, ce, nl, cs)
            wx, _ = close64(ox, r[0], r64[0], 'DPM evaluation %d nodes' % k)
            we, _ = close64(oe, r[1], r64[1], 'DPM evaluation %d edges' % k)
            worst = max(worst, wx, we)
    print("50-NFE teacher-forced worst |err| (24 of 313 molecules, every evaluation): %.2e" % worst)


# ---- one solver object, two rounds with different atom counts (fused.dpm_update's per-round atom counts) -------------
class _FixedNodes:
    """nodes_dist stand-in: returns preset atom counts (consumes no random numbers)."""

    def __init__(self, rounds):
        self.rounds, self.calls = rounds, 0

    def sample(self, n):
        out = torch.tensor(self.rounds[self.calls])
        self.calls += 1
        assert len(out) == n
        return out


class _Ctx:
    def sample_batch(self, n_nodes):
        return torch.randn(len(n_nodes), 1)


@pytest.mark.parametrize("hip_graph", [False, True])
def test_two_dpm_rounds_through_one_sampling_fn(hip_graph):
    """get_sampling_fn builds ONE DPM_Solver_hybrid and reuses it for every round.  Two rounds with the same batch size and
    padded width but different atom counts: the second must equal the same round sampled through a fresh solver (same
    generator state), i.e. nothing of round 1 (atom counts, buffers, self-conditioning state) may leak into it."""
    from jodo_amd.diffusion import NoiseScheduleVP
    from jodo_amd.sampling import get_sampling_fn
    from jodo_amd.utils import get_data_inverse_scaler
    cfg = make_config('vpsde_qm9_cond_jodo')
    cfg.device = torch.device(DEV)
    cfg.sampling.steps, cfg.sampling.method = 6, 'fast'
    cfg.sampling['dpm_solver_method'], cfg.sampling['dpm_solver_order'] = 'singlestep_fixed', 2
    model = make_model(cfg, 12, DEV, head_gain=10.0)
    ns = NoiseScheduleVP(cfg.sde.schedule)
    r1, r2 = [27, 5, 9, 14, 3], [8, 27, 2, 20, 11]
    inv = get_data_inverse_scaler(cfg)

    def run_rounds(lists, state=None):
        if state is not None:
            torch.set_rng_state(state[0]); torch.cuda.set_rng_state(state[1], DEV)
        fn = get_sampling_fn(cfg, ns, _FixedNodes([sum(lists, [])]), 5, 5 * len(lists), inv, prop_dist=_Ctx(), return_raw=True,
                             hip_graph=hip_graph)
        return fn(model)

    torch.manual_seed(31)
    both = run_rounds([r1, r2])
    torch.manual_seed(31)
    first = run_rounds([r1])
    state = (torch.get_rng_state(), torch.cuda.get_rng_state(DEV))
    second = run_rounds([r2], state)
    assert len(both) == 10
    for got, want in zip(both, first + second):
        assert got[0].shape == want[0].shape
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    assert [int(m[0].shape[0]) for m in both] == r1 + r2


def test_cond_sampling_eval_fn_on_the_hip_model():
    """get_cond_sampling_eval_fn (sampling.py:283-392) with the HIP conditional model: the sampled molecules equal those of
    get_sampling_fn from the same generator state (same rounds, same draws), and the score is the stub classifier's scaled MAE
    recomputed here from the returned molecules.  (The reference's own run of this function is reproduced on the CPU with the
    oracle model: tests/test_oracle_golden.py::test_cond_sampling_eval_fn_reproduces_the_reference.)"""
    from jodo_amd.diffusion import NoiseScheduleVP
    from jodo_amd.sampling import get_cond_sampling_eval_fn, get_sampling_fn
    from jodo_amd.utils import get_data_inverse_scaler
    from oracle.stubs import StubClassifier
    cfg = make_config('vpsde_qm9_cond_jodo')
    cfg.device = torch.device(DEV)
    cfg.sampling.steps, cfg.sampling.method = 5, 'ancestral'
    model = make_model(cfg, 12, DEV, head_gain=10.0)
    ns = NoiseScheduleVP(cfg.sde.schedule)
    lists = [[27, 5, 9, 14], [8, 27, 2, 20]]
    inv = get_data_inverse_scaler(cfg)
    mean, mad = 75.3, 6.3

    class Ctx:
        def __init__(self):
            self.seen = []

        def sample_batch(self, n_nodes):
            c = torch.randn(len(n_nodes), 1)
            self.seen.append(c.clone())
            return c

    ctx = Ctx()
    torch.manual_seed(41)
    fn = get_cond_sampling_eval_fn(cfg, ns, _FixedNodes([sum(lists, [])]), 4, 7, inv, prop_dist=ctx,
                                   prop_norm={cfg.cond_property: {'mean': mean, 'mad': mad}})
    mols, score = fn(model, StubClassifier())
    assert len(mols) == 7 and [int(m[0].shape[0]) for m in mols] == sum(lists, [])[:7]
    torch.manual_seed(41)
    plain = get_sampling_fn(cfg, ns, _FixedNodes([sum(lists, [])]), 4, 8, inv, prop_dist=Ctx(), return_raw=True, fused_decode=False,
                            device_noise=False)(model)
    for got, want in zip(mols, plain):
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    # the score from the returned molecules: classifier on (one-hot atoms, positions) per molecule, MAE against the drawn targets
    errs = []
    targets = torch.cat(ctx.seen).squeeze(-1)
    for k, (pos, at, et, fc) in enumerate(mols):
        h0 = torch.nn.functional.one_hot(at, cfg.data.atom_types).float()
        w = torch.arange(1, h0.shape[1] + 1, dtype=torch.float32)
        pred = ((h0 * w).sum(1) * 0.1 + pos.square().sum(1)).mean()
        errs.append(abs(float(pred) * mad + mean - (float(targets[k]) * mad + mean)))
    assert abs(score - sum(errs) / len(errs)) < 1e-3 * max(1.0, score)        # outputNorm['alpha'] = 1


# ---- the RCCL leg at world size 1: the code the 8-GPU run takes, executed on the one GPU there is -------------
_NCCL_WORKER = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, os.environ['JODO_ROOT']); sys.path.insert(0, os.path.join(os.environ['JODO_ROOT'], 'tests'))
from helpers import make_config, make_model
from jodo_amd.diffusion import NoiseScheduleVP
from jodo_amd.dist import gather_sampled
from jodo_amd.models import load_dataset_info, get_node_dist
from jodo_amd.sampling import get_sampling_fn
from jodo_amd.utils import get_data_inverse_scaler
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
out = {}
for method, steps in (('ancestral', 4), ('fast', 6)):
    cfg = make_config('vpsde_qm9_uncond_jodo')
    cfg.device = torch.device('cuda:0')
    cfg.sampling.steps, cfg.sampling.method = steps, method
    cfg.sampling['dpm_solver_method'], cfg.sampling['dpm_solver_order'] = 'singlestep_fixed', 2
    model = make_model(cfg, 5, 'cuda:0', head_gain=30.0)
    ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
    nodes_dist, inv = get_node_dist(load_dataset_info('qm9_with_h')), get_data_inverse_scaler(cfg)
    torch.manual_seed(77)
    want = get_sampling_fn(cfg, ns, nodes_dist, 5, 9, inv, return_raw=True)(model)          # the reference's procedure, unsharded
    res = {}
    for mode in ('parity', 'perf'):
        fn = get_sampling_fn(cfg, ns, nodes_dist, 5, 9, inv, shard=(0, 1), shard_mode=mode, seed=77)
        mine = fn(model)
        assert all(t.is_cuda for r in fn.last_decoded for t in r), "the gather must take the device tensors jodo_decode left"
        full = gather_sampled(fn.last_decoded, fn.last_indices, device='cuda:0')        # all_reduce + all_gather over RCCL
        torch.cuda.synchronize()
        res[mode] = dict(n=len(full), indices_ok=fn.last_indices == list(range(10)),
                         same_as_local=all(all(torch.equal(a, b) for a, b in zip(m, w)) for m, w in zip(full, mine)),
                         sizes_as_unsharded=[int(m[0].shape[0]) for m in full] == [int(m[0].shape[0]) for m in want],
                         sizes=[int(m[0].shape[0]) for m in full], mols=full)
    again = get_sampling_fn(cfg, ns, nodes_dist, 5, 9, inv, shard=(0, 1), shard_mode='parity', seed=77)
    again(model)
    rep = gather_sampled(again.last_decoded, again.last_indices, device='cuda:0')
    res['parity']['replayable'] = all(all(torch.equal(a, b) for a, b in zip(m, w)) for m, w in zip(rep, res['parity']['mols']))
    for mode in res:
        del res[mode]['mols']
    out[method] = res
# the collective itself on every dtype of the wire format
for dt in (torch.uint8, torch.int8, torch.int32, torch.int64, torch.float32):
    t = (torch.arange(12, device='cuda:0') % 5).to(dt).reshape(3, 4)
    buf = [torch.empty_like(t)]
    dist.all_gather(buf, t)
    assert torch.equal(buf[0], t), dt
dist.barrier()
out['backend'] = dist.get_backend()
dist.destroy_process_group()
print('RESULT ' + json.dumps(out))
"""


def test_rccl_leg_at_world_size_one():
    """dist.init_process_group('nccl', world_size=1) on cuda:0 in a child process: sharded rounds (parity and perf mode, ancestral
    and hybrid DPM-solver) through get_sampling_fn(shard=(0, 1)) + dist.gather_sampled on the decoded DEVICE tensors: the gathered
    list must equal the rank's own molecules bit for bit and in global order, carry the atom counts of the unsharded run (shared
    seed), and a parity-mode run must be replayable.  (Molecule-for-molecule equality with the UNSHARDED run is the gloo test's
    job, tests/test_dist_gloo.py: parity mode replays CPU draws, an unsharded GPU run draws on the device.)  These are the exact
    calls the 8-GPU run makes (all_reduce MAX of the padded sizes, one all_gather per tensor of u8 / i8 / i32 /
    i64 / f32), with HSA_ENABLE_IPC_MODE_LEGACY as the launcher exports it."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, JODO_ROOT=root, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    p = subprocess.run([sys.executable, '-c', _NCCL_WORKER], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1]
    out = json.loads(line[7:])
    assert out['backend'] == 'nccl'
    for method in ('ancestral', 'fast'):
        par, perf = out[method]['parity'], out[method]['perf']
        assert par['n'] == 10 and par['indices_ok'] and par['same_as_local'] and par['sizes_as_unsharded'] and par['replayable'], out
        assert perf['n'] == 10 and perf['indices_ok'] and perf['same_as_local'] and perf['sizes_as_unsharded'] and len(set(perf['sizes'])) > 1, out


# ---- checkpoint ingestion on the device --------------------------------------------------------
def test_checkpoint_ema_overwrite_reaches_the_kernels(tmp_path):
    """Reference-format checkpoint (utils.py:23-30: DataParallel keys, positional EMA shadow list) -> load_for_sampling ->
    HIP forward equals the oracle on the EMA weights; then a reference-style in-place overwrite `param.data.copy_(...)`
    (models/ema.py:52-55 — bumps no tensor version) between two rounds: the next round's FIRST evaluation must already use
    the new weights (content fingerprint checked when a new batch's plan is built, before the packed blob is fetched)."""
    from jodo_amd.models import utils as mutils
    from jodo_amd.models import deterministic_init_
    from jodo_amd.models.ema import ExponentialMovingAverage
    from jodo_amd.utils import load_for_sampling, save_checkpoint
    from helpers import random_inputs
    cfg = make_config('vpsde_qm9_uncond_jodo')
    cfg.device = torch.device('cpu')
    raw = mutils.create_model(cfg)
    deterministic_init_(raw.module, seed=1)
    ema = ExponentialMovingAverage(raw.parameters(), decay=0.999)
    shadow_src = make_model(cfg, 2, 'cpu', gain=1.3, coord_scale=0.05)
    with torch.no_grad():
        for s, p in zip(ema.shadow_params, shadow_src.parameters()):
            s.copy_(p)
    path = str(tmp_path / 'checkpoint_9.pth')
    save_checkpoint(path, dict(optimizer=None, model=raw, ema=ema, step=9))
    cfg.device = torch.device(DEV)
    model, ema2, step = load_for_sampling(path, cfg, use_ema=True)
    assert step == 9
    hp = O.Hyper.from_config(cfg)
    n_nodes = [12, 29, 3, 17]
    xh, ex, nl, ctx, nm, em = random_inputs(hp, n_nodes, seed=2)
    d = lambda a: a.to(DEV)

    def hip(nm_, em_):
        with torch.no_grad():
            o = model(d(nl), d(xh), nm_, em_, edge_x=d(ex), cond_x=None, cond_edge_x=None, noise_level=d(nl))
        torch.cuda.synchronize()
        return o[0].cpu(), o[1].cpu()

    def oracle(src):
        return oracle_32_64(state_dict_cpu(src), hp, xh, nm, em, ex, None, None, nl)

    got = hip(d(nm), d(em))
    want, w64 = oracle(shadow_src)
    close64(got[0], want[0], w64[0], 'EMA weights nodes')
    close64(got[1], want[1], w64[1], 'EMA weights edges')
    # reference-style EMA restore: param.data.copy_ (no version bump), then a NEW round (new mask tensors)
    third = make_model(cfg, 3, 'cpu', gain=0.8, coord_scale=0.05)
    versions = [p._version for p in model.parameters()]
    for p, q in zip(model.parameters(), third.parameters()):
        p.data.copy_(q.data.to(DEV))
    assert [p._version for p in model.parameters()] == versions
    got = hip(d(nm.clone()), d(em.clone()))
    want, w64 = oracle(third)
    close64(got[0], want[0], w64[0], 'copied weights nodes')
    close64(got[1], want[1], w64[1], 'copied weights edges')
    assert not torch.allclose(want[0], oracle(shadow_src)[0][0], atol=1e-3)          # the two weight sets really differ
