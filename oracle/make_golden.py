"""Generate tests/golden/*.npz from the REAL reference (build container only).

Imports /root/reference under oracle/standins, initialises the reference's own DGT_concat /
cond_DGT_concat with `jodo_amd.models.init_utils.deterministic_init_` (weights are a pure function of
(seed, parameter name), so they need not be stored) and records inputs and the reference's outputs:

  fwd_qm9.npz / fwd_geom.npz / fwd_cond.npz   one first-step call (cond = None) and one
                                               self-conditioned call, non-uniform noise levels
  traj_qm9_anc5.npz                            5-step ancestral trajectory (reference AncestralSampler)
                                               with every noise tensor recorded, + decoded molecules
  traj_cond_dpm4.npz                           4-NFE hybrid DPM-solver trajectory on the conditional
                                               model (BASELINE config 5 path), recorded position noise

  fwd_geom_big.npz / fwd_geom384_big.npz      GEOM molecules at the top of the size range, n = 181 (the dataset
                                               maximum, datasets/datasets_config.py:58), 140, 100 — node strips
                                               spanning 4-6 work-item parts, circulant pair walks of 90 offsets
  blocks_qm9.npz / blocks_geom.npz             the reference's own per-block tensors (h, edge_attr, pos after every
                                               e_block, models/mol_gnn.py:562-568) captured with forward hooks
  traj_qm9_anc50.npz                           50-step ancestral trajectory; besides the replayable noise it records
                                               every step's input state and the reference's prediction (teacher forcing)
  fwd_geom_base.npz                            the README's GEOM Base model: nf = 128, n_layers = 6 (README.md:150)
  cond_eval.npz                                the reference's get_cond_sampling_eval_fn (sampling.py:283-392): molecules + scaled MAE, stub classifier
  fwd_geom_l8.npz                              GEOM nf = 256 with n_layers = 8 (BASELINE configs[2] as worded), mlp_ratio 4
  traj_geom_anc3.npz                           3-step ancestral trajectory of the GEOM model (3 bond channels: aromatic decode)
  grad_qm9.npz                                 the reference's own training loss + loss.backward() gradients of selected
                                               parameters on a small batch (pins the oracle's autograd)
  traj_qm9_cfg0.npz                            BASELINE configs[0] at its own size: the reference's get_sampling_fn, batch 64, 50 ancestral
                                               steps, CPU; end state + decodes + seed (the noise is regenerated from the seed)
  traj_cond_dpm_multi8.npz / _single3.npz / _single1.npz
                                               hybrid DPM-solver: 2nd-order multistep (8 NFE), single-step order 3
                                               (6 NFE) and order 1 (3 NFE)

While doing so it asserts the oracle restatement (oracle/dgt_oracle.py) against the reference:
faithful == reference bit-for-bit, dense within 1e-5.
Run:  python oracle/make_golden.py [fixture-name-prefix ...]     (no argument = all)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_import import load_reference, reference_config          # noqa: E402
from oracle import dgt_oracle as O                                      # noqa: E402
from jodo_amd.models.init_utils import deterministic_init_              # noqa: E402
from oracle.stubs import StubClassifier, FixedNodes, NormalContext      # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
HEAD_GAIN = 30.0      # scale the heads' last layers so that argmax / threshold decodes are not degenerate


def build_reference_model(ref, cfg_name, seed, head_gain=1.0, nf=None, n_layers=None):
    cfg = reference_config(cfg_name)
    cfg.device = torch.device('cpu')
    if nf is not None:
        cfg.model.nf = nf                                         # README.md:168 `--config.model.nf 384`
    if n_layers is not None:
        cfg.model.n_layers = n_layers                             # BASELINE configs[2] as worded: "nf=256, 8 layers"
    model = ref.models.utils._MODELS[cfg.model.name](cfg).eval()
    deterministic_init_(model, seed=seed)
    if head_gain != 1.0:
        with torch.no_grad():
            for k in ('node_pred_mlp.4.weight', 'edge_type_mlp.4.weight', 'edge_exist_mlp.4.weight'):
                model.state_dict()[k].mul_(head_gain)
    return cfg, model


def masks(n_nodes):
    B, N = len(n_nodes), max(n_nodes)
    nm = torch.zeros(B, N)
    for i, n in enumerate(n_nodes):
        nm[i, :n] = 1
    em = nm.unsqueeze(1) * nm.unsqueeze(2) * (~torch.eye(N, dtype=torch.bool)).unsqueeze(0)
    return nm.unsqueeze(2), em.reshape(-1, 1)


def forward_fixture(ref, cfg_name, n_nodes, seed, fname, nf=None, n_layers=None):
    cfg, model = build_reference_model(ref, cfg_name, seed, nf=nf, n_layers=n_layers)
    hp = O.Hyper.from_config(cfg)
    g = torch.Generator().manual_seed(seed + 100)
    B, N = len(n_nodes), max(n_nodes)
    nm, em = masks(n_nodes)
    xh = torch.randn(B, N, 3 + hp.in_node_dim, generator=g) * nm
    xh[:, :, :3] = xh[:, :, :3] - xh[:, :, :3].sum(1, keepdim=True) / nm.sum(1, keepdim=True) * nm
    ex = torch.randn(B, N, N, hp.edge_ch, generator=g)
    ex = (torch.tril(ex.permute(0, 3, 1, 2), -1) + torch.tril(ex.permute(0, 3, 1, 2), -1).transpose(-1, -2)).permute(0, 2, 3, 1)
    ex = ex * em.reshape(B, N, N, 1)
    nl = torch.randn(B, generator=g) * 2.0                        # per-molecule noise levels (training-like)
    ctx = torch.randn(B, 1, generator=g) if hp.cond_ch else None
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        kw = dict(edge_x=ex, noise_level=nl, context=ctx)
        r1 = model(torch.ones(B), xh, nm, em, cond_x=None, cond_edge_x=None, **kw)
        r2 = model(torch.ones(B), xh, nm, em, cond_x=r1[0], cond_edge_x=r1[1], **kw)
        for cx, cex, want in ((None, None, r1), (r1[0], r1[1], r2)):
            f = O.forward_faithful(sd, hp, xh, nm, em, ex, cx, cex, nl, ctx)
            d = O.forward_dense(sd, hp, xh, nm, em, ex, cx, cex, nl, ctx)
            assert torch.equal(f[0], want[0]) and torch.equal(f[1], want[1]), "faithful oracle != reference"
            err = max((d[0] - want[0]).abs().max().item(), (d[1] - want[1]).abs().max().item())
            assert err < 1e-5, "dense oracle vs reference: %g" % err
    np.savez_compressed(os.path.join(OUT, fname), torch_num_threads=torch.get_num_threads(), cfg_name=cfg_name, seed=seed, n_nodes=np.array(n_nodes),
                        nf=int(cfg.model.nf), n_layers=int(cfg.model.n_layers), xh=xh.numpy(), edge_x=ex.numpy(), noise_level=nl.numpy(),
                        context=ctx.numpy() if ctx is not None else np.zeros(0, np.float32),
                        out1_x=r1[0].numpy(), out1_e=r1[1].numpy(), out2_x=r2[0].numpy(), out2_e=r2[1].numpy())
    print(fname, 'ok; |out| =', r2[0].abs().max().item(), r2[1].abs().max().item())


def ancestral_fixture(ref, fname, steps=5, n_nodes=(9, 5, 17, 12), seed=21, cfg_name='vpsde_qm9_uncond_jodo'):
    cfg, model = build_reference_model(ref, cfg_name, seed, head_gain=HEAD_GAIN)
    cfg.sampling.steps = steps
    S = ref.sampling
    ns = ref.diffusion.noise_schedule.NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0,
                                                      continuous_beta_1=cfg.sde.continuous_beta_1)
    time_steps = torch.linspace(ns.T, 1e-3, steps)
    sampler = S.AncestralSampler(ns, time_steps, cfg.model.pred_data, cfg.pred_edge, cfg.model.self_cond,
                                 ref.utils.get_self_cond_fn(cfg))
    n_nodes = list(n_nodes)
    B, N = len(n_nodes), max(n_nodes)
    nm, em = masks(n_nodes)
    torch.manual_seed(seed)
    node_nf = cfg.data.atom_types + int(cfg.model.include_fc_charge)
    z = S.sample_combined_position_feature_noise(B, N, node_nf, nm)
    ez = S.sample_symmetric_edge_feature_noise(B, N, cfg.model.edge_ch, em)
    rec_node, rec_edge = [], []
    orig_n, orig_e = S.sample_combined_position_feature_noise, S.sample_symmetric_edge_feature_noise

    def rn(*a, **k):
        v = orig_n(*a, **k)
        rec_node.append(v.clone())
        return v

    def re_(*a, **k):
        v = orig_e(*a, **k)
        rec_edge.append(v.clone())
        return v

    S.sample_combined_position_feature_noise, S.sample_symmetric_edge_feature_noise = rn, re_
    try:
        with torch.no_grad():
            x_mean, e_mean = sampler.sampling(model, z, nm, em, ez, None)
    finally:
        S.sample_combined_position_feature_noise, S.sample_symmetric_edge_feature_noise = orig_n, orig_e
    inv = ref.utils.get_data_inverse_scaler(cfg)
    pos, one_hot, fc, et = S.post_process(x_mean.clone(), cfg.data.atom_types, cfg.model.include_fc_charge, nm, inv,
                                          e_mean.clone(), em, cfg.data.compress_edge)
    # decision margins of the discrete decodes (distance of the decisive value to its threshold)
    _, h_cat, h_int, h_edge = inv(x_mean[:, :, :3], x_mean[:, :, 3:-1], x_mean[:, :, -1:], nm, e_mean, em)
    top2 = h_cat.topk(2, dim=2).values
    m_atom = (top2[..., 0] - top2[..., 1])[nm[..., 0] > 0].min().item()
    m_fc = (0.5 - (h_int - h_int.round()).abs())[nm[..., 0] > 0].min().item()
    emk = em.reshape(B, N, N) > 0
    m_exist = (h_edge[..., 0] - 0.5).abs()[emk].min().item()
    o3 = h_edge[..., 1] * 3.
    m_order = torch.stack([(o3 - t).abs() for t in (0.5, 1.5, 2.5)]).min(0).values[emk].min().item() / 3.
    np.savez_compressed(os.path.join(OUT, fname), torch_num_threads=torch.get_num_threads(), cfg_name=cfg_name, seed=seed, steps=steps, head_gain=HEAD_GAIN,
                        n_nodes=np.array(n_nodes),
                        z=z.numpy(), edge_z=ez.numpy(), node_noise=torch.stack(rec_node).numpy(),
                        edge_noise=torch.stack(rec_edge).numpy(), x_mean=x_mean.numpy(), edge_x_mean=e_mean.numpy(),
                        pos=pos.numpy(), atom_type=one_hot.argmax(2).numpy(), fc=fc.numpy(), edge_type=et.numpy(),
                        margins=np.array([m_atom, m_fc, m_exist, m_order]))
    print(fname, 'ok; margins atom/fc/exist/order =', m_atom, m_fc, m_exist, m_order,
          'atom types', np.unique(one_hot.argmax(2).numpy()), 'bond types', np.unique(et.numpy()))


def blocks_fixture(ref, cfg_name, n_nodes, seed, fname):
    """The reference's own tensors after every e_block (mol_gnn.py:562-568): h [Nn,D], edge_attr [E,De] (sparse,
    row-major (b,i,j) order of dense_to_sparse), pos after remove_mean_with_mask — for a self-conditioned call."""
    cfg, model = build_reference_model(ref, cfg_name, seed)
    hp = O.Hyper.from_config(cfg)
    g = torch.Generator().manual_seed(seed + 100)
    B, N = len(n_nodes), max(n_nodes)
    nm, em = masks(n_nodes)
    xh = torch.randn(B, N, 3 + hp.in_node_dim, generator=g) * nm
    xh[:, :, :3] = xh[:, :, :3] - xh[:, :, :3].sum(1, keepdim=True) / nm.sum(1, keepdim=True) * nm
    ex = torch.randn(B, N, N, hp.edge_ch, generator=g)
    ex = (torch.tril(ex.permute(0, 3, 1, 2), -1) + torch.tril(ex.permute(0, 3, 1, 2), -1).transpose(-1, -2)).permute(0, 2, 3, 1)
    ex = ex * em.reshape(B, N, N, 1)
    nl = torch.randn(B, generator=g) * 2.0
    rec = []
    hooks = [model._modules['e_block_%d' % i].register_forward_hook(lambda m, a, out: rec.append([t.detach().clone() for t in out]))
             for i in range(hp.n_layers)]
    with torch.no_grad():
        r1 = model(torch.ones(B), xh, nm, em, edge_x=ex, noise_level=nl, cond_x=None, cond_edge_x=None, context=None)
        del rec[:]
        r2 = model(torch.ones(B), xh, nm, em, edge_x=ex, noise_level=nl, cond_x=r1[0], cond_edge_x=r1[1], context=None)
    for h_ in hooks:
        h_.remove()
    assert len(rec) == hp.n_layers
    hs = torch.stack([r[0] for r in rec]).reshape(hp.n_layers, B, N, -1)
    es = torch.stack([r[1] for r in rec])                                   # [L, E, De]
    ps = torch.stack([ref.models.utils.remove_mean_with_mask(r[2].reshape(B, N, 3), nm) for r in rec])
    # the dense oracle's intermediates against the reference's own
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        _, _, inter = O.forward_dense(sd, hp, xh, nm, em, ex, r1[0], r1[1], nl, None, return_intermediates=True)
    eoff = 0
    for b, n in enumerate(n_nodes):
        for l in range(hp.n_layers):
            blk = inter[b][l]
            assert (blk['h'] - hs[l, b, :n]).abs().max() < 2e-5
            assert (blk['pos'] - ps[l, b, :n]).abs().max() < 2e-5
            offd = ~torch.eye(n, dtype=torch.bool)
            assert (blk['e'][offd] - es[l, eoff:eoff + n * (n - 1)]).abs().max() < 2e-5
        eoff += n * (n - 1)
    np.savez_compressed(os.path.join(OUT, fname), torch_num_threads=torch.get_num_threads(), cfg_name=cfg_name, seed=seed, n_nodes=np.array(n_nodes),
                        xh=xh.numpy(), edge_x=ex.numpy(), noise_level=nl.numpy(), out1_x=r1[0].numpy(), out1_e=r1[1].numpy(),
                        out2_x=r2[0].numpy(), out2_e=r2[1].numpy(), h=hs.numpy(), e=es.numpy(), pos=ps.numpy())
    print(fname, 'ok; blocks', hp.n_layers, 'E', es.shape[1])


def ancestral_tf_fixture(ref, fname, steps=50, n_nodes=(9, 5, 17, 12), seed=23):
    """Long ancestral trajectory (K = 50, the upper end of SURVEY.md §8c's K-step range) with, per step, the
    input state and the reference model's prediction, so a kernel can be teacher-forced along it."""
    cfg, model = build_reference_model(ref, 'vpsde_qm9_uncond_jodo', seed, head_gain=HEAD_GAIN)
    cfg.sampling.steps = steps
    S = ref.sampling
    ns = ref.diffusion.noise_schedule.NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0,
                                                      continuous_beta_1=cfg.sde.continuous_beta_1)
    sampler = S.AncestralSampler(ns, torch.linspace(ns.T, 1e-3, steps), cfg.model.pred_data, cfg.pred_edge,
                                 cfg.model.self_cond, ref.utils.get_self_cond_fn(cfg))
    n_nodes = list(n_nodes)
    B, N = len(n_nodes), max(n_nodes)
    nm, em = masks(n_nodes)
    torch.manual_seed(seed)
    node_nf = cfg.data.atom_types + int(cfg.model.include_fc_charge)
    z = S.sample_combined_position_feature_noise(B, N, node_nf, nm)
    ez = S.sample_symmetric_edge_feature_noise(B, N, cfg.model.edge_ch, em)
    rec_node, rec_edge, rec_in, rec_out = [], [], [], []
    orig_n, orig_e = S.sample_combined_position_feature_noise, S.sample_symmetric_edge_feature_noise

    def rn(*a, **k):
        v = orig_n(*a, **k)
        rec_node.append(v.clone())
        return v

    def re_(*a, **k):
        v = orig_e(*a, **k)
        rec_edge.append(v.clone())
        return v

    def model_rec(t, x, node_mask, edge_mask, **kw):
        out = model(t, x, node_mask, edge_mask, **kw)
        rec_in.append((x.clone(), kw['edge_x'].clone(), kw['noise_level'].clone()))
        rec_out.append((out[0].clone(), out[1].clone()))
        return out

    S.sample_combined_position_feature_noise, S.sample_symmetric_edge_feature_noise = rn, re_
    try:
        with torch.no_grad():
            x_mean, e_mean = sampler.sampling(model_rec, z, nm, em, ez, None)
    finally:
        S.sample_combined_position_feature_noise, S.sample_symmetric_edge_feature_noise = orig_n, orig_e
    inv = ref.utils.get_data_inverse_scaler(cfg)
    pos, one_hot, fc, et = S.post_process(x_mean.clone(), cfg.data.atom_types, cfg.model.include_fc_charge, nm, inv,
                                          e_mean.clone(), em, cfg.data.compress_edge)
    np.savez_compressed(os.path.join(OUT, fname), torch_num_threads=torch.get_num_threads(), seed=seed, steps=steps, head_gain=HEAD_GAIN, n_nodes=np.array(n_nodes),
                        z=z.numpy(), edge_z=ez.numpy(), node_noise=torch.stack(rec_node).numpy(),
                        edge_noise=torch.stack(rec_edge).numpy(), x_mean=x_mean.numpy(), edge_x_mean=e_mean.numpy(),
                        pos=pos.numpy(), atom_type=one_hot.argmax(2).numpy(), fc=fc.numpy(), edge_type=et.numpy(),
                        step_x=torch.stack([r[0] for r in rec_in]).numpy(), step_edge_x=torch.stack([r[1] for r in rec_in]).numpy(),
                        step_noise_level=torch.stack([r[2] for r in rec_in]).numpy(),
                        step_pred=torch.stack([r[0] for r in rec_out]).numpy(),
                        step_edge_pred=torch.stack([r[1] for r in rec_out]).numpy())
    print(fname, 'ok; steps', len(rec_in), 'atom types', np.unique(one_hot.argmax(2).numpy()), 'bond types', np.unique(et.numpy()))


def dpm_fixture(ref, fname, nfe=4, n_nodes=(9, 5, 17, 12), seed=31, method='singlestep_fixed', order=2):
    cfg, model = build_reference_model(ref, 'vpsde_qm9_cond_jodo', seed, head_gain=HEAD_GAIN)
    cfg.sampling.steps = nfe
    cfg.sampling.method = 'fast'
    cfg.sampling.dpm_solver_method = method                   # keys the cond config lacks
    cfg.sampling.dpm_solver_order = order
    M = ref.mix_dpm_solver
    S = ref.sampling
    ns = ref.diffusion.noise_schedule.NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0,
                                                      continuous_beta_1=cfg.sde.continuous_beta_1)
    solver = M.DPM_Solver_hybrid(ns, cfg)
    n_nodes = list(n_nodes)
    B, N = len(n_nodes), max(n_nodes)
    nm, em = masks(n_nodes)
    torch.manual_seed(seed)
    node_nf = cfg.data.atom_types + int(cfg.model.include_fc_charge)
    z = S.sample_combined_position_feature_noise(B, N, node_nf, nm)
    ez = S.sample_symmetric_edge_feature_noise(B, N, cfg.model.edge_ch, em)
    ctx = torch.randn(B, 1)
    rec = []
    orig = M.sample_center_gravity_zero_gaussian_with_mask

    def rp(*a, **k):
        v = orig(*a, **k)
        rec.append(v.clone())
        return v

    M.sample_center_gravity_zero_gaussian_with_mask = rp
    try:
        x, ex = solver.sampling(model, z, nm, em, ez, ctx)
    finally:
        M.sample_center_gravity_zero_gaussian_with_mask = orig
    np.savez_compressed(os.path.join(OUT, fname), torch_num_threads=torch.get_num_threads(), seed=seed, nfe=nfe, head_gain=HEAD_GAIN, n_nodes=np.array(n_nodes),
                        method=method, order=order, z=z.numpy(), edge_z=ez.numpy(), context=ctx.numpy(), pos_noise=torch.stack(rec).numpy(),
                        x=x.numpy(), edge_x=ex.numpy())
    print(fname, 'ok; noise draws', len(rec))


def grad_fixture(ref, fname, n_nodes=(5, 9, 7), seed=41):
    """The reference's OWN training loss and loss.backward() (losses.py:286-385: get_sde_graph_loss_fn, self-conditioned
    branch taken, eval mode so that dropout is the identity) on a small synthetic batch of the QM9 model.  Recorded: what the
    reference fed to the grad-enabled model call, the targets it built (scaled data, Kabsch-aligned positions), its loss, and
    its gradients of a few parameters spread over the network (edge FFN of two blocks, an attention projection, the
    coordinate MLP, the embeddings, a head) — the pins of the oracle's autograd (oracle/train_ref.py)."""
    import random as pyrandom
    cfg, model = build_reference_model(ref, 'vpsde_qm9_uncond_jodo', seed)
    L = ref.losses
    ns = ref.diffusion.noise_schedule.NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0,
                                                      continuous_beta_1=cfg.sde.continuous_beta_1)
    scaler = ref.utils.get_data_scaler(cfg)
    n_nodes = list(n_nodes)
    B, N = len(n_nodes), max(n_nodes)
    nm, em = masks(n_nodes)
    g = torch.Generator().manual_seed(seed)
    at = torch.randint(0, cfg.data.atom_types, (B, N), generator=g)
    bond = torch.randint(0, 4, (B, N, N), generator=g)
    bond = torch.triu(bond, 1); bond = bond + bond.transpose(1, 2)
    e_exist = (bond > 0).float()
    batch = dict(positions=torch.randn(B, N, 3, generator=g) * nm,
                 atom_mask=nm[..., 0], edge_mask=em,
                 atom_one_hot=torch.nn.functional.one_hot(at, cfg.data.atom_types).float() * nm,
                 edge_one_hot=torch.stack([e_exist, bond.float() / 3.], -1) * em.reshape(B, N, N, 1),      # compress_edge: (exists, order / 3)
                 formal_charges=(torch.randint(-1, 2, (B, N, 1), generator=g).float()) * nm)
    loss_fn = L.get_sde_graph_loss_fn(ns, False, scaler, cfg)
    rec = {}
    inner = model.forward

    def recording_forward(t, xh, node_mask, edge_mask, context=None, **kw):
        if torch.is_grad_enabled():
            rec.update(t=t.clone(), z_t=xh.clone(), edge_z_t=kw['edge_x'].clone(), noise_level=kw['noise_level'].clone(),
                       cond_x=None if kw.get('cond_x') is None else kw['cond_x'].clone(),
                       cond_edge_x=None if kw.get('cond_edge_x') is None else kw['cond_edge_x'].clone())
        out = inner(t, xh, node_mask, edge_mask, context, **kw)
        if torch.is_grad_enabled():
            rec.update(pred=out[0].detach().clone(), edge_pred=out[1].detach().clone())
        return out

    model.forward = recording_forward
    orig_align = L.get_align_position

    def rec_align(z_t, xh):
        a = orig_align(z_t, xh)
        rec.update(align_pos=a.clone(), xh=xh.clone())
        return a

    L.get_align_position = rec_align
    for tries in range(64):                                    # a python-random seed whose first draw takes the self-cond branch
        pyrandom.seed(seed + tries)
        if pyrandom.random() < 0.5:
            pyrandom.seed(seed + tries)
            break
    torch.manual_seed(seed)
    model.zero_grad()
    try:
        loss = loss_fn(model, batch)
        loss.backward()
    finally:
        L.get_align_position = orig_align
        del model.forward
    assert rec['cond_x'] is not None
    xh, edge_x, _, _, _ = L.process_edge_batch(batch, cfg.device, cfg.model.include_fc_charge, scaler, None)
    assert torch.equal(xh, rec['xh'])
    alpha_t, sigma_t = ns.marginal_prob(rec['t'])
    names = ['e_block_0.ff_linear3.weight', 'e_block_0.ff_linear3.bias', 'e_block_0.ff_linear4.weight', 'e_block_0.ff_linear4.bias',
             'e_block_5.ff_linear3.weight', 'e_block_5.ff_linear4.weight', 'e_block_5.ff_linear4.bias',
             'e_block_3.attn_mpnn.lin_edge0.weight', 'e_block_2.node2edge_lin.weight', 'e_block_7.equi_update.coord_mlp.2.weight',
             'e_block_4.equi_update.coord_norm.scale', 'e_block_6.edge_emb.weight', 'edge_emb.weight', 'node_emb.weight',
             'edge_exist_mlp.4.weight', 'time_mlp.0.weights', 'e_block_5.dist_layer.means.weight']
    params = dict(model.named_parameters())
    # the oracle's autograd against the reference's, here and now (the committed fixture is checked again by the CPU suite)
    from oracle import train_ref as T
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    hp = O.Hyper.from_config(cfg)
    px, pe = O.forward_dense(sd, hp, rec['z_t'], nm, em, rec['edge_z_t'], rec['cond_x'], rec['cond_edge_x'], rec['noise_level'], None)
    lw = [float(w) for w in cfg.model.loss_weights.split(',')]
    ol = T.sde_graph_loss(px, pe, xh, edge_x, rec['align_pos'], nm, em, alpha_t, sigma_t, lw, cfg.training.reduce_mean)
    ol.backward()
    assert abs(ol.item() - loss.item()) < 1e-5 * abs(loss.item()), (ol.item(), loss.item())
    for k in names:
        a, b = sd[k].grad, params[k].grad
        rel = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
        assert rel < 2e-4, "oracle gradient of %s: rel err %g" % (k, rel)
    np.savez_compressed(os.path.join(OUT, fname), torch_num_threads=torch.get_num_threads(), cfg_name='vpsde_qm9_uncond_jodo', seed=seed, n_nodes=np.array(n_nodes),
                        t=rec['t'].numpy(), z_t=rec['z_t'].numpy(), edge_z_t=rec['edge_z_t'].numpy(), noise_level=rec['noise_level'].numpy(),
                        cond_x=rec['cond_x'].numpy(), cond_edge_x=rec['cond_edge_x'].numpy(), xh=xh.numpy(), edge_x=edge_x.numpy(),
                        align_pos=rec['align_pos'].numpy(), alpha_t=alpha_t.numpy(), sigma_t=sigma_t.numpy(),
                        pred=rec['pred'].numpy(), edge_pred=rec['edge_pred'].numpy(), loss=np.float64(loss.item()),
                        grad_names=np.array(names), **{'grad_%d' % i: params[k].grad.numpy() for i, k in enumerate(names)})
    print(fname, 'ok; loss', loss.item(), 'max |grad|', max(params[k].grad.abs().max().item() for k in names))


def cond_eval_fixture(ref, fname, steps=4, n_nodes=(9, 5, 17, 12, 3, 18), batch=3, n_samples=5, seed=35):
    """The reference's get_cond_sampling_eval_fn (sampling.py:283-392) on its own conditional model: two rounds of three molecules,
    ancestral sampler, a stub property classifier; records what it returned (molecules, scaled MAE)."""
    cfg, model = build_reference_model(ref, 'vpsde_qm9_cond_jodo', seed, head_gain=HEAD_GAIN)
    cfg.sampling.steps = steps
    cfg.sampling.method = 'ancestral'
    S = ref.sampling
    ns = ref.diffusion.noise_schedule.NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0,
                                                      continuous_beta_1=cfg.sde.continuous_beta_1)
    prop_norm = {cfg.cond_property: {'mean': 75.3, 'mad': 6.3}}
    fn = S.get_cond_sampling_eval_fn(cfg, ns, FixedNodes(n_nodes), batch, n_samples, ref.utils.get_data_inverse_scaler(cfg),
                                     prop_dist=NormalContext(), prop_norm=prop_norm)
    torch.manual_seed(seed)
    mols, score = fn(model, StubClassifier())
    assert len(mols) == n_samples
    out = dict(n_mols=len(mols), score=score, n_nodes=np.array(n_nodes), batch=batch, n_samples=n_samples, seed=seed, steps=steps,
               head_gain=HEAD_GAIN, cond_property=str(cfg.cond_property), prop_mean=75.3, prop_mad=6.3)
    for i, (pos, at, et, fc) in enumerate(mols):
        out['pos_%d' % i] = pos.numpy(); out['atom_%d' % i] = at.numpy(); out['edge_%d' % i] = et.numpy(); out['fc_%d' % i] = fc.numpy()
    np.savez_compressed(os.path.join(OUT, fname), torch_num_threads=torch.get_num_threads(), **out)
    print(fname, 'ok; score', score, 'atoms', [int(m[0].shape[0]) for m in mols])


def cfg0_fixture(ref, fname, batch=64, steps=50, seed=42, model_seed=42):
    """BASELINE.json configs[0] at its own size: the reference's own get_sampling_fn (sampling.py:148-232) — vpsde_qm9_uncond_jodo,
    batch 64, 50 ancestral steps, PyTorch CPU — one round, seeded `torch.manual_seed(seed)` right before the call, so every draw
    (atom counts from the reference's DistributionNodes over its own histogram, z_T, edge_z_T, the per-step noise) is a function of the
    seed and is NOT stored: the replay regenerates it from the CPU generator in the same order (jodo_amd.sampling shard_mode='parity').
    Stored: what the sampler returned (sampling.py:591-594), what post_process / mol_process made of it (before the final shuffle),
    the drawn atom counts and two checksums of the noise stream.  Heads scaled like every trajectory fixture (decodes not degenerate)."""
    import random
    cfg, model = build_reference_model(ref, 'vpsde_qm9_uncond_jodo', model_seed, head_gain=HEAD_GAIN)
    cfg.sampling.steps = steps
    assert cfg.sampling.method == 'ancestral'
    S = ref.sampling
    ns = ref.diffusion.noise_schedule.NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0,
                                                      continuous_beta_1=cfg.sde.continuous_beta_1)
    # datasets/datasets_config.py by path: the package's __init__ pulls in the PyG data pipeline (not on this path, not importable here)
    import importlib.util
    from oracle.ref_import import REFERENCE_ROOT
    spec = importlib.util.spec_from_file_location('jodo_ref_datasets_config', os.path.join(REFERENCE_ROOT, 'datasets', 'datasets_config.py'))
    dsc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dsc)
    info = dsc.get_dataset_info(cfg.data.info_name)
    nodes_dist = ref.models.node_distribution.get_node_dist(info)
    inv = ref.utils.get_data_inverse_scaler(cfg)
    fn = S.get_sampling_fn(cfg, ns, nodes_dist, batch, batch, inv)
    rec = {}
    orig_sampling, orig_pp, orig_mp = S.AncestralSampler.sampling, S.post_process, S.mol_process
    orig_n, orig_e = S.sample_combined_position_feature_noise, S.sample_symmetric_edge_feature_noise
    sums = {'node': [], 'edge': []}

    def rec_sampling(self, model_, z, node_mask, edge_mask, edge_z, context):
        rec['z'], rec['nm'], rec['em'] = z.clone(), node_mask.clone(), edge_mask.clone()
        out = orig_sampling(self, model_, z, node_mask, edge_mask, edge_z, context)
        rec['x_mean'], rec['edge_x_mean'] = out[0].clone(), out[1].clone()
        return out

    def rec_pp(*a, **k):
        out = orig_pp(*a, **k)
        rec['pos'], rec['one_hot'], rec['fc'], rec['et'] = [t.clone() for t in out]
        return out

    def rec_mp(one_hot, pos, fc, n_nodes, edge_types):
        rec['n_nodes'] = torch.as_tensor(n_nodes).clone()
        return orig_mp(one_hot, pos, fc, n_nodes, edge_types)

    def rn(*a, **k):
        v = orig_n(*a, **k)
        sums['node'].append(float(v.double().sum()))
        return v

    def re_(*a, **k):
        v = orig_e(*a, **k)
        sums['edge'].append(float(v.double().abs().sum()))
        return v

    S.AncestralSampler.sampling, S.post_process, S.mol_process = rec_sampling, rec_pp, rec_mp
    S.sample_combined_position_feature_noise, S.sample_symmetric_edge_feature_noise = rn, re_
    try:
        torch.manual_seed(seed)
        random.seed(seed)
        mols = fn(model)
    finally:
        S.AncestralSampler.sampling, S.post_process, S.mol_process = orig_sampling, orig_pp, orig_mp
        S.sample_combined_position_feature_noise, S.sample_symmetric_edge_feature_noise = orig_n, orig_e
    assert len(mols) == batch and len(sums['node']) == steps + 1 and len(sums['edge']) == steps + 1     # z_T + one draw per step
    x_mean, e_mean, nm, em = rec['x_mean'], rec['edge_x_mean'], rec['nm'], rec['em']
    B, N = x_mean.shape[0], x_mean.shape[1]
    _, h_cat, h_int, h_edge = inv(x_mean[:, :, :3], x_mean[:, :, 3:-1], x_mean[:, :, -1:], nm, e_mean, em)
    top2 = h_cat.topk(2, dim=2).values
    m_atom = (top2[..., 0] - top2[..., 1])[nm[..., 0] > 0].min().item()
    emk = em.reshape(B, N, N) > 0
    m_exist = (h_edge[..., 0] - 0.5).abs()[emk].min().item()
    np.savez_compressed(os.path.join(OUT, fname), torch_num_threads=torch.get_num_threads(), cfg_name='vpsde_qm9_uncond_jodo', seed=seed,
                        model_seed=model_seed, steps=steps, batch=batch, head_gain=HEAD_GAIN, n_nodes=rec['n_nodes'].numpy(),
                        z_sum=float(rec['z'].double().sum()), node_noise_sums=np.array(sums['node']), edge_noise_sums=np.array(sums['edge']),
                        x_mean=x_mean.numpy(), edge_x_mean=e_mean.numpy(), pos=rec['pos'].numpy(), atom_type=rec['one_hot'].argmax(2).numpy(),
                        fc=rec['fc'].numpy(), edge_type=rec['et'].numpy(), margins=np.array([m_atom, m_exist]))
    print(fname, 'ok; n_nodes', rec['n_nodes'].tolist()[:8], '... margins atom/exist', m_atom, m_exist,
          'atom types', np.unique(rec['one_hot'].argmax(2).numpy()), 'bond types', np.unique(rec['et'].numpy()))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    jobs = [
        ('fwd_qm9.npz', lambda f: forward_fixture(ref, 'vpsde_qm9_uncond_jodo', [3, 5, 9, 12, 17, 29], 11, f)),
        ('fwd_geom.npz', lambda f: forward_fixture(ref, 'vpsde_geom_uncond_jodo', [5, 23, 44, 61], 12, f)),
        ('fwd_cond.npz', lambda f: forward_fixture(ref, 'vpsde_qm9_cond_jodo', [4, 9, 18, 18, 27], 13, f)),
        ('fwd_geom384.npz', lambda f: forward_fixture(ref, 'vpsde_geom_uncond_jodo', [5, 23, 44], 14, f, nf=384)),   # BASELINE config 4 width
        ('fwd_geom_big.npz', lambda f: forward_fixture(ref, 'vpsde_geom_uncond_jodo', [100, 181, 140], 15, f)),      # dataset maximum n = 181
        ('fwd_geom384_big.npz', lambda f: forward_fixture(ref, 'vpsde_geom_uncond_jodo', [140, 100, 181], 16, f, nf=384)),
        ('blocks_qm9.npz', lambda f: blocks_fixture(ref, 'vpsde_qm9_uncond_jodo', [3, 9, 17, 29], 17, f)),
        ('blocks_geom.npz', lambda f: blocks_fixture(ref, 'vpsde_geom_uncond_jodo', [12, 33], 18, f)),
        ('fwd_geom_l8.npz', lambda f: forward_fixture(ref, 'vpsde_geom_uncond_jodo', [7, 30, 52, 52, 75], 19, f, n_layers=8)),   # BASELINE configs[2] as worded: nf 256, 8 layers, r 4
        # the README's GEOM "Base" model: --config.model.n_layers 6 --config.model.nf 128 (README.md:150,162); one molecule above a group (n > 128)
        ('fwd_geom_base.npz', lambda f: forward_fixture(ref, 'vpsde_geom_uncond_jodo', [9, 31, 64, 75, 131, 2], 20, f, nf=128, n_layers=6)),
        ('grad_qm9.npz', lambda f: grad_fixture(ref, f)),
        ('traj_qm9_anc5.npz', lambda f: ancestral_fixture(ref, f)),
        # GEOM (3 bond channels: the aromatic decode branch of sampling.py:79-81), short
        ('traj_geom_anc3.npz', lambda f: ancestral_fixture(ref, f, steps=3, n_nodes=(14, 7, 22), seed=30, cfg_name='vpsde_geom_uncond_jodo')),
        ('traj_qm9_anc50.npz', lambda f: ancestral_tf_fixture(ref, f)),
        ('traj_cond_dpm4.npz', lambda f: dpm_fixture(ref, f)),
        ('traj_cond_dpm_multi8.npz', lambda f: dpm_fixture(ref, f, nfe=8, seed=32, method='multistep', order=2)),
        ('traj_cond_dpm_single3.npz', lambda f: dpm_fixture(ref, f, nfe=6, seed=33, method='singlestep_fixed', order=3)),
        ('traj_cond_dpm_single1.npz', lambda f: dpm_fixture(ref, f, nfe=3, seed=34, method='singlestep_fixed', order=1)),
        ('cond_eval.npz', lambda f: cond_eval_fixture(ref, f)),
        # BASELINE configs[0] at its own size (batch 64, 50 steps) through the reference's get_sampling_fn
        ('traj_qm9_cfg0.npz', lambda f: cfg0_fixture(ref, f)),
    ]
    want = sys.argv[1:]
    for fname, job in jobs:
        if not want or any(fname.startswith(w) for w in want):
            job(fname)


if __name__ == '__main__':
    main()
