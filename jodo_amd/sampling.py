"""Sampling procedures that call the score network once per denoising step (host Python).

Same entry points and behaviour as /root/reference/sampling.py for the 3-D + edge ("vpsde_edge")
experiments: `get_sampling_fn` (:148-232), `AncestralSampler` (:518-596), `post_process` (:53-97),
`mol_process` (:12-32).  The model call is unchanged:
    model(vec_t, x, node_mask, edge_mask, edge_x=..., noise_level=..., cond_x=..., cond_edge_x=...,
          context=...)
so any module registered under the reference's names (ours: HIP-backed) plugs in.

Additions that do not change the reference behaviour:
  * `noise_fn` hook on the samplers: parity tests replay recorded noise instead of drawing it;
  * `shard=(rank, world)` on `get_sampling_fn`: each process samples a contiguous slice of every
    round's batch (jodo_amd/dist.py gathers the results) — replaces nn.DataParallel;
  * masks are built vectorised rather than with a Python loop over the batch (:195-196);
  * `hip_graph=True` on `get_sampling_fn`: the ancestral loop replays one captured HIP graph per step.
The 2-D-only sampler (`AncestralSampler_2D`) is out of scope.
"""
import random

import numpy as np
import torch
from torch.nn import functional as F

from .mix_dpm_solver import DPM_Solver_hybrid
from .models.utils import (assert_mean_zero_with_mask, model_hook, sample_combined_position_feature_noise,
                           sample_symmetric_edge_feature_noise)
from .utils import expand_dims, get_self_cond_fn


def mol_process(one_hot, x, formal_charges, n_nodes, edge_types=None):
    """Batch tensors -> list of per-molecule CPU tuples (pos[n,3], atom_type[n], edge_type[n,n], fc[n]).
    One device->host copy per tensor instead of one per molecule."""
    atom_type_all = one_hot.argmax(2).cpu()
    pos_all = x.detach().cpu()
    edge_all = edge_types.detach().cpu() if edge_types is not None else None
    fc_all = formal_charges.detach().cpu()
    mols = []
    for i in range(one_hot.shape[0]):
        n = int(n_nodes[i])
        if edge_all is None:
            mols.append((pos_all[i, :n], atom_type_all[i, :n]))
            continue
        fc = fc_all[i, :n, 0].long() if fc_all.shape[-1] != 0 else fc_all[i][:n]
        mols.append((pos_all[i, :n], atom_type_all[i, :n], edge_all[i, :n, :n], fc))
    return mols


def post_process(xh, atom_types, include_charge, node_mask, inverse_scaler, edge_x=None, edge_mask=None,
                 compress_edge=False):
    """Split xh [B,N,3+atom_types(+1)], undo normalisation, discretise.

    Atom type = argmax; formal charge = round; bonds (compress_edge): exist = ch0 >= 0.5,
    order = bucket(3*ch1) at 0.5/1.5/2.5, aromatic (3-channel data) ch2 >= 0.5 -> type 4 where no
    other order was assigned.  Otherwise: argmax+1 where any channel > 0.5."""
    pos = xh[:, :, :3]
    if include_charge:
        h_int, h_cat = xh[:, :, -1:], xh[:, :, 3:-1]
    else:
        h_int, h_cat = torch.zeros(0).to(xh.device), xh[:, :, 3:]
    assert h_cat.shape[-1] == atom_types

    if edge_x is None:
        pos, h_cat, h_int = inverse_scaler(pos, h_cat, h_int, node_mask)
    else:
        pos, h_cat, h_int, h_edge = inverse_scaler(pos, h_cat, h_int, node_mask, edge_x, edge_mask)
    h_cat = F.one_hot(torch.argmax(h_cat, dim=2), atom_types) * node_mask
    h_int = torch.round(h_int).long() * node_mask
    if edge_x is None:
        return pos, h_cat, h_int

    if compress_edge:
        exist = (h_edge[..., 0] >= 0.5).to(h_edge.dtype)
        o = h_edge[..., 1] * 3.
        order = torch.zeros_like(o)
        order[o >= 0.5] = 1.
        order[o >= 1.5] = 2.
        order[o >= 2.5] = 3.
        order = exist * order
        if h_edge.size(-1) == 3:
            arom = exist * (h_edge[..., 2] >= 0.5).to(h_edge.dtype)
            order[torch.bitwise_and(arom > 0., order == 0.)] = 4.
        h_edge = order
    else:
        any_on = torch.sum(h_edge > 0.5, dim=-1) != 0
        h_edge = any_on * (torch.argmax(h_edge, dim=-1) + 1.0)
    return pos, h_cat, h_int, h_edge


def build_masks(n_nodes, max_n_nodes, device):
    """node_mask [B,N,1], edge_mask [B*N*N,1] (prefix masks, diagonal removed) — sampling.py:193-201."""
    n = torch.as_tensor(n_nodes, dtype=torch.long)
    node_mask = (torch.arange(max_n_nodes).unsqueeze(0) < n.unsqueeze(1)).float()
    edge_mask = node_mask.unsqueeze(1) * node_mask.unsqueeze(2)
    edge_mask = edge_mask * (~torch.eye(max_n_nodes, dtype=torch.bool)).unsqueeze(0)
    bs = node_mask.size(0)
    return node_mask.unsqueeze(2).to(device), edge_mask.view(bs * max_n_nodes * max_n_nodes, 1).to(device)


class _ParityNoise:
    """Noise of the UNSHARDED batch replayed on every rank (shard_mode='parity'): each draw is made on the CPU with
    the full round's shapes, in the reference's draw order (sampling.py:204-209 initial noise, :576 / :588 per step,
    mix_dpm_solver.py:56 positions), then cut down to this rank's molecules [lo:hi] and its own padded width.  Every
    rank consumes the global CPU generator identically, so the slices are a partition of ONE draw and a world-size-N
    run reproduces the world-size-1 run molecule for molecule."""

    def __init__(self, n_nodes_round, lo, hi, node_nf, edge_nf, device):
        self.bs, self.N = len(n_nodes_round), int(max(n_nodes_round))
        self.lo, self.hi = lo, hi
        self.Nl = int(max(n_nodes_round[lo:hi])) if hi > lo else 0
        self.node_nf, self.edge_nf, self.device = node_nf, edge_nf, device
        self.nm, self.em = build_masks(n_nodes_round, self.N, 'cpu')

    def _cut(self, t, dims):
        if self.hi == self.lo:
            return None
        t = t[self.lo:self.hi, :self.Nl] if dims == 1 else t[self.lo:self.hi, :self.Nl, :self.Nl]
        return t.contiguous().to(self.device)

    def node(self):
        return self._cut(sample_combined_position_feature_noise(self.bs, self.N, self.node_nf, self.nm), 1)

    def edge(self):
        return self._cut(sample_symmetric_edge_feature_noise(self.bs, self.N, self.edge_nf, self.em), 2)

    def pos(self):
        from .models.utils import sample_center_gravity_zero_gaussian_with_mask
        return self._cut(sample_center_gravity_zero_gaussian_with_mask((self.bs, self.N, 3), 'cpu', self.nm), 1)

    def __call__(self, step, kind, like):             # noise_fn hook of AncestralSampler / DPM_Solver_hybrid
        return {'node': self.node, 'edge': self.edge, 'pos': self.pos}[kind]()


def get_sampling_fn(config, noise_scheduler, nodes_dist, batch_size, n_samples, inverse_scaler, eps=1e-3,
                    prop_dist=None, shard=None, return_raw=False, fused_decode=True, hip_graph=False,
                    shard_mode='perf', shard_assign='contiguous', seed=None, device_noise=None):
    """sampling.py:148-232.  Without `shard` this is the reference's procedure, RNG use included.

    shard=(rank, world) — one process per GPU (replaces nn.DataParallel).  Seeding contract: every rank passes the SAME
    `seed` (default config.seed).  The atom counts of all rounds and the conditioning context are drawn once from
    `torch.manual_seed(seed)` on every rank (identical lists, no communication).  Then
      shard_mode='perf'   the rounds * batch_size molecules are dealt to the ranks FIRST (contiguous slices, or
                          shard_assign='lpt': greedy longest-processing-time by n^2 for balance) and each rank cuts its
                          share into rounds of up to batch_size (BASELINE config 5: 10 000 molecules on 8 GPUs = one round
                          of 1250 per GPU, not four rounds of 313); the initial noise comes from the generator seeded
                          with a 63-bit hash of (seed, 1 + rank) (fused.mix64: unrelated streams for different pairs of any
                          size, none of them the atom-count stream `seed`), the per-step noise is drawn inside the fused
                          update kernels (device_noise, below) keyed by a hash of (seed, rank, round);
      shard_mode='parity' every round of the unsharded run is split across the ranks and the unsharded run's noise is
                          replayed (_ParityNoise): the gathered result equals the world-size-1 run.  Slow (full-batch CPU
                          draws per step) — for tests.
    device_noise — True: the per-step normal draws are generated inside the fused update kernels (Philox keyed by
    (seed, rank, round), jodo_sampler_step_rng / jodo_dpm_update_rng) instead of three torch.randn launches per step; same
    distribution, a different stream than the reference's.  Default: on for shard_mode='perf', off otherwise (an unsharded
    seeded run consumes torch's generator exactly like the reference).  Ignored on CPU tensors and when noise is replayed.
    Returns this rank's molecules; `sampling_fn.last_indices` holds their indices in the global order (rounds *
    batch_size molecules, as the unsharded run generates them) and `sampling_fn.last_decoded` the decoded, padded tensors of
    every round as they left the decode (on the GPU: the outputs of jodo_decode, still on the device) — what the caller's
    gather sends (jodo_amd/dist.gather_sampled)."""
    device = config.device
    steps = config.sampling.steps
    atom_types = config.data.atom_types
    include_fc = config.model.include_fc_charge
    node_nf = atom_types + int(include_fc)
    pred_edge = config.pred_edge
    edge_nf = config.model.edge_ch
    compress_edge = config.data.compress_edge
    if config.only_2D or not pred_edge:
        raise NotImplementedError("only the 3-D + edge (vpsde_edge) sampling path is in scope")
    if shard_mode not in ('perf', 'parity') or shard_assign not in ('contiguous', 'lpt'):
        raise ValueError("shard_mode in {'perf','parity'}, shard_assign in {'contiguous','lpt'}")
    if shard is not None and not (0 <= shard[0] < shard[1]):
        raise ValueError("shard=(rank, world) with 0 <= rank < world")

    if device_noise is None:
        device_noise = shard is not None and shard_mode == 'perf'
    rounds = int(np.ceil(n_samples / batch_size))
    round_counter = [0]
    if config.sampling.method == 'ancestral':
        # schedule scalars are computed on the host: the VP coefficients near t -> 0 are ill-conditioned
        # in fp32 (sigma_s = sqrt(1 - exp(2 log alpha_s)) with log alpha_s ~ 1e-7), so their low bits depend
        # on the math library; keeping them on the CPU makes trajectories reproducible across devices
        time_steps = torch.linspace(noise_scheduler.T, eps, steps)
        sampler = AncestralSampler(noise_scheduler, time_steps, config.model.pred_data, pred_edge,
                                   config.model.self_cond, get_self_cond_fn(config))
    elif config.sampling.method == 'fast':
        sampler = DPM_Solver_hybrid(noise_scheduler, config)
    else:
        raise ValueError('Invalid sampling method!')

    def one_round(model, n_nodes, context, noise=None):
        """n_nodes molecules (this process's share of a round) -> list of decoded molecule tuples."""
        bs = len(n_nodes)
        max_n = int(max(n_nodes))
        node_mask, edge_mask = build_masks(n_nodes, max_n, device)
        if noise is None:
            z = sample_combined_position_feature_noise(bs, max_n, node_nf, node_mask)
            edge_z = sample_symmetric_edge_feature_noise(bs, max_n, edge_nf, edge_mask)
        else:
            z, edge_z = noise.node(), noise.edge()
        assert_mean_zero_with_mask(z[:, :, :3], node_mask)
        sampler.noise_fn = noise
        sampler.device_noise = None
        if device_noise and noise is None and z.is_cuda:
            from . import fused
            sampler.device_noise = fused.DeviceNoise.for_rank(int(config.seed if seed is None else seed),
                                                              shard[0] if shard is not None else 0, round_counter[0])
        round_counter[0] += 1
        try:
            if hip_graph and noise is None and z.is_cuda and isinstance(sampler, AncestralSampler):
                # one captured HIP graph per round, replayed for every step (jodo_amd/graphed.py)
                from .graphed import GraphedAncestralRound
                x_node, x_edge = GraphedAncestralRound(sampler, model, node_mask, edge_mask, context).run(z, edge_z)
            elif (hip_graph and noise is None and z.is_cuda and isinstance(sampler, DPM_Solver_hybrid)
                  and sampler.method == 'singlestep_fixed' and sampler.order == 2):
                from .graphed import GraphedDPMRound
                x_node, x_edge = GraphedDPMRound(sampler, model, node_mask, edge_mask, context).run(z, edge_z)
            else:
                x_node, x_edge = sampler.sampling(model, z, node_mask, edge_mask, edge_z, context)
        finally:
            sampler.noise_fn = None
            sampler.device_noise = None
            unpin = model_hook(model, 'unpin_paths')       # (the graph-replayed rounds pin too)
            if unpin is not None:
                unpin()
        _warn_nan(model)
        if x_node.is_cuda and fused_decode and getattr(inverse_scaler, 'from_config', False):
            # device-side decode: compact u8/i8 results, one device->host copy per tensor
            from . import fused
            nd = fused.n_nodes_from_mask(node_mask)
            dec = fused.decode(config, x_node, x_edge, nd)
            if shard is not None:                              # the gather takes the decoded DEVICE tensors as they are (dist.gather_sampled)
                decoded_rounds.append(dec + (nd,))
            return fused.mols_from_decoded(*dec, n_nodes)
        pos, one_hot, fc, edge_types = post_process(x_node, atom_types, include_fc, node_mask, inverse_scaler, x_edge,
                                                    edge_mask, compress_edge)
        assert_mean_zero_with_mask(pos, node_mask)
        if shard is not None:
            fcq = fc[..., 0] if fc.shape[-1] != 0 else torch.zeros(fc.shape[:2], device=fc.device)
            decoded_rounds.append((pos, one_hot.argmax(2).to(torch.uint8), fcq.round().to(torch.int8), edge_types.to(torch.uint8),
                                   torch.as_tensor([int(n) for n in n_nodes], dtype=torch.int32, device=pos.device)))
        return mol_process(one_hot, pos, fc, n_nodes, edge_types)

    decoded_rounds = []          # per round of a sharded run: (pos [B,N,3] f32, atom_type [B,N] u8, charge [B,N] i8, bond [B,N,N] u8, n_nodes [B] i32)

    def sampling_fn(model):
        model.eval()
        mols = []
        del decoded_rounds[:]
        sampling_fn.last_decoded = decoded_rounds
        total = rounds * batch_size
        with torch.no_grad():
            if shard is None:                                  # the reference's procedure
                n_nodes_all = nodes_dist.sample(total)
                for r in range(rounds):
                    n_nodes = n_nodes_all[r * batch_size:(r + 1) * batch_size]
                    context = prop_dist.sample_batch(n_nodes).to(device) if prop_dist is not None else None
                    mols += one_round(model, n_nodes, context)
                    print('Generate {}, Total {}.'.format(len(mols), n_samples))
                sampling_fn.last_indices = list(range(total))
                if return_raw:
                    return mols
                random.shuffle(mols)
                return mols[:n_samples]

            rank, world = shard
            base = int(config.seed if seed is None else seed)
            torch.manual_seed(base)                            # identical on every rank
            n_nodes_all = nodes_dist.sample(total)
            if shard_mode == 'perf':
                context_all = prop_dist.sample_batch(n_nodes_all) if prop_dist is not None else None
                from .dist import assign_lpt, shard_range
                if shard_assign == 'lpt':
                    mine = assign_lpt(n_nodes_all.tolist(), world)[rank]
                else:
                    lo, hi = shard_range(total, rank, world)
                    mine = list(range(lo, hi))
                from .fused import mix64
                torch.manual_seed(mix64(base, 1 + rank) >> 1)  # this rank's own stream (CPU and device generators): a 63-bit
                                                               # hash of (seed, rank), any seed / rank size
                for r0 in range(0, len(mine), batch_size):
                    idx = torch.as_tensor(mine[r0:r0 + batch_size], dtype=torch.long)
                    context = context_all[idx].to(device) if context_all is not None else None
                    mols += one_round(model, n_nodes_all[idx], context)
                    if rank == 0:
                        print('Generate {} on rank 0, Total {}.'.format(len(mols), n_samples))
                sampling_fn.last_indices = list(mine)
            else:                                              # parity: replay the unsharded run's draws
                from .dist import shard_range
                indices = []
                for r in range(rounds):
                    n_nodes = n_nodes_all[r * batch_size:(r + 1) * batch_size]
                    context = prop_dist.sample_batch(n_nodes) if prop_dist is not None else None
                    lo, hi = shard_range(len(n_nodes), rank, world)
                    noise = _ParityNoise(n_nodes.tolist(), lo, hi, node_nf, edge_nf, device)
                    if hi == lo:                               # nothing of this round is ours: stay in step with the stream
                        _consume_round_noise(noise, sampler, steps)
                        continue
                    ctx = context[lo:hi].to(device) if context is not None else None
                    mols += one_round(model, n_nodes[lo:hi], ctx, noise)
                    indices += list(range(r * batch_size + lo, r * batch_size + hi))
                sampling_fn.last_indices = indices
        return mols                                            # caller gathers / shuffles

    sampling_fn.last_indices = None
    sampling_fn.last_decoded = decoded_rounds
    return sampling_fn


def full_edge_index(n_nodes, batch_size, device):
    """cond_gen/utils.py:15-38 (get_adj_matrix_fn): the fully connected edge list of `batch_size` graphs of `n_nodes` nodes, self-loops
    included, batch b offset by b * n_nodes, row-major (i, j) order — what the property classifier's EGNN is called with."""
    idx = torch.arange(n_nodes, device=device)
    rows = idx.repeat_interleave(n_nodes).unsqueeze(0) + (torch.arange(batch_size, device=device) * n_nodes).unsqueeze(1)
    cols = idx.repeat(n_nodes).unsqueeze(0) + (torch.arange(batch_size, device=device) * n_nodes).unsqueeze(1)
    return [rows.reshape(-1), cols.reshape(-1)]


def get_cond_sampling_eval_fn(config, noise_scheduler, nodes_dist, batch_size, n_samples, inverse_scaler, eps=1e-3,
                              prop_dist=None, prop_norm=None):
    """sampling.py:283-392: conditional sampling scored by a property classifier.  Returns sampling_fn(model, classifier) ->
    (molecules[:n_samples], mean |classifier(sample) - target| * outputNorm[cond_property]) like the reference; the classifier is the
    caller's (the reference's pretrained EGNN, cond_gen/ — out of this path's scope) and is called with the reference's keywords
    (h0, x, edges, edge_attr, node_mask, edge_mask, n_nodes).  The score network runs through the HIP kernels as in get_sampling_fn;
    the round's RNG use (atom counts, context, initial noise, per-step draws) is the reference's."""
    device = config.device
    steps = config.sampling.steps
    atom_types = config.data.atom_types
    include_fc = config.model.include_fc_charge
    node_nf = atom_types + int(include_fc)
    edge_nf = config.model.edge_ch
    compress_edge = config.data.compress_edge
    if config.only_2D or not config.pred_edge:
        raise NotImplementedError("only the 3-D + edge (vpsde_edge) sampling path is in scope")
    if config.sampling.method != 'ancestral':
        raise ValueError('Invalid sampling method!')                  # sampling.py:305
    prop = config.cond_property
    mean, mad = prop_norm[prop]['mean'], prop_norm[prop]['mad']
    output_norm = {'mu': 1., 'alpha': 1, 'homo': 1000., 'lumo': 1000., 'gap': 1000, 'Cv': 1.}
    rounds = int(np.ceil(n_samples / batch_size))
    time_steps = torch.linspace(noise_scheduler.T, eps, steps)
    sampler = AncestralSampler(noise_scheduler, time_steps, config.model.pred_data, config.pred_edge, config.model.self_cond,
                               get_self_cond_fn(config))

    def sampling_fn(model, classifier):
        model.eval()
        classifier.eval()
        mols, maes = [], []
        with torch.no_grad():
            n_nodes_all = nodes_dist.sample(rounds * batch_size)
            for r in range(rounds):
                n_nodes = n_nodes_all[r * batch_size:(r + 1) * batch_size]
                max_n = int(max(n_nodes))
                context = prop_dist.sample_batch(n_nodes).to(device) if prop_dist is not None else None
                node_mask, edge_mask = build_masks(n_nodes, max_n, device)
                z = sample_combined_position_feature_noise(batch_size, max_n, node_nf, node_mask)
                assert_mean_zero_with_mask(z[:, :, :3], node_mask)
                edge_z = sample_symmetric_edge_feature_noise(batch_size, max_n, edge_nf, edge_mask)
                try:
                    x_node, x_edge = sampler.sampling(model, z, node_mask, edge_mask, edge_z, context)
                finally:
                    unpin = model_hook(model, 'unpin_paths')
                    if unpin is not None:
                        unpin()
                _warn_nan(model)
                pos, one_hot, fc, edge_types = post_process(x_node, atom_types, include_fc, node_mask, inverse_scaler, x_edge,
                                                            edge_mask, compress_edge)
                assert_mean_zero_with_mask(pos, node_mask)
                bs, b_node, _ = pos.size()
                pred = classifier(h0=one_hot.reshape(bs * b_node, -1), x=pos.reshape(bs * b_node, -1),
                                  edges=full_edge_index(b_node, batch_size, device), edge_attr=None,
                                  node_mask=node_mask.reshape(bs * b_node, -1), edge_mask=edge_mask, n_nodes=b_node)
                assert context.size(-1) == 1
                target = context.clone().squeeze(-1) * mad + mean
                maes.append((pred * mad + mean - target).abs())
                mols += mol_process(one_hot, pos, fc, n_nodes, edge_types)
                print('Generate {}, Total {}.'.format(len(mols), n_samples))
        mae = torch.cat(maes)[:n_samples]
        return mols[:n_samples], mae.mean().item() * output_norm[prop]

    return sampling_fn


def _consume_round_noise(noise, sampler, steps):
    """Advance the CPU generator by exactly the draws one round makes (a rank whose slice of a round is empty)."""
    noise.node(); noise.edge()
    if isinstance(sampler, AncestralSampler):
        for _ in range(steps):
            noise.node(); noise.edge()
    else:
        for _ in range(sampler.noise_draws_per_round()):
            noise.pos()


def _warn_nan(model):
    """The reference prints 'Warning: detected nan, resetting output to zero.' in every forward that hits the guard
    (mol_gnn.py:587-589).  The kernels keep a sticky device counter instead of a host sync per step; it is read once
    per round here (one .item())."""
    take = model_hook(model, 'take_nan_count')
    if take is not None:
        n = take()
        if n:
            print('Warning: detected nan in %d score-network evaluation(s) of this round, position outputs were reset to zero.' % n)


def posterior_coefficients(ns, t, s):
    """Ancestral p(x_s | x_t, x0_pred) for a VP process: returns (c_x, c_pred, sigma) with
    x_s = c_x * x_t + c_pred * x0_pred + sigma * eps   (data-prediction form, sampling.py:536-572)."""
    alpha_t, sigma_t = ns.marginal_prob(t)
    alpha_s, sigma_s = ns.marginal_prob(s)
    a_ts = alpha_t / alpha_s
    var_ts = sigma_t ** 2 - a_ts ** 2 * sigma_s ** 2
    sigma = torch.sqrt(var_ts) * sigma_s / sigma_t
    return a_ts * sigma_s ** 2 / sigma_t ** 2, alpha_s * var_ts / sigma_t ** 2, sigma, alpha_t, sigma_t, a_ts, var_ts


class AncestralSampler:
    """Ancestral sampling for joint 2-D & 3-D generation; returns the noise-free mean of the last step."""

    def __init__(self, noise_scheduler, time_steps, model_pred_data, pred_edge=False, self_cond=False,
                 cond_process_fn=None, noise_fn=None, fused=True, device_noise=None):
        self.noise_scheduler = noise_scheduler
        # device_noise: a fused.DeviceNoise — the per-step normal draws are generated inside the fused update kernel
        # (counter-based Philox; same distribution, not the reference's stream); None = torch.randn draws in the
        # reference's shapes and order, so that seeded runs consume the generator exactly like the reference
        self.device_noise = device_noise
        # fused: on GPU tensors the update + noise construction run as one HIP kernel per tensor
        # (csrc/sampler_kernels.hip) instead of ~25 framework launches; same RNG draws in the same order
        self.fused = fused
        self.t_array = time_steps
        self.s_array = torch.cat([time_steps[1:], torch.zeros(1, device=time_steps.device)])
        self.model_pred_data = model_pred_data
        self.pred_edge = pred_edge
        self.self_cond = self_cond
        self.cond_process_fn = cond_process_fn
        self.noise_fn = noise_fn          # optional: noise_fn(step, kind, shape_like) -> tensor

    def _node_noise(self, i, x_mean, node_mask):
        if self.noise_fn is not None:
            return self.noise_fn(i, 'node', x_mean)
        return sample_combined_position_feature_noise(x_mean.shape[0], x_mean.shape[1], x_mean.shape[2] - 3,
                                                      node_mask)

    def _edge_noise(self, i, edge_mean, edge_mask):
        if self.noise_fn is not None:
            return self.noise_fn(i, 'edge', edge_mean)
        return sample_symmetric_edge_feature_noise(edge_mean.shape[0], edge_mean.shape[1], edge_mean.shape[-1],
                                                   edge_mask)

    def init_state(self, z_T, edge_z_T):
        return dict(x=z_T, edge_x=edge_z_T, cond_x=None, cond_edge_x=None, x_mean=None, edge_x_mean=None)

    def step(self, model, i, st, node_mask, edge_mask, context=None):
        """One denoising step i (model evaluation + ancestral update) on the state dict `st`."""
        ns = self.noise_scheduler
        x, edge_x = st['x'], st['edge_x']
        bs = x.shape[0]
        t, s = self.t_array[i], self.s_array[i]
        c_x, c_pred, sigma, alpha_t, sigma_t, a_ts, var_ts = posterior_coefficients(ns, t, s)
        vec_t = torch.ones(bs, device=x.device) * t
        noise_level = torch.ones(bs, device=x.device) * torch.log(alpha_t ** 2 / sigma_t ** 2)
        if self.self_cond:
            assert self.model_pred_data
            pred_t, edge_pred_t = model(vec_t, x, node_mask, edge_mask, edge_x=edge_x, noise_level=noise_level,
                                        cond_x=st['cond_x'], cond_edge_x=st['cond_edge_x'], context=context)
            if i == 1 and x.is_cuda:
                # first self-conditioned evaluation of the round: from here on the inputs keep their structure (symmetric
                # state and predictions, one noise level per batch), so the HIP model may stop launching the kernel
                # variants its device flags rule out (jodo_amd/models/dgt.py pin_paths; one host sync per round)
                pin = model_hook(model, 'pin_paths')
                if pin is not None:
                    pin()
            st['cond_x'], st['cond_edge_x'] = self.cond_process_fn(pred_t, edge_pred_t)
        else:
            pred_t, edge_pred_t = model(vec_t, x, node_mask, edge_mask, edge_x=edge_x, noise_level=noise_level,
                                        context=context)
        if self.fused and x.is_cuda and self.model_pred_data:
            from . import fused
            if st.get('_bufs') is None:
                st['_bufs'] = fused.StepBuffers(x, edge_x)
                st['_n_nodes'] = fused.n_nodes_from_mask(node_mask)
            N, nd, ch = x.shape[1], x.shape[2] - 3, edge_x.shape[-1]
            if self.noise_fn is not None:
                # replayed draws (parity runs, tests against the reference's recorded trajectories): the recorded noise is
                # already masked / CoM-free / symmetric and the kernel's own masking, CoM removal and mirroring are
                # idempotent on it; node noise first, then edge noise (the reference's draw order)
                node_eps = self.noise_fn(i, 'node', x)
                eps = fused.split_replayed_noise(node_eps, self.noise_fn(i, 'edge', edge_x))
                rng = None
            elif self.device_noise is not None:
                eps, rng = (None, None, None), self.device_noise       # drawn inside the kernel (Philox)
            else:
                eps = (torch.randn((bs, N, 3), device=x.device),           # draw order of models/utils.py:67-99
                       torch.randn((bs, N, nd), device=x.device),
                       torch.randn((bs, ch, N, N), device=x.device))
                rng = None
            st['x'], st['edge_x'], st['x_mean'], st['edge_x_mean'] = fused.sampler_step(
                st['_bufs'], st['_n_nodes'], float(c_x), float(c_pred), float(sigma), x, edge_x, pred_t, edge_pred_t,
                *eps, rng=rng)
            return st
        # the coefficients are scalars shared by the batch (the reference broadcasts them through
        # .repeat(bs) + expand_dims, sampling.py:569-589 — same values, same products)
        if self.model_pred_data:
            x_mean = c_x * x + c_pred * pred_t
            edge_x_mean = c_x * edge_x + c_pred * edge_pred_t
        else:
            k = var_ts / a_ts / sigma_t
            x_mean = x / a_ts - k * pred_t
            edge_x_mean = edge_x / a_ts - k * edge_pred_t
        # RNG order: node noise, then edge noise (as the reference)
        st['x'] = x_mean + sigma * self._node_noise(i, x_mean, node_mask)
        st['edge_x'] = edge_x_mean + sigma * self._edge_noise(i, edge_x_mean, edge_mask)
        st['x_mean'], st['edge_x_mean'] = x_mean, edge_x_mean
        return st

    def sampling(self, model, z_T, node_mask, edge_mask, edge_z_T=None, context=None):
        if not self.pred_edge:
            raise NotImplementedError("edge-free sampling is out of scope")
        st = self.init_state(z_T, edge_z_T)
        try:
            for i in range(len(self.t_array)):
                st = self.step(model, i, st, node_mask, edge_mask, context)
        finally:
            unpin = model_hook(model, 'unpin_paths')       # the pin of step 1 is scoped to this round
            if unpin is not None:
                unpin()
        assert_mean_zero_with_mask(st['x_mean'][:, :, :3], node_mask)
        return st['x_mean'], st['edge_x_mean']
