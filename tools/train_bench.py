"""Training-step timing of the HIP score network: get_step_fn (jodo_amd/losses.py) on a synthetic batch of
config.training.batch_size molecules with the dataset's atom-count histogram — per-molecule noise levels, the 50 % self-conditioning
double forward, dropout 0.1, loss.backward() through jodo_train_backward, AdamW + clipping + EMA.  Prints one JSON line.

    python tools/train_bench.py [--workload qm9|geom] [--batch B] [--steps K] [--warmup W]

Work model: a grad-enabled forward + backward costs 3 x the forward's projection flops (forward, input gradient, weight gradient);
the no-grad self-conditioning forward (half of the steps) one more; `gemm_flops_per_step` prices the projections only
(oracle.algorithmic_flops with per-molecule time rows), against the fp32 MFMA peak."""
import argparse
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synthetic_batch(cfg, n_nodes, seed):
    from jodo_amd.sampling import build_masks
    g = torch.Generator().manual_seed(seed)
    B, N = len(n_nodes), max(n_nodes)
    nm, em = build_masks(n_nodes, N, 'cpu')
    at = torch.randint(0, cfg.data.atom_types, (B, N), generator=g)
    nb = 4 if cfg.model.edge_ch == 2 else 5
    bond = torch.triu(torch.randint(0, nb, (B, N, N), generator=g), 1)
    bond = bond + bond.transpose(1, 2)
    if cfg.model.edge_ch == 2:
        eoh = torch.stack([(bond > 0).float(), bond.clamp(max=3).float() / 3.], -1)
    else:
        eoh = torch.stack([(bond > 0).float(), bond.clamp(max=3).float() / 3., (bond == 4).float()], -1)
    return dict(positions=torch.randn(B, N, 3, generator=g) * nm, atom_mask=nm[..., 0], edge_mask=em,
                atom_one_hot=torch.nn.functional.one_hot(at, cfg.data.atom_types).float() * nm,
                edge_one_hot=eoh * em.reshape(B, N, N, 1), formal_charges=torch.randint(-1, 2, (B, N, 1), generator=g).float() * nm)


def _same_coin_flips(seed, steps):
    """loss_fn draws random.random() once per step for the 50 % self-conditioning forward: every timed leg starts from the same seed, so
    all legs (and all A/B runs) time the same number of double forwards.  Returns that number."""
    random.seed(seed)
    n = sum(random.random() < 0.5 for _ in range(steps))
    random.seed(seed)
    return n


def run(workload='qm9', batch=0, steps=10, warmup=3, seed=42, options=None, save_always=False):
    from jodo_amd import configs, losses as L
    from jodo_amd.diffusion import NoiseScheduleVP
    from jodo_amd.models import get_model_class, deterministic_init_, load_dataset_info, get_node_dist
    from jodo_amd.models.ema import ExponentialMovingAverage
    from jodo_amd.utils import get_data_scaler
    name, info = dict(qm9=('vpsde_qm9_uncond_jodo', 'qm9_with_h'), geom=('vpsde_geom_uncond_jodo', 'geom_with_h_1'))[workload]
    cfg = configs.get(name)
    dev = torch.device('cuda:0')
    cfg.device = dev
    B = batch or int(cfg.training.batch_size)
    torch.manual_seed(seed)
    random.seed(seed)
    n_nodes = get_node_dist(load_dataset_info(info)).sample(B).tolist()
    data = synthetic_batch(cfg, n_nodes, seed)
    model = deterministic_init_(get_model_class(cfg.model.name)(cfg), seed=seed).to(dev)
    model.train_options = dict(options or {})          # jodo_train_set_option values (A/B runs), e.g. {0: 0, 1: 0} = op-by-op
    model.train_save_always = bool(save_always)
    ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
    state = dict(model=model, optimizer=L.get_optimizer(cfg, model.parameters()),
                 ema=ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_decay), step=1)
    step_fn = L.get_step_fn(ns, True, L.optimization_manager(cfg), get_data_scaler(cfg), cfg)
    losses = []
    for _ in range(warmup):
        losses.append(float(step_fn(state, data)))
    torch.cuda.synchronize()
    n_selfcond = _same_coin_flips(seed + 7, steps)
    t0 = time.perf_counter()
    # (the loss stays on the device inside the timed loops: the reference's loop reads it every `log_freq` steps, run_lib.py, not every
    # step — a float() per step would put a host synchronisation into every step that a training run does not have)
    held = []
    for _ in range(steps):
        held.append(step_fn(state, data).detach())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    losses += [float(x) for x in held]
    # What a real training loop sees: a shuffling loader hands over NEW atom counts every step, so the engine cache of
    # models/dgt.py never hits — every step pays jodo_train_create (host tables, upload) and the host transfer of the counts.
    # Batches are generated up front and already on the device (the loader itself is outside the path).
    dist_ = get_node_dist(load_dataset_info(info))
    fresh = []
    for k in range(warmup + steps):
        nk = dist_.sample(B).tolist()
        fresh.append({kk: v.to(dev) for kk, v in synthetic_batch(cfg, nk, seed + 1 + k).items()})
    for k in range(warmup):
        float(step_fn(state, fresh[k]))
    torch.cuda.synchronize()
    _same_coin_flips(seed + 7, steps)
    t0 = time.perf_counter()
    held = []
    for k in range(warmup, warmup + steps):
        held.append(step_fn(state, fresh[k]).detach())
    torch.cuda.synchronize()
    dt_fresh = (time.perf_counter() - t0) / steps
    losses += [float(x) for x in held]
    # The reference's loop (run_lib.py:346-351): the loader yields CPU dicts, process_edge_batch moves them to the device inside the
    # step.  Atom counts and the mask check are then done on the host copy (jodo_amd/losses.py _host_counts), the upload goes through
    # pinned memory, and — with the clipping history on the device and the loss read every log_freq steps only — a step has NO host
    # synchronisation: the host queues ahead of the card.
    fresh_cpu = [synthetic_batch(cfg, dist_.sample(B).tolist(), seed + 101 + k) for k in range(warmup + steps)]
    for k in range(warmup):
        float(step_fn(state, fresh_cpu[k]))
    torch.cuda.synchronize()
    _same_coin_flips(seed + 7, steps)
    t0 = time.perf_counter()
    held = []
    for k in range(warmup, warmup + steps):
        held.append(step_fn(state, fresh_cpu[k]).detach())
    torch.cuda.synchronize()
    dt_loader = (time.perf_counter() - t0) / steps
    losses += [float(x) for x in held]
    del fresh_cpu
    # engine creation alone (host side of a new batch): n_host -> handle + tables + upload
    tcr = time.perf_counter()
    for k in range(3):
        model._train_engines.clear()
        model._train_engine(fresh[k]['atom_mask'].unsqueeze(-1), fresh[k]['edge_mask'], dev)
    torch.cuda.synchronize()
    create_ms = (time.perf_counter() - tcr) / 3 * 1e3
    del fresh
    # where a step's wall time goes (host-synchronised sections over the same step, 6 repetitions)
    loss_fn = L.get_sde_graph_loss_fn(ns, True, get_data_scaler(cfg), cfg, None)
    opt_fn = L.optimization_manager(cfg)
    sec = dict(loss_forward=0.0, backward=0.0, clip_and_optimizer=0.0, ema=0.0)
    reps = 6
    for _ in range(reps):
        torch.cuda.synchronize(); t = [time.perf_counter()]
        state['optimizer'].zero_grad()
        loss = loss_fn(model, data)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        loss.backward()
        torch.cuda.synchronize(); t.append(time.perf_counter())
        opt_fn(state['optimizer'], model.parameters(), step=state['step'])
        torch.cuda.synchronize(); t.append(time.perf_counter())
        state['ema'].update(model.parameters())
        torch.cuda.synchronize(); t.append(time.perf_counter())
        for k, a, b in zip(sec, t, t[1:]):
            sec[k] += (b - a) * 1e3 / reps
    # forward / backward alone (no self-conditioning forward, no optimiser), HIP events
    from jodo_amd.sampling import build_masks
    nm, em = build_masks(n_nodes, max(n_nodes), dev)
    xh = torch.randn(B, max(n_nodes), 3 + model.dims.nd, device=dev) * nm
    ex = torch.randn(B, max(n_nodes), max(n_nodes), model.dims.ch, device=dev)
    ex = (ex + ex.transpose(1, 2)) * em.reshape(B, max(n_nodes), max(n_nodes), 1)
    nl = torch.randn(B, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    fwd = bwd = float('inf')
    for _ in range(4):
        model.zero_grad()
        ev[0].record()
        ox, oe = model(nl, xh, nm, em, edge_x=ex, cond_x=None, cond_edge_x=None, noise_level=nl)
        ev[1].record()
        (ox.square().sum() + oe.square().sum()).backward()
        ev[2].record()
        torch.cuda.synchronize()
        fwd, bwd = min(fwd, ev[0].elapsed_time(ev[1])), min(bwd, ev[1].elapsed_time(ev[2]))       # (best of four: host jitter between the two calls counts here)
    sys.path.insert(0, ROOT)
    from oracle import dgt_oracle as O                      # work model only (checker-side formulas)
    hp = O.Hyper.from_config(cfg)
    f_fwd = O.algorithmic_flops(hp, n_nodes, shared_time=False)['total']
    peak = 157.3e12
    return dict(workload=name, batch=B, steps=steps, s_per_step=dt_loader, molecules_per_s=B / dt_loader,
                loader_batches=dict(s_per_step=dt_loader, molecules_per_s=B / dt_loader,
                                    note='every step a new CPU batch dict, moved to the device inside the step as in the reference loop '
                                         '(run_lib.py:346-351, losses.py:470-476); no host synchronisation in a step: the headline'),
                fresh_batches=dict(s_per_step=dt_fresh, molecules_per_s=B / dt_fresh, engine_create_ms=create_ms,
                                   note='every step a new draw of atom counts, batches already on the device (the counts are read back: '
                                        'one synchronisation per step): round 5\'s earlier headline'),
                fixed_batch=dict(s_per_step=dt, molecules_per_s=B / dt, note='one batch repeated (engine cache hits): round 4\'s figure'),
                self_conditioned_steps='%d of %d in every timed leg' % (n_selfcond, steps),
                loss_first=losses[0], loss_last=losses[-1],
                forward_ms=fwd, backward_ms=bwd, sections_ms=sec, forward_algorithmic_flops=f_fwd,
                forward_frac_of_fp32_mfma_peak=f_fwd / (fwd * 1e-3) / peak, backward_frac_of_fp32_mfma_peak=2 * f_fwd / (bwd * 1e-3) / peak,
                note='one optimiser step = (50 %: no-grad self-conditioning forward) + grad forward + backward + AdamW / clipping / EMA; '
                     'forward_ms / backward_ms: one grad-enabled forward and its backward alone (HIP events); fractions price the '
                     'SURVEY.md 8d forward count (x 2 for the two gradient products of every projection) against 157.3 TFLOP/s — '
                     'first-correct kernels (train_ops.h), not yet tuned')


def cpu_step(workload='qm9', batch=16, steps=2, seed=42, threads=16):
    """The same optimiser-free step on the host: forward + loss.backward() through the port of the reference's sparse formulation
    (oracle.forward_faithful under torch.autograd, eval-mode dropout) — what the reference's CPU training step costs, per molecule."""
    from jodo_amd import configs
    from jodo_amd.models import get_model_class, deterministic_init_, load_dataset_info, get_node_dist
    from jodo_amd.sampling import build_masks
    from oracle import dgt_oracle as O
    name, info = dict(qm9=('vpsde_qm9_uncond_jodo', 'qm9_with_h'), geom=('vpsde_geom_uncond_jodo', 'geom_with_h_1'))[workload]
    cfg = configs.get(name)
    torch.manual_seed(seed)
    torch.set_num_threads(threads)
    n_nodes = get_node_dist(load_dataset_info(info)).sample(batch).tolist()
    model = deterministic_init_(get_model_class(cfg.model.name)(cfg), seed=seed)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    hp = O.Hyper.from_config(cfg)
    N = max(n_nodes)
    nm, em = build_masks(n_nodes, N, 'cpu')
    xh = torch.randn(batch, N, 3 + hp.in_node_dim) * nm
    ex = torch.randn(batch, N, N, hp.edge_ch)
    ex = (ex + ex.transpose(1, 2)) * em.reshape(batch, N, N, 1)
    nl = torch.randn(batch)
    ts = []
    for _ in range(steps + 1):
        t0 = time.perf_counter()
        ox, oe = O.forward_faithful(sd, hp, xh, nm, em, ex, None, None, nl, None)
        (ox.square().sum() + oe.square().sum()).backward()
        ts.append(time.perf_counter() - t0)
        for v in sd.values():
            v.grad = None
    dt = sum(ts[1:]) / steps
    return dict(batch=batch, threads=threads, s_per_forward_backward=dt, molecules_per_s=batch / dt,
                note='oracle.forward_faithful + loss.backward() under torch.autograd on the host, one grad-enabled forward and backward')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='qm9')
    ap.add_argument('--batch', type=int, default=0)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--cpu', action='store_true', help='also time the host port of the same forward + backward (small batch)')
    ap.add_argument('--train-opt', action='append', default=[], help='jodo_train_set_option=value (A/B runs), e.g. 0=0 1=0: op-by-op forward / backward')
    ap.add_argument('--save-always', action='store_true', help='the no-grad self-conditioning forward keeps its activations too (round-4 behaviour)')
    a = ap.parse_args()
    out = run(a.workload, a.batch, a.steps, a.warmup, options={int(o.split('=')[0]): int(o.split('=')[1]) for o in a.train_opt}, save_always=a.save_always)
    out['train_options'] = a.train_opt; out['save_always'] = a.save_always
    if a.cpu:
        out['cpu'] = cpu_step(a.workload)
        out['gpu_over_cpu_forward_backward'] = (out['batch'] / ((out['forward_ms'] + out['backward_ms']) * 1e-3)) / out['cpu']['molecules_per_s']
    print(json.dumps(out))
