"""Training step around the HIP score network, host Python with the reference's entry points (/root/reference/losses.py):

  get_optimizer             :14-26    Adam / AdamW(amsgrad, weight_decay 1e-12)
  optimization_manager      :75-94    linear warm-up + adaptive gradient clipping (gradient_clipping :29-50, Queue :53-72)
  get_step_fn               :97-125   zero_grad -> loss -> backward -> optimise -> EMA update; evaluation under the EMA weights
  get_sde_graph_loss_fn     :286-385  per-molecule t, forward diffusion of nodes and edges, Kabsch-aligned position target,
                                      50 % self-conditioning double forward, weighted data-prediction loss
  kabsch_batch, get_align_position / get_align_noise  :388-440
  process_edge_batch        :470-498

The arithmetic of the model call — forward with kept activations and loss.backward() — is jodo_train_forward / jodo_train_backward
(csrc/dgt_train.hip) behind models/dgt.py; everything here is the thin tensor algebra the reference also keeps in Python.
"""
import os
import random

import numpy as np
import torch

from .models.utils import (remove_mean_with_mask, sample_combined_position_feature_noise,
                           sample_symmetric_edge_feature_noise)
from .utils import expand_dims, get_self_cond_fn


def get_optimizer(config, params):
    """losses.py:14-27 of the reference: the same optimisers with the same hyper-parameters.  On the GPU the parameters are moved onto
    one flat buffer and the update is one kernel over it (jodo_amd/optim.py FlatAdam -> jodo_adam_step: torch's single-tensor Adam / AdamW
    formulas; clipping + update 3.2 ms -> a few launches per step at QM9, where the 351-tensor multi-tensor pass was bound by its host
    side).  JODO_OPTIM_FLAT=0 keeps torch's fused multi-tensor optimisers (round 4's form), JODO_OPTIM_FUSED=0 torch's defaults; CPU
    parameters always get torch's own."""
    o = config.optim
    params = list(params)
    on_gpu = bool(params) and all(p.is_cuda for p in params)
    if on_gpu and os.environ.get('JODO_OPTIM_FLAT', '1') != '0' and o.optimizer in ('Adam', 'AdamW'):
        from .optim import FlatAdam
        if o.optimizer == 'Adam':
            return FlatAdam(params, lr=o.lr, betas=(o.beta1, 0.999), eps=o.eps, weight_decay=o.weight_decay, amsgrad=False, decoupled=False)
        return FlatAdam(params, lr=o.lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-12, amsgrad=True, decoupled=True)
    fused = on_gpu and os.environ.get('JODO_OPTIM_FUSED', '1') != '0'
    if o.optimizer == 'Adam':
        return torch.optim.Adam(params, lr=o.lr, betas=(o.beta1, 0.999), eps=o.eps, weight_decay=o.weight_decay, fused=fused or None)
    if o.optimizer == 'AdamW':
        return torch.optim.AdamW(params, lr=o.lr, amsgrad=True, weight_decay=1e-12, fused=fused or None)
    raise NotImplementedError(f'Optimizer {o.optimizer} not supported yet!')


class Queue:
    """Recent gradient norms, newest first, bounded."""

    def __init__(self, max_len=50):
        self.items, self.max_len = [], max_len

    def __len__(self):
        return len(self.items)

    def add(self, item):
        self.items.insert(0, item)
        del self.items[self.max_len:]

    def mean(self):
        return np.mean(self.items)

    def std(self):
        return np.std(self.items)


def _flat_gradient(params):
    """The one buffer every p.grad is a slice of (jodo_train_backward writes all gradients into one allocation, jodo_amd/train.py;
    autograd hands the slices to p.grad without copying), as a 1-D tensor over that storage — or None when the gradients are ordinary
    separate tensors, or do not tile their storage in the layout of jodo_amd/optim.py slice_offsets."""
    from .optim import flat_view
    return flat_view([p.grad for p in params])


def _clip_grad_norm(params, max_norm):
    """torch.nn.utils.clip_grad_norm_(params, max_norm, 2.0) — on the flat gradient buffer two launches instead of a multi-tensor
    pass over 351 tensors each for the norms and for the scaling (the same formula: coef = min(1, max_norm / (norm + 1e-6)))."""
    flat = _flat_gradient(params)
    if flat is None:
        return torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm, norm_type=2.0)
    total = torch.linalg.vector_norm(flat, 2.0)
    flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
    return total


def gradient_clipping(params, gradnorm_queue, max_grad, disable_log):
    """max_grad <= 1: plain norm clipping.  Otherwise the allowed norm follows the recent history: 1.5 x mean + 2 x std of
    the queue, capped at max_grad; the (clipped) norm is pushed to the queue."""
    params = list(params)
    if max_grad <= 1.0:
        _clip_grad_norm(params, max_grad)
        return None
    allowed = min(1.5 * gradnorm_queue.mean() + 2 * gradnorm_queue.std(), max_grad)
    grad_norm = _clip_grad_norm(params, float(allowed))
    gradnorm_queue.add(float(min(float(grad_norm), allowed)))
    if not disable_log and float(grad_norm) > 1.5 * gradnorm_queue.mean() + 2 * gradnorm_queue.std():
        print(f'Clipped gradient with value {grad_norm:.1f} while allowed {allowed:.1f}')
    return grad_norm


def optimization_manager(config):
    gradnorm_queue = Queue()
    gradnorm_queue.add(3000)                       # large first entry, flushed by the history
    disable_log = config.optim.disable_grad_log
    device_queue = {}                              # device -> DeviceGradNormQueue (the same history, kept where the gradient is)

    def optimize_fn(optimizer, params, step, lr=config.optim.lr, warmup=config.optim.warmup, grad_clip=config.optim.grad_clip):
        if warmup > 0:
            for g in optimizer.param_groups:
                g['lr'] = lr * np.minimum(step / warmup, 1.0)
        if grad_clip >= 0:
            params = list(params)
            flat = None
            if grad_clip > 1.0 and disable_log and params and params[0].is_cuda and os.environ.get('JODO_OPTIM_FLAT', '1') != '0':
                from .optim import flat_gradient
                flat = flat_gradient(params)
            if flat is not None:
                # nothing is printed, so nothing on the host needs the norm: history, decision and scaling stay on the device
                # (jodo_gradnorm_clip) and the step runs without a host synchronisation
                dq = device_queue.get(flat.device)
                if dq is None:
                    from .optim import DeviceGradNormQueue
                    dq = device_queue[flat.device] = DeviceGradNormQueue(flat.device, first=3000.0)
                dq.clip_(flat, grad_clip)
            else:
                gradient_clipping(params, gradnorm_queue, grad_clip, disable_log)
        optimizer.step()

    optimize_fn.gradnorm_queue, optimize_fn.device_queue = gradnorm_queue, device_queue     # (tests read the histories)
    return optimize_fn


def get_step_fn(noise_scheduler, train, optimize_fn, scaler, config, prop_dist=None):
    if not config.pred_edge or config.only_2D:
        raise NotImplementedError("the HIP path implements the 3D graph model (pred_edge, not only_2D)")
    loss_fn = get_sde_graph_loss_fn(noise_scheduler, train, scaler, config, prop_dist)
    # A step on the HIP module has no host synchronisation of its own (tests/test_train_gpu.py), so the host could queue arbitrarily far
    # ahead of the card (its side of a QM9 step is ~9 ms, the card's ~18).  The step therefore waits for the step BEFORE the one it has
    # just queued: the card always has a whole step of work left, the queue — and the batches, staging buffers and activations it pins —
    # never holds more than two.  JODO_TRAIN_RUNAHEAD = number of steps the host may be ahead (default 1; 0: wait for every step;
    # -1: unbounded).
    runahead = int(os.environ.get('JODO_TRAIN_RUNAHEAD', '1'))
    data_parallel = torch.distributed.is_available() and os.environ.get('JODO_TRAIN_ALLREDUCE', '1') != '0'

    in_flight = []

    def step_fn(state, batch):
        model = state['model']
        if train:
            optimizer = state['optimizer']
            optimizer.zero_grad()
            loss = loss_fn(model, batch)
            loss.backward()
            if data_parallel and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                # one process per GPU, each on its own molecules: one all-reduce over the flat gradient buffer (jodo_amd/dist.py)
                from .dist import allreduce_gradients
                allreduce_gradients(list(model.parameters()), weight=batch['atom_mask'].shape[0])
            optimize_fn(optimizer, model.parameters(), step=state['step'])
            state['step'] += 1
            state['ema'].update(model.parameters())
            if loss.is_cuda and runahead >= 0:
                ev = torch.cuda.Event()
                ev.record()
                in_flight.append(ev)
                while len(in_flight) > runahead:
                    in_flight.pop(0).synchronize()
            return loss
        averaged = state['ema']                      # evaluation runs under the averaged weights, then puts the live ones back
        with torch.no_grad():
            averaged.store(model.parameters())
            averaged.copy_to(model.parameters())
            _weights_changed(model)
            try:
                return loss_fn(model, batch)
            finally:
                averaged.restore(model.parameters())
                _weights_changed(model)

    return step_fn


def _weights_changed(model):
    """EMA copy_to / restore write through `.data` (no version bump): tell the HIP module its packed inference weights are stale."""
    inner = getattr(model, 'module', model)
    if hasattr(inner, 'invalidate_packed_weights'):
        inner.invalidate_packed_weights()


@torch.no_grad()
def kabsch_batch(coords_pred, coords_tar):
    """Per molecule the rotation R minimising |coords_pred - coords_tar R^T| (Kabsch): A = P^T Q = U S V^T, R = U diag(1, 1, sign det A) V^T."""
    A = torch.einsum('...ki,...kj->...ij', coords_pred, coords_tar)
    if A.is_cuda and A.dim() == 3 and A.dtype == torch.float32:
        # one thread per molecule (csrc/train_step.hip k_kabsch: Jacobi in double) instead of torch.linalg.svd, whose error check reads
        # `info` back: the last host synchronisation of a training step
        import ctypes
        from . import capi
        A = A.contiguous()
        R = torch.empty_like(A)
        lib = capi.lib()
        lib.jodo_kabsch_rotations.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        capi.check(lib.jodo_kabsch_rotations(A.shape[0], capi.ptr(A), capi.ptr(R), capi.current_stream_ptr()), 'jodo_kabsch_rotations')
        return R
    U, _, Vt = torch.linalg.svd(A)
    fix = torch.ones(A.size(0), 3, device=A.device, dtype=A.dtype)
    fix[:, -1] = torch.sign(torch.det(A))
    return torch.einsum('...ij,...jk,...kl->...il', U, torch.diag_embed(fix), Vt)


@torch.no_grad()
def get_align_position(z_t, xh):
    rot = kabsch_batch(z_t[:, :, :3], xh[:, :, :3])
    return torch.einsum('...ki,...ji->...jk', rot, xh[:, :, :3])


@torch.no_grad()
def get_align_noise(z_t, xh, alpha_t, sigma_t, noise, node_mask):
    pos_t = z_t[:, :, :3]
    aligned = get_align_position(z_t, xh)
    noise[:, :, :3] = (pos_t - expand_dims(alpha_t, 3) * aligned) / expand_dims(sigma_t, 3)
    return noise


class _PinnedStager:
    """Persistent pinned staging buffers for the host -> device copies of a batch: a ring of three sets (one per batch in flight), a
    set reused only after the copies that last read it have completed.  (torch's `pin_memory()` allocates page-locked memory per call:
    20 ms per QM9 batch, measured.)"""

    def __init__(self, depth=3):
        self.sets = [dict() for _ in range(depth)]
        self.events = [None] * depth
        self.at, self.cur = 0, 0

    def begin(self):
        self.cur = self.at
        self.at = (self.at + 1) % len(self.sets)
        if self.events[self.cur] is not None:
            self.events[self.cur].synchronize()

    def put(self, key, t, dev):
        n = t.numel()
        buf = self.sets[self.cur].get(key)
        if buf is None or buf.numel() < n or buf.dtype != t.dtype:
            buf = self.sets[self.cur][key] = torch.empty(int(n * 1.25) + 16, dtype=t.dtype, pin_memory=True)
        view = buf[:n].view(t.shape)
        # numpy, not Tensor.copy_: above its grain size a CPU torch operation opens an OpenMP region, and the pool's threads then spin on
        # every core for a while — next to the thread that launches kernels and autograd's backward thread (measured: a QM9 step took
        # 33 - 40 ms instead of 19 with three such copies in it)
        np.copyto(view.numpy(), t.detach().contiguous().numpy())
        return view.to(dev, non_blocking=True)

    def end(self):
        ev = torch.cuda.Event()
        ev.record()
        self.events[self.cur] = ev


_STAGERS = {}


def _host_counts(atom_mask, edge_mask):
    """Atom counts of a batch that is still on the host (the reference's loader yields CPU tensors, moved to the device by
    process_edge_batch, losses.py:470-476) — with the layout check the HIP training path otherwise runs on the device and reads back
    (prefix node masks, edge mask = node x node without the diagonal).  None for batches that are already on a device."""
    if atom_mask.device.type != 'cpu' or edge_mask.device.type != 'cpu':
        return None
    # (numpy throughout: single-threaded — see _PinnedStager.put)
    B, N = atom_mask.shape[0], atom_mask.shape[1]
    nm = atom_mask.detach().reshape(B, N).numpy()
    counts = np.rint(nm.sum(1)).astype(np.int32)
    prefix = np.arange(N)[None, :] < counts[:, None]
    if not np.array_equal(prefix.astype(nm.dtype), nm):
        raise ValueError("node_mask must be a prefix mask (real atoms first)")
    want = prefix[:, :, None] & prefix[:, None, :] & ~np.eye(N, dtype=bool)[None]
    em = edge_mask.detach().reshape(B, N, N).numpy()
    if not np.array_equal(want.astype(em.dtype), em):
        raise ValueError("edge_mask must be node_mask x node_mask with the diagonal removed")
    return counts


@torch.no_grad()
def process_edge_batch(batch, device, include_charges, scaler, prop_norm):
    """Batch dict of the data loader -> (xh [B,N,3+nd], edge_x [B,N,N,ch], node_mask [B,N,1], edge_mask, context): positions
    centred per molecule, everything through the training scaler, conditioning properties standardised by their (mean, mad)."""
    dev = torch.device(device)
    stage = None
    if dev.type == 'cuda' and any(torch.is_tensor(v) and v.device.type == 'cpu' for v in batch.values()):
        stage = _STAGERS.setdefault(str(dev), _PinnedStager())
        stage.begin()

    def on(key):
        t = batch[key]
        if stage is not None and t.device.type == 'cpu':
            # the loader's CPU tensor goes through a pinned staging buffer and an asynchronous copy: `.to(device)` of pageable memory
            # waits for everything queued on the stream (the previous step's backward and update) before it even starts
            return stage.put(key, t, dev)
        return t.to(dev)

    node_mask, edge_mask = on('atom_mask').unsqueeze(2), on('edge_mask')
    counts = _host_counts(batch['atom_mask'], batch['edge_mask'])
    if counts is not None:
        node_mask._jodo_counts = counts                      # read by the HIP module's training path (models/dgt.py _train_engine)
    charges = on('formal_charges') if include_charges else torch.zeros(0, device=dev)
    centred = remove_mean_with_mask(on('positions'), node_mask)
    pos, atoms, charges, edges = scaler(centred, on('atom_one_hot'), charges, node_mask, on('edge_one_hot'), edge_mask)
    context = None
    if 'context' in batch:
        context = on('context')
        for col, stats in enumerate(prop_norm.values()):
            context[:, col] = (context[:, col] - stats['mean']) / stats['mad']
    if stage is not None:
        stage.end()
    return torch.cat([pos, atoms, charges], dim=2), edges, node_mask, edge_mask, context


def get_sde_graph_loss_fn(noise_scheduler, train, scaler, config, prop_norm=None):
    """loss_fn(model, batch) -> scalar: node, position and edge data-prediction loss of one batch at per-molecule random times."""
    m = config.model
    device, include_charges, reduce_mean = config.device, m.include_fc_charge, config.training.reduce_mean
    noise_align, pred_data, self_cond = m.noise_align, m.pred_data, m.self_cond
    w_pos, w_atom, w_edge = (float(w) for w in m.loss_weights.split(','))
    cond_process_fn = get_self_cond_fn(config) if self_cond else None

    def loss_fn(model, batch):
        model.train() if train else model.eval()
        xh, edge_x, node_mask, edge_mask, context = process_edge_batch(batch, device, include_charges, scaler, prop_norm)
        B = xh.shape[0]
        n_nodes = node_mask.squeeze(-1).sum(-1)
        t = torch.rand(B, device=xh.device) * (1. - 1e-5) + 1e-5
        alpha_t, sigma_t = noise_scheduler.marginal_prob(t)
        noise = sample_combined_position_feature_noise(B, xh.shape[1], xh.shape[2] - 3, node_mask)
        edge_noise = sample_symmetric_edge_feature_noise(B, edge_x.shape[1], edge_x.shape[-1], edge_mask)
        z_t = expand_dims(alpha_t, 3) * xh + expand_dims(sigma_t, 3) * noise
        edge_z_t = expand_dims(alpha_t, 4) * edge_x + expand_dims(sigma_t, 4) * edge_noise
        if noise_align:
            if pred_data:
                align_pos = get_align_position(z_t, xh)
            else:
                noise = get_align_noise(z_t, xh, alpha_t, sigma_t, noise, node_mask)
        else:
            align_pos = xh[:, :, :3]
        noise_level = torch.log(alpha_t ** 2 / sigma_t ** 2)
        kw = dict(edge_x=edge_z_t, noise_level=noise_level, context=context)
        if self_cond:
            assert pred_data
            cond_x = cond_edge_x = None
            if random.random() < 0.5:
                with torch.no_grad():
                    cond_x, cond_edge_x = model(t, z_t, node_mask, edge_mask, cond_x=None, cond_edge_x=None, **kw)
                    cond_x, cond_edge_x = cond_process_fn(cond_x.detach(), cond_edge_x.detach())
            pred, edge_pred = model(t, z_t, node_mask, edge_mask, cond_x=cond_x, cond_edge_x=cond_edge_x, **kw)
        else:
            pred, edge_pred = model(t, z_t, node_mask, edge_mask, **kw)
        if pred_data:
            tar_pos, tar_atom, tar_edge = align_pos, xh[:, :, 3:], edge_x
        else:
            tar_pos, tar_atom, tar_edge = noise[:, :, :3], noise[:, :, 3:], edge_noise
        l_pos = torch.square(pred[:, :, :3] - tar_pos).mean(-1).sum(-1)
        l_atom = torch.square(pred[:, :, 3:] - tar_atom).mean(-1).sum(-1)
        l_edge = torch.square(tar_edge - edge_pred).mean(-1).reshape(B, -1).sum(-1)
        if reduce_mean:
            l_pos, l_atom = l_pos / n_nodes, l_atom / n_nodes
            l_edge = l_edge / (edge_mask.reshape(B, -1).sum(-1) + 1e-8)
        losses = w_pos * l_pos + w_atom * l_atom + w_edge * l_edge
        if pred_data:
            losses = torch.sqrt(alpha_t / sigma_t) * losses
        return losses.mean()

    return loss_fn
