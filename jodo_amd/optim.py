"""Optimiser side of the training step on flat buffers (host mirror of csrc/train_step.hip).

The reference builds `torch.optim.AdamW(params, amsgrad=True, weight_decay=1e-12)` or `torch.optim.Adam` (losses.py:14-26), clips the
gradient against the recent history of its norm (gradient_clipping :29-50, Queue :53-72) and keeps an exponential moving average of
the 351 parameter tensors (models/ema.py).  On the GPU each of those was a pass over a LIST of tensors — list building, per-tensor state
look-ups and a host synchronisation for the norm, 4.6 ms of a 23 ms step at the QM9 batch with the card idle.  Here

  * `flatten_parameters` re-points every parameter at its slice of ONE allocation (values kept; `p.data_ptr()` changes once),
  * `FlatAdam` is a `torch.optim.Optimizer` whose `step()` is one kernel over that allocation (`jodo_adam_step`), with the moment
    buffers flat as well and exposed per parameter (as views) through the usual `optimizer.state[p]` / `state_dict()`,
  * `DeviceGradNormQueue` keeps the clipping history on the device (`jodo_gradnorm_clip`): no host synchronisation in a step.

There is no CPU fallback: `get_optimizer` (jodo_amd/losses.py) hands CPU parameters to torch's own optimisers, and FlatAdam refuses them.
"""
import ctypes

import torch

from . import capi


ALIGN = 4          # floats: every slice starts on a 16-byte boundary (the training GEMM's 16-byte loads, csrc/train_gemm.hip, need that of
                   # weights and gradients; a 3-float bias would otherwise push everything behind it off alignment)


def slice_offsets(numels, align=ALIGN):
    """Start of every slice in a flat buffer (floats) and the buffer's length: slices in order, each start rounded up to `align`; the
    gaps (at most align - 1 floats each) stay zero for ever — zero gradient, zero moments, zero update."""
    offs, at = [], 0
    for n in numels:
        at = (at + align - 1) // align * align
        offs.append(at)
        at += n
    return offs, (at + align - 1) // align * align


def flat_view(tensors):
    """The 1-D tensor over the storage that `tensors` tile in list order with the slice_offsets layout (float32, contiguous) — or None."""
    if not tensors or tensors[0] is None:
        return None
    t0 = tensors[0]
    store = t0.untyped_storage()
    base = store.data_ptr()
    if any(t is None for t in tensors):
        return None
    offs, total = slice_offsets([t.numel() for t in tensors])
    for t, o in zip(tensors, offs):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.untyped_storage().data_ptr() != base or t.storage_offset() != o:
            return None
    if total * 4 != store.nbytes():
        return None
    return torch.empty(0, dtype=torch.float32, device=t0.device).set_(store, 0, (total,))


def carve(flat, shapes, offsets):
    """The slices of `flat` as tensors of the given shapes."""
    out = []
    for shp, o in zip(shapes, offsets):
        n = 1
        for d in shp:
            n *= d
        out.append(flat[o:o + n].view(shp))
    return out


def flatten_parameters(params):
    """Re-point every parameter at its slice of one new allocation (registration order, slice_offsets layout, values copied).  Returns the
    flat tensor; parameters that already have that layout are left where they are."""
    params = list(params)
    flat = flat_view([p.data for p in params])
    if flat is not None:
        return flat
    if any(p.dtype != torch.float32 for p in params):
        raise TypeError("flatten_parameters: float32 parameters only")
    offs, total = slice_offsets([p.numel() for p in params])
    flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
    with torch.no_grad():
        for p, piece in zip(params, carve(flat, [tuple(p.shape) for p in params], offs)):
            piece.copy_(p.data)
            p.data = piece
    return flat


def _ptr_flat(tensors, total):
    """Per-step form of flat_view: EVERY tensor must sit at its slice of one allocation (address compare per tensor: a few hundred
    integer compares, no device work, no synchronisation).  Round 5 compared only the first and the last slice; a middle gradient
    replaced by a separate tensor (a hook, `p.grad = ...`, partial accumulation) then went unnoticed and the kernel read stale bytes
    of the old flat buffer (round-5 advisor finding)."""
    t0 = tensors[0]
    if t0 is None or t0.dtype != torch.float32:
        return None
    store = t0.untyped_storage()
    base = store.data_ptr()
    if store.nbytes() != 4 * total or t0.storage_offset() != 0:
        return None
    at = 0
    for t in tensors:
        if t is None or t.dtype != torch.float32 or t.data_ptr() != base + 4 * at or not t.is_contiguous():
            return None
        at = (at + t.numel() + ALIGN - 1) // ALIGN * ALIGN
    if at != total:
        return None
    return torch.empty(0, dtype=torch.float32, device=t0.device).set_(store, 0, (total,))


def flat_total(params):
    return slice_offsets([p.numel() for p in params])[1]


def flat_parameters(params, total=None):
    """The flat buffer `params` (a list) are slices of, or None (address check of every parameter: see _ptr_flat)."""
    if not params[0].is_cuda:
        return None
    return _ptr_flat([p.data for p in params], flat_total(params) if total is None else total)


def flat_gradient(params, total=None, full_check=False):
    """The flat gradient buffer of `params` (a list) — jodo_amd/train.py hands every gradient out as a slice of one allocation with the
    slice_offsets layout — or None when the gradients are ordinary separate tensors."""
    if total is None:
        total = flat_total(params)
    if not full_check:
        base = _ptr_flat([p.grad for p in params], total)
        if base is not None:
            return base
    return flat_view([p.grad for p in params])


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam (decoupled=False) / AdamW (decoupled=True) on one flat parameter buffer: the single-tensor formulas of
    torch/optim/adam.py / adamw.py, one launch per step.  One parameter group; `param_groups[0]['lr']` may be changed between steps
    (the reference's warm-up does, losses.py:84-86)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, decoupled=True):
        params = list(params)
        if not params or isinstance(params[0], dict):
            raise ValueError("FlatAdam takes one flat list of parameters")
        if not all(p.is_cuda for p in params):
            raise capi.JodoHipError("FlatAdam runs on the GPU only (no CPU fallback): use torch.optim for CPU parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, decoupled=decoupled))
        self._params = params
        self._flat = flatten_parameters(params)
        self._total = self._flat.numel()
        self._m = torch.zeros_like(self._flat)
        self._v = torch.zeros_like(self._flat)
        self._vmax = torch.zeros_like(self._flat) if amsgrad else None
        self._step = 0
        self._step_t = torch.zeros((), dtype=torch.float32)          # one host tensor shared by every state entry, like torch's 'step'
        self._checked = False
        shapes = self._shapes = [tuple(p.shape) for p in params]
        offs = self._offs = slice_offsets([p.numel() for p in params])[0]
        ms, vs = carve(self._m, shapes, offs), carve(self._v, shapes, offs)
        xs = carve(self._vmax, shapes, offs) if amsgrad else None
        for i, p in enumerate(params):
            st = self.state[p]
            st['step'] = self._step_t
            st['exp_avg'], st['exp_avg_sq'] = ms[i], vs[i]
            if amsgrad:
                st['max_exp_avg_sq'] = xs[i]
        L = capi.lib()
        L.jodo_adam_step.argtypes = [ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_double] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                                                                                      ctypes.c_void_p]
        self._L = L

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        if flat_parameters(self._params, self._total) is None:
            raise RuntimeError("FlatAdam: the parameters were moved off the flat buffer (model.to(...) after the optimiser was built?)")
        self._step += 1
        self._step_t.fill_(self._step)
        b1, b2 = g['betas']

        def launch(lo, hi, grad):                                     # floats lo .. hi of the flat buffers, grad = the gradient of that range
            sub = lambda t_: None if t_ is None else t_[lo:hi]
            capi.check(self._L.jodo_adam_step(hi - lo, capi.ptr(self._flat[lo:hi]), capi.ptr(grad), capi.ptr(self._m[lo:hi]), capi.ptr(self._v[lo:hi]),
                                              capi.ptr(sub(self._vmax)), float(g['lr']), float(b1), float(b2), float(g['eps']), float(g['weight_decay']),
                                              self._step, 1 if g['decoupled'] else 0, 1 if g['amsgrad'] else 0, capi.current_stream_ptr()), 'jodo_adam_step')

        have = [p.grad is not None for p in self._params]
        if all(have):
            grad = flat_gradient(self._params, self._total, full_check=not self._checked)
            self._checked = True
            if grad is None:                                         # ordinary separate gradients: gathered into one buffer (one launch)
                grad = torch.zeros_like(self._flat)
                torch._foreach_copy_(carve(grad, self._shapes, self._offs), [p.grad for p in self._params])
            launch(0, self._total, grad)
            touched = self._params
        else:
            # parameters without a gradient are left alone, as torch's optimisers leave them (frozen layers): one launch per run of
            # consecutive parameters that have one (slices start on 16-byte boundaries, so every run does).  All runs share this
            # optimiser's step count (torch counts steps per parameter).
            touched, i, n = [], 0, len(self._params)
            while i < n:
                if not have[i]:
                    i += 1
                    continue
                j = i
                while j < n and have[j]:
                    j += 1
                lo, hi = self._offs[i], self._offs[j - 1] + self._params[j - 1].numel()
                grad = torch.zeros(hi - lo, dtype=torch.float32, device=self._flat.device)
                torch._foreach_copy_(carve(grad, self._shapes[i:j], [o - lo for o in self._offs[i:j]]), [p.grad for p in self._params[i:j]])
                launch(lo, hi, grad)
                touched += self._params[i:j]
                i = j
        # the kernel wrote through raw pointers: tell autograd / the packed-weight cache of the HIP module (models/dgt.py _weights)
        if touched:
            torch.autograd.graph.increment_version(touched)
        return loss

    _HYPER = ('lr', 'betas', 'eps', 'weight_decay', 'amsgrad', 'decoupled')

    def load_state_dict(self, state_dict):
        """Values go INTO the flat moment buffers (torch's loader would replace the per-parameter views by separate tensors).  Accepts
        this class's own state_dict and a torch.optim.Adam / AdamW one of the same parameter list (the reference's checkpoints,
        utils.py:7-20): one parameter group over exactly these parameters, state for all of them or for none; only the
        hyper-parameters this optimiser implements are taken, `maximize` is refused."""
        groups = state_dict['param_groups']
        if len(groups) != 1:
            raise ValueError("FlatAdam.load_state_dict: %d parameter groups, this optimiser has one" % len(groups))
        if len(groups[0].get('params', ())) != len(self._params):
            raise ValueError("FlatAdam.load_state_dict: the saved group holds %d parameters, this optimiser %d" %
                             (len(groups[0].get('params', ())), len(self._params)))
        if groups[0].get('maximize', False):
            raise ValueError("FlatAdam.load_state_dict: maximize=True is not implemented")
        if bool(groups[0].get('amsgrad', False)) != bool(self.param_groups[0]['amsgrad']):
            raise ValueError("FlatAdam.load_state_dict: amsgrad=%s in the file, %s here (the max-moment buffer is allocated at construction)" %
                             (groups[0].get('amsgrad', False), self.param_groups[0]['amsgrad']))
        packed = state_dict['state']
        have = [i in packed for i in range(len(self._params))]
        if any(have) and not all(have):
            raise ValueError("FlatAdam.load_state_dict: state for %d of %d parameters (all or none)" % (sum(have), len(have)))
        if all(have):
            steps = set()
            for i, p in enumerate(self._params):
                src, st = packed[i], self.state[p]
                for k in ('exp_avg', 'exp_avg_sq', 'max_exp_avg_sq'):
                    if k in st:
                        if k not in src:
                            raise ValueError("FlatAdam.load_state_dict: parameter %d has no '%s'" % (i, k))
                        st[k].copy_(src[k])
                steps.add(int(src['step']))
            if len(steps) != 1:
                raise ValueError("FlatAdam.load_state_dict: per-parameter step counts differ (%s); this optimiser keeps one" % sorted(steps)[:4])
            self._step = steps.pop()
        else:
            self._step = 0
            self._m.zero_(); self._v.zero_()
            if self._vmax is not None:
                self._vmax.zero_()
        self._step_t.fill_(self._step)
        for k in self._HYPER:
            if k in groups[0]:
                self.param_groups[0][k] = groups[0][k]


class DeviceGradNormQueue:
    """The reference's Queue of recent gradient norms (losses.py:53-72: newest first, at most 50, mean / std of what is there) kept on the
    device together with the clipping decision that reads it (jodo_gradnorm_clip): a step never waits for the norm."""

    def __init__(self, device, first=3000.0, max_len=50):
        if max_len != 50:
            raise ValueError("the device history holds 50 norms")
        st = torch.zeros(52, dtype=torch.float64)
        st[0], st[50], st[51] = first, 1, 1
        self.state = st.to(device)
        self.coef = torch.ones((), dtype=torch.float32, device=device)
        self.allowed = torch.zeros((), dtype=torch.float32, device=device)
        L = capi.lib()
        L.jodo_gradnorm_clip.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self._L = L

    def clip_(self, flat_grad, max_grad):
        """flat_grad *= min(1, allowed / (|flat_grad| + 1e-6)), allowed = min(1.5 mean + 2 std of the history, max_grad); the history
        gets min(norm, allowed).  Returns the (device) norm before clipping."""
        total = torch.linalg.vector_norm(flat_grad, 2.0)
        capi.check(self._L.jodo_gradnorm_clip(capi.ptr(total), capi.ptr(self.state), float(max_grad), capi.ptr(self.coef), capi.ptr(self.allowed),
                                              capi.current_stream_ptr()), 'jodo_gradnorm_clip')
        flat_grad.mul_(self.coef)
        return total

    def items(self):
        """The history as a host list, newest first (synchronises; tests and logging)."""
        st = self.state.cpu()
        cnt, nxt = int(st[50]), int(st[51])
        return [float(st[(nxt - 1 - i) % 50]) for i in range(cnt)]
