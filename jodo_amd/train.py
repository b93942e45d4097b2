"""Python plumbing of the training side behind the C ABI; device tensors in, device tensors out, no CPU
fallback.

  TrainEngine / dgt_autograd   the whole score network under autograd: jodo_train_forward keeps the activations,
                               jodo_train_backward returns the gradient of every parameter (csrc/dgt_train.hip, train_ops.h,
                               train_gemm.hip) — what loss.backward() does in /root/reference/losses.py:286-385.  models/dgt.py
                               routes a grad-enabled forward here, so `loss.backward()` works on the registered module.

Checked by tests/test_train_gpu.py against torch.autograd through the CPU oracle, whose gradients are pinned by the reference's
own loss.backward() (tests/golden/grad_qm9.npz)."""
import ctypes

import numpy as np
import torch

from . import capi


class TrainEngine:
    """One batch shape (atom counts) of the training path: owns the jodo_train handle, its device tables and the workspace that
    carries the activations from forward to backward.  `lib` / `stream_ptr` exist for the build container's CPU suite, which
    drives the host-emulation build of the same sources (tests/emul/) with host tensors; the product path (models/dgt.py) never
    passes them and always runs libjodo_hip.so on the current HIP stream."""

    @staticmethod
    def named_table(named_shapes):
        """The jodo_tensor array (names + shapes, no data) that jodo_train_create reads: built once per module."""
        n = len(named_shapes)
        arr = (capi.JodoTensor * n)()
        keep = []
        for i, (name, shape) in enumerate(named_shapes):
            shp = (ctypes.c_int64 * max(len(shape), 1))(*shape)
            nm = name.encode()
            keep.append((shp, nm))
            arr[i] = capi.JodoTensor(nm, None, shp, len(shape))
        return dict(arr=arr, keep=keep, n=n)

    @staticmethod
    def new_pool(slots=2):
        """Activation workspaces shared by the engines of one module.  A forward takes a slot that no pending backward needs; with
        `slots` = 2 two grad-enabled forwards may precede their backwards (gradient accumulation over two micro-batches, two loss
        terms — the reference's module allows any number).  When every slot is waiting for a backward the oldest is
        reused and ITS backward raises."""
        return {'slots': [{'buf': None, 'stamp': 0, 'live': False} for _ in range(slots)], 'stamp': 0}

    def __init__(self, cfg_struct, n_nodes, N, named_shapes, device, lib=None, stream_ptr=None, pool=None, options=None):
        self.L = lib if lib is not None else capi.lib()
        self._check = capi.check if lib is None else self._check_foreign
        self._stream = stream_ptr if stream_ptr is not None else capi.current_stream_ptr
        self.device = device
        named = named_shapes if isinstance(named_shapes, dict) else self.named_table(named_shapes)
        self._named = named                                      # keeps the ctypes name / shape storage alive
        self.n_params = named['n']
        n_host = np.ascontiguousarray(np.asarray(n_nodes, dtype=np.int32))
        self.B, self.N = int(n_host.shape[0]), int(N)
        self.handle = ctypes.c_void_p()
        self._check(self.L.jodo_train_create(ctypes.byref(cfg_struct), self.B, self.N, n_host.ctypes.data_as(ctypes.c_void_p), named['arr'],
                                             self.n_params, ctypes.byref(self.handle)), 'jodo_train_create')
        self.L.jodo_train_desc_bytes.restype = ctypes.c_size_t
        self.L.jodo_train_workspace_bytes.restype = ctypes.c_size_t
        self.L.jodo_train_desc_bytes.argtypes = [ctypes.c_void_p]
        self.L.jodo_train_workspace_bytes.argtypes = [ctypes.c_void_p]
        for opt, val in (options or {}).items():                 # jodo_train_set_option, e.g. {0: 0} = op-by-op forward (tests)
            self._check(self.L.jodo_train_set_option(self.handle, int(opt), int(val)), 'jodo_train_set_option')
        self.desc = torch.empty(self.L.jodo_train_desc_bytes(self.handle), dtype=torch.uint8, device=device)
        # The activation workspace (2.6 GB at QM9 batch 128) is shared by every engine of a module through `pool`: data loaders
        # produce a new set of atom counts every step, so handles come and go while the pool's buffers, grown to the largest request,
        # serve them all.  A slot's stamp names the forward whose activations it holds.
        self.ws_bytes = int(self.L.jodo_train_workspace_bytes(self.handle))
        self.pool = pool if pool is not None else self.new_pool()
        self._upload_tables()
        self.flags = torch.zeros(8, dtype=torch.int32, device=device)
        self._slot = None

    def _check_foreign(self, code, what=''):
        if code != 0:
            self.L.jodo_last_error.restype = ctypes.c_char_p
            raise capi.JodoHipError("%s failed (%d): %s" % (what, code, (self.L.jodo_last_error() or b'').decode()))

    def __del__(self):
        try:
            self.L.jodo_train_destroy.argtypes = [ctypes.c_void_p]
            self.L.jodo_train_destroy(self.handle)
        except Exception:
            pass

    def _upload_tables(self):
        """Index tables -> device without a stream synchronisation: the handle's host image is staged through one of the pool's pinned
        buffers (a ring of four; a buffer is reused only after the copy that last read it has completed) and copied asynchronously on
        the current stream.  (jodo_train_upload, the plain C entry, synchronises instead: a shuffling loader creates an engine per
        step, and that synchronisation was the last one left in a training step.)"""
        n = int(self.desc.numel())
        if self._stream is not capi.current_stream_ptr:          # a caller-chosen stream: the plain (synchronising) upload on that stream
            self._check(self.L.jodo_train_upload(self.handle, capi.ptr(self.desc), self._stream()), 'jodo_train_upload')
            return
        ring = self.pool.setdefault('pin', {'bufs': [None] * 4, 'events': [None] * 4, 'next': 0})
        i = ring['next']
        ring['next'] = (i + 1) % len(ring['bufs'])
        if ring['events'][i] is not None:
            ring['events'][i].synchronize()
        if ring['bufs'][i] is None or ring['bufs'][i].numel() < n:
            ring['bufs'][i] = torch.empty(int(n * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
        self.L.jodo_train_desc_host.restype = ctypes.c_void_p
        self.L.jodo_train_desc_host.argtypes = [ctypes.c_void_p]
        src = self.L.jodo_train_desc_host(self.handle)
        if not src:
            raise capi.JodoHipError("jodo_train_desc_host returned NULL")
        ctypes.memmove(ring['bufs'][i].data_ptr(), src, n)
        self.desc.copy_(ring['bufs'][i][:n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring['events'][i] = ev

    @staticmethod
    def _ptrs(tensors):
        return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def _take_slot(self, device, keep):
        """A workspace slot for a forward: one that no pending backward needs (the most recently used of those first, so that a
        plain forward / backward loop stays in one buffer), else the oldest.  keep: this forward's backward will need it."""
        pool = self.pool
        free = [s_ for s_ in pool['slots'] if not s_['live']]
        slot = max(free, key=lambda s_: s_['stamp']) if free else min(pool['slots'], key=lambda s_: s_['stamp'])
        if slot['buf'] is None or slot['buf'].numel() < self.ws_bytes or slot['buf'].device != device:
            # grow with headroom: a shuffling loader's batches differ by a few per cent in sum n^2, and re-allocating gigabytes
            # whenever a slightly larger one arrives stalls the step (batch 2 048: 440 ms per step instead of 190 while the
            # maximum was still being found)
            first = slot['buf'] is None
            slot['buf'] = None                                   # release before growing
            slot['buf'] = torch.zeros(self.ws_bytes if first else int(self.ws_bytes * 1.2), dtype=torch.uint8, device=device)
        pool['stamp'] += 1
        slot['stamp'], slot['live'] = pool['stamp'], bool(keep)
        return slot

    def forward(self, params, xh, edge_x, cond_x, cond_edge_x, noise_level, context, dropout_p, seed, keep=False, save_activations=True):
        """params: contiguous float32 tensors in the order of `named_shapes`.  Returns (out_xh, out_edge); the activations stay
        in a workspace slot of the pool (self.stamp names it) — until the next forward, or, with keep=True, until `backward` /
        `release` has been called for that stamp."""
        assert len(params) == self.n_params
        slot = self._take_slot(xh.device, keep)
        self._slot, self.stamp = slot, slot['stamp']
        out_x, out_e = torch.empty_like(xh), torch.empty_like(edge_x)
        if bool(save_activations) != getattr(self, '_saving', True):    # option 2: a forward nobody differentiates skips backward-only stores
            self._check(self.L.jodo_train_set_option(self.handle, 2, 1 if save_activations else 0), 'jodo_train_set_option')
            self._saving = bool(save_activations)
        self._check(self.L.jodo_train_forward(
            self.handle, capi.ptr(self.desc), self._ptrs(params), self.n_params, capi.ptr(xh), capi.ptr(edge_x), capi.ptr(cond_x),
            capi.ptr(cond_edge_x), capi.ptr(noise_level), capi.ptr(context), ctypes.c_float(dropout_p), ctypes.c_uint64(seed),
            capi.ptr(out_x), capi.ptr(out_e), capi.ptr(self.flags), capi.ptr(slot['buf']), self._stream()), 'jodo_train_forward')
        return out_x, out_e

    def release(self, stamp):
        """The activations of forward `stamp` are no longer needed (its backward ran, or never will): its slot may serve the next
        forward.  Called by `backward`, and by the autograd node's guard when a graph is dropped without a backward (an exception
        between forward and backward, a train-mode forward run for logging) — without it the slot stayed live forever and the
        following forward / backward loop allocated, and kept, a second workspace (gigabytes, see above)."""
        release_slot(self.pool, stamp)

    def slot_of(self, stamp):
        for s_ in self.pool['slots']:
            if s_['stamp'] == stamp and s_['buf'] is not None:
                return s_
        return None

    def debug_fetch(self, what, layer):
        """tests: a kept activation of the last forward (jodo_train_debug_locate: 0 = hhat [Nn, D], 1 = alpha [R, H] of block `layer`)."""
        off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
        self.L.jodo_train_debug_locate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self._check(self.L.jodo_train_debug_locate(self.handle, int(what), int(layer), ctypes.byref(off), ctypes.byref(cnt)), 'jodo_train_debug_locate')
        buf = self._slot['buf']
        return buf[off.value:off.value + 4 * cnt.value].view(torch.float32).clone()

    def backward(self, params, noise_level, d_out_x, d_out_e, dropout_p, seed, stamp=None):
        """Gradients of every parameter for the forward `stamp` (default: this engine's last forward)."""
        slot = self.slot_of(self.stamp if stamp is None else stamp)
        if slot is None:
            raise RuntimeError("the activations of this forward were overwritten by later training-path forwards of the same module "
                               "(%d activation workspaces per module, INTEGRATION.md §5); call backward earlier" % len(self.pool['slots']))
        # one allocation for all gradients (the library then zeroes them with a single fill), every slice on a 16-byte boundary — the
        # layout of jodo_amd/optim.py slice_offsets, which the flat optimiser's parameter buffer has too
        lay = self.__dict__.get('_grad_layout')
        if lay is None:
            from .optim import slice_offsets
            offs, total = slice_offsets([p.numel() for p in params])
            lay = self._grad_layout = ([tuple(p.shape) for p in params], offs, total)
        from .optim import carve
        flat = torch.empty(lay[2], dtype=torch.float32, device=params[0].device)
        tail = lay[1][-1] + params[-1].numel()
        if tail != lay[2]:                                       # padding behind the LAST slice: no gradient buffer follows it, so the
            flat[tail:].zero_()                                  # library's fill does not reach it, and the norm of `flat` reads it
        grads = carve(flat, lay[0], lay[1])
        self._check(self.L.jodo_train_backward(
            self.handle, capi.ptr(self.desc), self._ptrs(params), self._ptrs(grads), self.n_params, capi.ptr(noise_level),
            capi.ptr(d_out_x), capi.ptr(d_out_e), ctypes.c_float(dropout_p), ctypes.c_uint64(seed), capi.ptr(slot['buf']),
            self._stream()), 'jodo_train_backward')
        slot['live'] = False                                     # = release(stamp); a second backward over the same activations still works until a forward takes the slot
        return grads


def release_slot(pool, stamp):
    for s_ in pool['slots']:
        if s_['stamp'] == stamp:
            s_['live'] = False


class _SlotGuard:
    """Lives on the autograd context of one training-path forward; when the context dies (backward done and graph freed, or the
    graph dropped without a backward) the forward's workspace slot is released."""

    def __init__(self, pool, stamp):
        self.pool, self.stamp = pool, stamp

    def __del__(self):
        try:
            release_slot(self.pool, self.stamp)
        except Exception:
            pass


class _DGTTrainFn(torch.autograd.Function):
    """The score network as one autograd node: inputs need no gradient (the self-conditioning inputs are detached,
    losses.py:339); the parameters get theirs from jodo_train_backward."""

    @staticmethod
    def forward(ctx, engine, dropout_p, seed, xh, edge_x, cond_x, cond_edge_x, noise_level, context, *params):
        ps = [p.detach().contiguous() for p in params]
        out_x, out_e = engine.forward(ps, xh, edge_x, cond_x, cond_edge_x, noise_level, context, dropout_p, seed, keep=True)
        ctx.engine, ctx.dropout_p, ctx.seed, ctx.ps, ctx.nl = engine, dropout_p, seed, ps, noise_level
        ctx.stamp = engine.stamp
        ctx.slot_guard = _SlotGuard(engine.pool, engine.stamp)       # releases the slot when this context is dropped, backward or not
        return out_x, out_e

    @staticmethod
    def backward(ctx, d_out_x, d_out_e):
        grads = ctx.engine.backward(ctx.ps, ctx.nl, d_out_x.contiguous(), d_out_e.contiguous(), ctx.dropout_p, ctx.seed, stamp=ctx.stamp)
        return (None,) * 9 + tuple(grads)


def dgt_autograd(engine, dropout_p, seed, xh, edge_x, cond_x, cond_edge_x, noise_level, context, params):
    return _DGTTrainFn.apply(engine, dropout_p, seed, xh, edge_x, cond_x, cond_edge_x, noise_level, context, *params)
