"""Python plumbing for the caller-side HIP kernels (csrc/sampler_kernels.hip): the fused ancestral update
and the fused decode.  Device tensors in, device tensors out; raises (no CPU fallback) on CPU tensors —
the callers in sampling.py pick the framework path themselves when they run on the CPU (host-logic tests).
"""
import ctypes

import torch

from . import capi
from .utils import _norm_factors


def _f32c(t, name):
    if not t.is_cuda or t.dtype != torch.float32:
        raise TypeError("%s must be a float32 GPU tensor" % name)
    return t if t.is_contiguous() else t.contiguous()


def n_nodes_from_mask(node_mask):
    """int32 [B] atom counts on the mask's device (masks are prefix masks, sampling.py:193-201)."""
    B = node_mask.shape[0]
    return node_mask.reshape(B, -1).sum(1).round().to(torch.int32).contiguous()


class StepBuffers:
    """Ping-pong state buffers of one sampling round (allocated once, reused every step)."""

    def __init__(self, x, edge_x):
        # explicit row-major buffers: empty_like would inherit the permuted strides of the reference's
        # edge-noise expression (z.permute(0, 2, 3, 1) * mask), which the kernels do not follow
        new = lambda t: torch.empty(t.shape, dtype=torch.float32, device=t.device)
        self.x = [new(x), new(x)]
        self.e = [new(edge_x), new(edge_x)]
        self.x_mean = new(x)
        self.e_mean = new(edge_x)
        self.cur = 0


def mix64(*words):
    """splitmix64 finaliser chained over integer words -> one 64-bit value: distinct (seed, rank, round) triples give
    unrelated keys whatever their sizes (no shifting, no truncation: seeds above 2^32 and ranks above 2^16 stay distinct)."""
    M = 0xFFFFFFFFFFFFFFFF
    z = 0x9E3779B97F4A7C15
    for w in words:
        w = int(w)
        while True:                                           # every 64-bit limb of the word, at least one
            z = (z + (w & M) + 0x9E3779B97F4A7C15) & M
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
            z ^= z >> 31
            w >>= 64
            if w <= 0:
                break
    return z


class DeviceNoise:
    """In-kernel normal draws (jodo_sampler_step_rng / jodo_dpm_update_rng, include/jodo_hip.h): Philox4x32-10 keyed by a
    64-bit seed; the draw index counts the updates of a round.  `for_rank` hashes (seed, rank, round) into the key
    (mix64), so the streams of different triples are unrelated."""

    def __init__(self, seed, draw=0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.draw = int(draw)

    @classmethod
    def for_rank(cls, seed, rank=0, round_index=0):
        return cls(mix64(seed, rank, round_index))

    def next_draw(self):
        d = self.draw
        self.draw += 1
        return d


def split_replayed_noise(node_noise, edge_noise):
    """Recorded reference noise (node [B,N,3+nd] masked + CoM-free, edge [B,N,N,ch] symmetric + masked) as the RAW draws the
    kernels take: masking, centre-of-mass removal and lower-triangle mirroring are idempotent on them."""
    return (node_noise[:, :, :3].contiguous(), node_noise[:, :, 3:].contiguous(),
            edge_noise.permute(0, 3, 1, 2).contiguous())


def sampler_step(bufs, n_nodes_dev, c_x, c_pred, sigma, x, edge_x, pred, edge_pred, eps_pos=None, eps_feat=None, eps_edge=None,
                 rng=None):
    """x_mean = c_x x + c_pred pred; x_next = x_mean + sigma * eps (both tensors), eps from RAW normal draws
    in the reference's shapes (see include/jodo_hip.h) or, with rng = DeviceNoise, drawn inside the kernel.
    Returns (x_next, edge_next, x_mean, edge_mean); the returned tensors live in `bufs` and stay valid until the
    call after next."""
    B, N, F = x.shape
    ch = edge_x.shape[-1]
    x, edge_x = _f32c(x, 'x'), _f32c(edge_x, 'edge_x')
    pred, edge_pred = _f32c(pred, 'pred'), _f32c(edge_pred, 'edge_pred')
    nxt = bufs.cur ^ 1
    xn, en = bufs.x[nxt], bufs.e[nxt]
    if xn.data_ptr() == x.data_ptr() or en.data_ptr() == edge_x.data_ptr():
        raise RuntimeError("sampler_step: output buffer aliases the input state")
    cf = lambda v: ctypes.c_float(float(v))
    if rng is not None:
        capi.check(capi.lib().jodo_sampler_step_rng(
            B, N, F, ch, capi.ptr(n_nodes_dev), cf(c_x), cf(c_pred), cf(sigma), None, None, ctypes.c_uint64(rng.seed),
            ctypes.c_uint32(rng.next_draw()), capi.ptr(x), capi.ptr(edge_x), capi.ptr(pred), capi.ptr(edge_pred), capi.ptr(xn),
            capi.ptr(en), capi.ptr(bufs.x_mean), capi.ptr(bufs.e_mean), capi.current_stream_ptr()), 'jodo_sampler_step_rng')
    else:
        eps_pos, eps_feat, eps_edge = _f32c(eps_pos, 'eps_pos'), _f32c(eps_feat, 'eps_feat'), _f32c(eps_edge, 'eps_edge')
        if eps_pos.shape != (B, N, 3) or eps_feat.shape != (B, N, F - 3) or eps_edge.shape != (B, ch, N, N):
            raise ValueError("noise draws must have the reference's shapes [B,N,3], [B,N,nd], [B,ch,N,N]")
        capi.check(capi.lib().jodo_sampler_step(
            B, N, F, ch, capi.ptr(n_nodes_dev), cf(c_x), cf(c_pred), cf(sigma), capi.ptr(x), capi.ptr(edge_x), capi.ptr(pred),
            capi.ptr(edge_pred), capi.ptr(eps_pos), capi.ptr(eps_feat), capi.ptr(eps_edge), capi.ptr(xn), capi.ptr(en),
            capi.ptr(bufs.x_mean), capi.ptr(bufs.e_mean), capi.current_stream_ptr()), 'jodo_sampler_step')
    bufs.cur = nxt
    return xn, en, bufs.x_mean, bufs.e_mean


def decode(config, xh, edge_x, n_nodes_dev):
    """post_process + inverse scaling on the device.  Returns compact device tensors
    (pos f32 [B,N,3], atom_type u8 [B,N], fc i8 [B,N], edge_type u8 [B,N,N])."""
    xh, edge_x = _f32c(xh, 'xh'), _f32c(edge_x, 'edge_x')
    B, N, F = xh.shape
    atom_types = int(config.data.atom_types)
    include_fc = int(bool(config.model.include_fc_charge))
    if F != 3 + atom_types + include_fc:
        raise ValueError("xh has %d features, config says %d" % (F, 3 + atom_types + include_fc))
    nf = _norm_factors(config)
    edge_norm = nf[3] if len(nf) > 3 else 1
    dev = xh.device
    pos = torch.empty(B, N, 3, device=dev)
    at = torch.empty(B, N, dtype=torch.uint8, device=dev)
    fc = torch.empty(B, N, dtype=torch.int8, device=dev)
    et = torch.empty(B, N, N, dtype=torch.uint8, device=dev)
    capi.check(capi.lib().jodo_decode(
        B, N, atom_types, include_fc, int(edge_x.shape[-1]), int(bool(config.data.compress_edge)),
        int(bool(config.data.centered)), ctypes.c_float(float(nf[0])), ctypes.c_float(float(nf[1])),
        ctypes.c_float(float(nf[2])), ctypes.c_float(float(edge_norm)), capi.ptr(n_nodes_dev), capi.ptr(xh),
        capi.ptr(edge_x), capi.ptr(pos), capi.ptr(at), capi.ptr(fc), capi.ptr(et), capi.current_stream_ptr()),
        'jodo_decode')
    return pos, at, fc, et


def mols_from_decoded(pos, at, fc, et, n_nodes):
    """One device->host copy per tensor, then per-molecule views in the reference's tuple format
    (pos[n,3] f32, atom_type[n] i64, edge_type[n,n] f32, fc[n] i64) — sampling.py:12-32."""
    pos, at, fc, et = pos.cpu(), at.cpu().long(), fc.cpu().long(), et.cpu().float()
    mols = []
    for i, n in enumerate(n_nodes):
        n = int(n)
        mols.append((pos[i, :n], at[i, :n], et[i, :n, :n], fc[i, :n]))
    return mols


class _DpmBuffers:
    """Output buffers of the fused solver updates: a small ring, so that an update never writes a tensor that the
    solver still holds (state at the start of the outer step, intermediate states, the previous output)."""

    def __init__(self, x, edge_x, depth=4):
        new = lambda t: torch.empty(t.shape, dtype=torch.float32, device=t.device)
        self.x = [new(x) for _ in range(depth)]
        self.e = [new(edge_x) for _ in range(depth)]
        self.eps = torch.empty(x.shape[0], x.shape[1], 3, dtype=torch.float32, device=x.device)
        self.cur = 0


def dpm_update(solver, coef, x_pos, x_base, edge_base, P, DA, DB, PP, n_nodes_dev, eps=None, rng=None):
    """jodo_dpm_update (include/jodo_hip.h): coef = [cx, cp, sigma, a, b, c, c2, 0]; P / DA / DB / PP are
    (node prediction, edge prediction) pairs; n_nodes_dev = int32 [B] atom counts of THIS round (the solver computes them
    once per `sampling` call).  Position noise: `eps` (a replayed draw [B,N,3]; masking / CoM removal are idempotent),
    in-kernel draws with rng = DeviceNoise, otherwise a raw N(0,1) [B,N,3] draw (the reference's shape and draw,
    mix_dpm_solver.py:56); nothing is drawn when sigma == 0 (last update of a round).  Returns (x_out, edge_out)."""
    B, N, F = x_base.shape
    ch = edge_base.shape[-1]
    if n_nodes_dev.shape[0] != B or n_nodes_dev.device != x_base.device:
        raise ValueError("dpm_update: n_nodes_dev does not belong to this batch")
    bufs = getattr(solver, '_dpm_bufs', None)
    if bufs is None or bufs.x[0].shape != x_base.shape or bufs.e[0].shape != edge_base.shape or bufs.x[0].device != x_base.device:
        bufs = solver._dpm_bufs = _DpmBuffers(x_base, edge_base)
    live = {t.data_ptr() for t in (x_pos, x_base, edge_base, P[0], P[1], DA[0], DA[1], DB[0], DB[1], PP[0])}
    for _ in range(len(bufs.x)):
        bufs.cur = (bufs.cur + 1) % len(bufs.x)
        if bufs.x[bufs.cur].data_ptr() not in live and bufs.e[bufs.cur].data_ptr() not in live:
            break
    else:
        raise RuntimeError("dpm_update: no free output buffer")
    xo, eo = bufs.x[bufs.cur], bufs.e[bufs.cur]
    if rng is None:
        if eps is not None:
            bufs.eps.copy_(eps)
        elif coef[2] != 0.0:
            bufs.eps.normal_()
    c8 = (ctypes.c_float * 8)(*coef)
    # contiguous fp32 views are bound to names until the launch is enqueued: a temporary freed earlier could be handed
    # out again by the caching allocator for the next temporary and be overwritten before the kernel reads it
    t = [_f32c(v, n) for v, n in ((x_pos, 'x_pos'), (x_base, 'x_base'), (edge_base, 'edge_base'), (P[0], 'P'), (P[1], 'eP'),
                                  (DA[0], 'DA'), (DA[1], 'eDA'), (DB[0], 'DB'), (DB[1], 'eDB'), (PP[0], 'PP'))]
    if rng is not None:
        draw = rng.next_draw() if coef[2] != 0.0 else 0
        capi.check(capi.lib().jodo_dpm_update_rng(
            B, N, F, ch, capi.ptr(n_nodes_dev), c8, None, None, 0, 0, ctypes.c_uint64(rng.seed), ctypes.c_uint32(draw),
            ctypes.c_uint32(0), *[capi.ptr(v) for v in t], capi.ptr(xo), capi.ptr(eo), capi.current_stream_ptr()),
            'jodo_dpm_update_rng')
    else:
        capi.check(capi.lib().jodo_dpm_update(
            B, N, F, ch, capi.ptr(n_nodes_dev), c8, None, None, 0, 0, *[capi.ptr(v) for v in t],
            capi.ptr(bufs.eps), capi.ptr(xo), capi.ptr(eo), capi.current_stream_ptr()), 'jodo_dpm_update')
    del t
    return xo, eo
