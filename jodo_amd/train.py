"""Python plumbing of the training-side slice: the backward of phase D of a DGT block — edge residual,
LayerNorm2 + modulate, edge FFN (/root/reference/models/mol_gnn.py:313-317) — as HIP kernels behind the C ABI
(jodo_edge_ffn_backward, csrc/train_kernels.hip).  Device tensors in, device tensors out; no CPU fallback.

This is a building block, not a training loop: the forward of the HIP modules still refuses to run under autograd
(models/dgt.py); the slice is exercised by tests/test_train_gpu.py against torch.autograd through the CPU oracle, whose
gradients are pinned by the reference's own loss.backward() (tests/golden/grad_qm9.npz)."""
import ctypes

import numpy as np
import torch

from . import capi


class EdgeFFNBackward:
    """Packs ff_linear3 / ff_linear4 of one block once (W, W^T in MFMA operand order) and runs the backward on batches of edge rows."""

    def __init__(self, W3, b3, W4, b4, device):
        L = capi.lib()
        self.H, self.De = int(W3.shape[0]), int(W3.shape[1])
        if tuple(W4.shape) != (self.De, self.H) or self.H % self.De:
            raise ValueError("ff_linear3.weight [r De, De] and ff_linear4.weight [De, r De] expected")
        self.r = self.H // self.De
        L.jodo_edge_ffn_pack_size.restype = ctypes.c_size_t
        n = L.jodo_edge_ffn_pack_size(self.De, self.r)
        host = np.zeros(n, dtype=np.float32)
        self.offs = (ctypes.c_int64 * 4)()
        w3 = np.ascontiguousarray(W3.detach().float().cpu().numpy())
        w4 = np.ascontiguousarray(W4.detach().float().cpu().numpy())
        capi.check(L.jodo_edge_ffn_pack(self.De, self.r, w3.ctypes.data_as(ctypes.c_void_p), w4.ctypes.data_as(ctypes.c_void_p),
                                        host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), self.offs), 'jodo_edge_ffn_pack')
        self.packed = torch.from_numpy(host).to(device)
        self.b3 = b3.detach().float().to(device).contiguous()
        self.b4 = b4.detach().float().to(device).contiguous()
        self.device = device

    def __call__(self, e_in, ehat, mods, row_mod, mod_off, d_out):
        """e_in, ehat, d_out [rows, De]; mods [U, 6 De] (edge_time_mlp chunks es1 ec1 eg1 es2 ec2 eg2); row_mod int32 [rows];
        mod_off int32 [U + 1] (rows of a modulation row are contiguous).  Returns dict of gradients (PyTorch layouts)."""
        for name, t in (('e_in', e_in), ('ehat', ehat), ('mods', mods), ('d_out', d_out)):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise TypeError("%s must be a contiguous float32 GPU tensor" % name)
        rows, U = int(e_in.shape[0]), int(mods.shape[0])
        if e_in.shape != (rows, self.De) or ehat.shape != e_in.shape or d_out.shape != e_in.shape or mods.shape != (U, 6 * self.De):
            raise ValueError("shape mismatch")
        if row_mod.dtype != torch.int32 or mod_off.dtype != torch.int32 or row_mod.shape != (rows,) or mod_off.shape != (U + 1,):
            raise ValueError("row_mod int32 [rows], mod_off int32 [U + 1]")
        L = capi.lib()
        L.jodo_edge_ffn_backward_workspace.restype = ctypes.c_size_t
        nbytes = L.jodo_edge_ffn_backward_workspace(rows, self.De, self.r)
        if nbytes == 0:
            raise capi.JodoHipError("jodo_edge_ffn_backward: unsupported shape De=%d mlp_ratio=%d (this slice: 64 / 2)" % (self.De, self.r))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)
        out = dict(d_e_in=new(rows, self.De), d_ehat=new(rows, self.De), d_mods=new(U, 6 * self.De), dW3=new(self.H, self.De),
                   db3=new(self.H), dW4=new(self.De, self.H), db4=new(self.De))
        capi.check(L.jodo_edge_ffn_backward(
            rows, self.De, self.r, U, capi.ptr(e_in), capi.ptr(ehat), capi.ptr(row_mod), capi.ptr(mod_off), capi.ptr(mods),
            capi.ptr(self.packed), self.offs, capi.ptr(self.b3), capi.ptr(self.b4), capi.ptr(d_out), capi.ptr(out['d_e_in']),
            capi.ptr(out['d_ehat']), capi.ptr(out['d_mods']), capi.ptr(out['dW3']), capi.ptr(out['db3']), capi.ptr(out['dW4']),
            capi.ptr(out['db4']), capi.ptr(ws), capi.current_stream_ptr()), 'jodo_edge_ffn_backward')
        self._ws = ws                                   # keep the scratch alive until the stream has consumed it
        return out
