"""MI355X-native Diffusion Graph Transformer score networks, registered under the reference's names.

`DGT_concat` / `Cond_DGT_concat` are drop-in replacements for the classes of the same registry name
in /root/reference/models/mol_gnn.py (:410-594, :597-794):

  * same constructor argument (`config`) and the same config keys are read;
  * same parameter names, shapes and registration order (state_dict keys, EMA shadow-list order,
    `strict=True` checkpoint loading) — the parameter tree below is built from stock torch containers
    in the reference's construction order, so even `torch.manual_seed(s); Model(config)` yields the
    same initial weights;
  * same call: `model(t, xh, node_mask, edge_mask, context=None, edge_x=..., cond_x=..., cond_edge_x=...,
    noise_level=...)` -> `(xh_pred [B,N,3+nd], edge_pred [B,N,N,ch])`, inputs untouched.

The arithmetic is NOT done by these torch modules: `forward` packs the weights once (MFMA operand
order, csrc/dgt_pack.cpp), builds a plan per batch of atom counts, and calls
`jodo_dgt_forward` in libjodo_hip.so on the current HIP stream.  There is no CPU or eager fallback:
on a non-GPU tensor, with gradients enabled, or with an unsupported config, forward raises.
"""
import ctypes

import numpy as np
import torch
from torch import nn

from .. import capi
from .dims import ModelDims
from . import utils


# ---------------------------------------------------------------------------------------------
# parameter containers (hold weights only; mirror the reference's module tree)
# ---------------------------------------------------------------------------------------------
def _time_seq(time_dim, out_dim):
    return nn.Sequential(nn.SiLU(), nn.Linear(time_dim, out_dim))


class _SinusoidParams(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weights = nn.Parameter(torch.randn(dim // 2))


class _GaussianParams(nn.Module):
    def __init__(self, K, time_dim):
        super().__init__()
        self.means = nn.Embedding(1, K - 1)
        self.stds = nn.Embedding(1, K - 1)
        self.time_mlp = _time_seq(time_dim, 2)
        nn.init.uniform_(self.means.weight, 0, 3)
        nn.init.uniform_(self.stds.weight, 0, 3)


class _CoorsNormParams(nn.Module):
    def __init__(self, scale_init):
        super().__init__()
        self.scale = nn.Parameter(torch.zeros(1).fill_(scale_init))


class _AttnParams(nn.Module):
    def __init__(self, node_dim, head_ch, n_extra, n_heads, edge_dim):
        super().__init__()
        sub_heads = n_heads - n_extra
        sub_ch = (n_heads * head_ch) // sub_heads
        self.lin_key = nn.Linear(node_dim, sub_heads * sub_ch)
        self.lin_query = nn.Linear(node_dim, sub_heads * sub_ch)
        self.lin_value = nn.Linear(node_dim, n_heads * head_ch)
        self.lin_edge0 = nn.Linear(edge_dim, sub_heads * sub_ch, bias=False)
        self.lin_edge1 = nn.Linear(edge_dim, n_heads * head_ch, bias=False)
        for m in (self.lin_key, self.lin_query, self.lin_value, self.lin_edge0, self.lin_edge1):
            m.reset_parameters()          # the reference layer re-initialises after construction


class _EquiUpdateParams(nn.Module):
    def __init__(self, hidden, edge_dim, dist_dim, time_dim, n_extra):
        super().__init__()
        self.coord_norm = _CoorsNormParams(1e-2)
        self.time_mlp = _time_seq(time_dim, hidden * 2)
        self.input_lin = nn.Linear(hidden * 2 + edge_dim + dist_dim, hidden)
        self.coord_mlp = nn.Sequential(nn.Linear(hidden, hidden), nn.SiLU(), nn.Linear(hidden, 1 + n_extra, bias=False))


class _BlockParams(nn.Module):
    def __init__(self, node_dim, edge_dim, time_dim, n_extra, n_heads, mlp_ratio):
        super().__init__()
        self.edge_emb = nn.Linear(edge_dim * 2, edge_dim)
        self.node2edge_lin = nn.Linear(node_dim, edge_dim)
        self.attn_mpnn = _AttnParams(node_dim, node_dim // n_heads, n_extra, n_heads, edge_dim)
        self.ff_linear1 = nn.Linear(node_dim, node_dim * mlp_ratio)
        self.ff_linear2 = nn.Linear(node_dim * mlp_ratio, node_dim)
        self.ff_linear3 = nn.Linear(edge_dim, edge_dim * mlp_ratio)
        self.ff_linear4 = nn.Linear(edge_dim * mlp_ratio, edge_dim)
        self.equi_update = _EquiUpdateParams(node_dim, edge_dim, edge_dim, time_dim, n_extra)
        self.node_time_mlp = _time_seq(time_dim, node_dim * 6)
        self.edge_time_mlp = _time_seq(time_dim, edge_dim * 6)
        self.dist_layer = _GaussianParams(edge_dim, time_dim)


def _mlp3(i, h, m, o):
    return nn.Sequential(nn.Linear(i, h), nn.SiLU(), nn.Linear(h, m), nn.SiLU(), nn.Linear(m, o))


def _drop_packed_after_load(module, incompatible_keys):
    """load_state_dict post hook (module-level so that the module stays picklable)."""
    module.invalidate_packed_weights()


_UNSUPPORTED = (('dist_gbf', True), ('cond_time', True), ('pred_data', True), ('gbf_name', 'CondGaussianLayer'),
                ('CoM', True), ('softmax_inf', True))


class _DGTBase(nn.Module):
    conditional = False

    def __init__(self, config):
        super().__init__()
        m = config.model
        for key, want in _UNSUPPORTED:
            if getattr(m, key) != want:
                raise NotImplementedError(
                    "config.model.%s=%r: the HIP path implements %r only (the JODO configs' setting)"
                    % (key, getattr(m, key), want))
        if not self.conditional and getattr(m, 'trans_name', 'TransMixLayer') != 'TransMixLayer':
            raise NotImplementedError("config.model.trans_name=%r: only TransMixLayer is implemented" % m.trans_name)
        in_node_dim = config.data.atom_types + int(m.include_fc_charge)
        D, De, L = m.nf, m.nf // 4, m.n_layers
        T = D * 4
        cond_ch = int(m.cond_ch) if self.conditional else 0
        if D not in (128, 256, 384) or m.n_heads != 16 or m.n_extra_heads != 2 or m.mlp_ratio not in (2, 4):
            raise NotImplementedError("HIP kernels are built for nf in {128, 256, 384}, n_heads=16, n_extra_heads=2, "
                                      "mlp_ratio in {2,4} (got nf=%d heads=%d/%d ratio=%d)" %
                                      (D, m.n_heads, m.n_extra_heads, m.mlp_ratio))
        # kernel_layout (not a reference key): 'wide' runs the width-generic kernel set at nf=256 too (tests)
        wide = getattr(m, 'kernel_layout', 'auto') == 'wide'
        self.dims = ModelDims(D, L, m.n_heads, m.n_extra_heads, m.mlp_ratio, in_node_dim, m.edge_ch, cond_ch, wide=wide)
        if self.dims.cep > 32 or self.dims.KEH % 32 or not 3 <= self.dims.KEH // 32 <= 15 or self.dims.KEH // 32 == 14 or L > 16:
            raise NotImplementedError("n_layers=%d at nf=%d: the edge head input (De + n_layers x %d = %d) must be a multiple of 32 in "
                                      "[96, 480] and the per-block edge readout (%d) at most 32 wide" %
                                      (L, D, self.dims.cep, self.dims.KEH, self.dims.ce))
        self.edge_th = float(m.edge_quan_th)
        self.spatial_cut_off = float(m.spatial_cut_off)
        self.n_layers = L
        self.dropout_p = m.dropout           # identity at inference; kept for config parity

        # ---- parameter tree, reference construction/registration order ----
        self.node_emb = nn.Linear(in_node_dim * 2, D)
        self.edge_emb = nn.Linear(m.edge_ch * 2 + De, De)
        self.dist_layer = _GaussianParams(De, T)
        cat_node, cat_edge = (D * 2) // L, (De * 2) // L
        for i in range(L):
            self.add_module("e_block_%d" % i, _BlockParams(D, De, T, m.n_extra_heads, m.n_heads, m.mlp_ratio))
            self.add_module("node_%d" % i, nn.Linear(D, cat_node))
            self.add_module("edge_%d" % i, nn.Linear(De, cat_edge))
        self.node_pred_mlp = _mlp3(cat_node * L + D, D, D // 2, in_node_dim)
        self.edge_type_mlp = _mlp3(cat_edge * L + De, De, De // 2, m.edge_ch - 1)
        self.edge_exist_mlp = _mlp3(cat_edge * L + De, De, De // 2, 1)
        self.time_mlp = nn.Sequential(_SinusoidParams(16), nn.Linear(17, T), nn.GELU(), nn.Linear(T, T))
        if self.conditional:
            self.cond_mlp = nn.Sequential(nn.Linear(1, D), nn.GELU(), nn.Linear(D, D))
            self.cond_lin = nn.Linear(cond_ch * D, T)

        # ---- runtime state (not part of state_dict) ----
        self._packed = None           # (version_key, blob_dev, woff_host ctypes array)
        self._plans = {}              # plan cache keyed by the mask storage (address, shape, version), see _plan
        self._splits = {}             # sub-batch splits of n_streams > 1, keyed likewise
        self.n_streams = 1            # > 1: sub-batches evaluated concurrently on that many HIP streams (_forward_split)
        self._cfg_struct = None
        self.register_load_state_dict_post_hook(_drop_packed_after_load)
        self.last_flags = None        # device int32[8] of the last call (NaN guard etc.)
        self.warn_nan = True
        # OPT-IN (default off): under pinned paths (pin_paths: what the samplers do after a round's first self-conditioned evaluation)
        # the folded pair update runs its projections in the split-bf16 form — three bf16 terms per operand, six bf16 MFMAs per K = 16
        # step, fp32 accumulation: fp32-equivalent arithmetic, not bit-identical to the default (csrc/dgt_kernels_split.h,
        # JODO_OPT_SPLIT_BF16).  nf 256 (pair update + node kernel) and nf 384 (pair update) unconditional models; ignored elsewhere.  The default path and every headline number
        # stay exact fp32.
        self.split_bf16 = False       # True: pair update + node kernel; 'attention': also the attention kernel (experiments: measured slower)
        self._split_tape = None       # (weights key, device uint8 tensor): the split form's static weight tape

    # -- C structs ---------------------------------------------------------------------------
    class _Cfg(ctypes.Structure):
        _fields_ = [('nf', ctypes.c_int32), ('n_layers', ctypes.c_int32), ('n_heads', ctypes.c_int32),
                    ('n_extra', ctypes.c_int32), ('mlp_ratio', ctypes.c_int32), ('in_node_dim', ctypes.c_int32),
                    ('edge_ch', ctypes.c_int32), ('cond_ch', ctypes.c_int32),
                    ('spatial_cut_off', ctypes.c_float), ('edge_quan_th', ctypes.c_float), ('layout', ctypes.c_int32)]

    def _cfg(self):
        if self._cfg_struct is None:
            d = self.dims
            self._cfg_struct = self._Cfg(d.D, d.L, d.H, d.XH, d.r, d.nd, d.ch, d.cond_ch, self.spatial_cut_off,
                                         self.edge_th, 1 if d.wide else 0)
        return self._cfg_struct

    # -- weights -----------------------------------------------------------------------------
    def _weights(self, device):
        # cheap key checked on every call: tensor versions (bumped by optimizer steps, load_state_dict, p.copy_) and
        # storage addresses (model.to(...)).  In-place writes through `.data` (the reference's EMA copy_to / restore,
        # models/ema.py:44-66) bump neither: a content fingerprint is compared whenever a new plan is built (once per
        # sampling round, _plan below), and invalidate_packed_weights() forces a re-pack explicitly.
        key = (str(device),) + tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())
        if self._packed is None or self._packed[0] != key:
            # C-side packer (csrc/dgt_pack.cpp, jodo_dgt_pack_weights): the state_dict goes through the C ABI as named
            # fp32 tensors; tests/py_packing_model.py is an independent Python packer kept for tests (blob equality)
            blob_dev, woff_c, n_woff = capi.pack_weights(self._cfg(), self.state_dict(), device)
            self._packed = (key, blob_dev, woff_c, n_woff)
            self._packed_fingerprint = self._fingerprint()
        return self._packed

    def _fingerprint(self):
        """Content fingerprint of all parameters: one fused norm launch + one reduction, one host sync."""
        ps = [p.detach() for p in self.parameters()]
        norms = torch._foreach_norm(ps)
        w = torch.arange(1, len(norms) + 1, device=norms[0].device, dtype=torch.float64)
        return float((torch.stack([n.double() for n in norms]) * (1.0 + 1e-3 * w)).sum().item())

    def _recheck_weights(self):
        if self._packed is not None and self._fingerprint() != getattr(self, '_packed_fingerprint', None):
            self._packed = None

    # -- plans ---------------------------------------------------------------------------------
    def _plan(self, node_mask, edge_mask, device, validate=True):
        # keyed by the mask's storage address, shape and version counter; the entry keeps the mask alive, so its storage cannot be
        # recycled for a different mask while the plan is cached (data_ptr alone would not be a safe key).  Not by tensor identity:
        # torch.nn.DataParallel — how the reference's create_model wraps the model (models/utils.py:27) — scatters every call's
        # arguments, and with one device that hands the module a fresh VIEW of the same mask on every call (same storage, same
        # version counter); identity would rebuild the plan on each of a round's 1000 calls.
        key = (node_mask.data_ptr(), tuple(node_mask.shape), tuple(node_mask.stride()), str(node_mask.device))
        plan = self._plans.get(key)
        if plan is not None and plan['mask_version'] == node_mask._version:
            return plan
        self._recheck_weights()                          # new batch (= new sampling round): catch `.data` weight updates
        B, N = node_mask.shape[0], node_mask.shape[1]
        nm = node_mask.reshape(B, N)
        n_nodes = nm.sum(1).round().to(torch.int32)
        if validate:
            prefix = (torch.arange(N, device=nm.device).unsqueeze(0) < n_nodes.unsqueeze(1)).to(nm.dtype)
            if not torch.equal(prefix, nm):
                raise ValueError("node_mask must be a prefix mask (real atoms first), as the samplers build it")
            em = edge_mask.reshape(B, N, N)
            want = prefix.unsqueeze(1) * prefix.unsqueeze(2) * (~torch.eye(N, dtype=torch.bool, device=nm.device))
            if not torch.equal(want.to(em.dtype), em):
                raise ValueError("edge_mask must be node_mask x node_mask with the diagonal removed")
        n_host = np.ascontiguousarray(n_nodes.cpu().numpy(), dtype=np.int32)     # one sync per new batch
        L = capi.lib()
        handle = ctypes.c_void_p()
        capi.check(L.jodo_plan_create(ctypes.byref(self._cfg()), B, N, n_host.ctypes.data_as(ctypes.c_void_p),
                                      int(getattr(self, 'max_chunk', 0)) | (int(getattr(self, 'pair_chunk', 0)) << 16)
                                      | (int(getattr(self, 'spair_chunk', 0)) << 24),
                                      ctypes.byref(handle)), 'jodo_plan_create')
        L.jodo_plan_desc_bytes.restype = ctypes.c_size_t
        L.jodo_plan_workspace_bytes.restype = ctypes.c_size_t
        desc = torch.empty(L.jodo_plan_desc_bytes(handle), dtype=torch.uint8, device=device)
        # zero-filled once: rows the kernels never write (diagonal edge rows on the pair path) must stay finite
        ws = torch.zeros(L.jodo_plan_workspace_bytes(handle), dtype=torch.uint8, device=device)
        if getattr(self, 'force_directed', False):       # tests: never use the symmetric pair kernels
            capi.check(L.jodo_debug_set_force_directed(handle, 1), 'jodo_debug_set_force_directed')
        for opt, val in getattr(self, 'plan_options', {}).items():     # {jodo_plan_option: value}, experiments only
            capi.check(L.jodo_plan_set_option(handle, int(opt), int(val)), 'jodo_plan_set_option')
        capi.check(L.jodo_plan_upload(handle, capi.ptr(desc), capi.current_stream_ptr()), 'jodo_plan_upload')
        torch.cuda.current_stream().synchronize()        # host staging buffer lives in the plan; be safe
        plan = dict(handle=handle, desc=desc, ws=ws, n_nodes=n_host, B=B, N=N, mask=node_mask,
                    mask_version=node_mask._version, flags=torch.zeros(8, dtype=torch.int32, device=device))
        stale = self._plans.pop(key, None)
        if stale is not None:
            L.jodo_plan_destroy(stale['handle'])
        if len(self._plans) >= 8:                        # bounded cache
            old = self._plans.pop(next(iter(self._plans)))
            L.jodo_plan_destroy(old['handle'])
        self._plans[key] = plan
        return plan

    def _replicate_for_data_parallel(self):
        """torch.nn.DataParallel with more than one device (the reference's create_model on a multi-GPU node, models/utils.py:27)
        shallow-copies the module into worker threads every forward: the copies would share this module's plan cache, ctypes
        handles and the packed weight blob on cuda:0.  Multi-GPU here is one process per GPU with the batch sharded by the sampler."""
        raise RuntimeError(
            "jodo_amd %s cannot be replicated by torch.nn.DataParallel over several devices: its plans, workspace and packed weights "
            "belong to one device.  Run one process per GPU (python -m torch.distributed.run --nproc-per-node N ...) and pass "
            "shard=(rank, world) to jodo_amd.sampling.get_sampling_fn (jodo_amd/dist.py, INTEGRATION.md section 3); "
            "DataParallel(model, device_ids=[one device]) is supported." % type(self).__name__)

    def __del__(self):
        try:
            L = capi.lib()
            for p in self._plans.values():
                L.jodo_plan_destroy(p['handle'])
        except Exception:
            pass

    # -- forward -------------------------------------------------------------------------------
    def forward(self, t, xh, node_mask, edge_mask, context=None, *args, **kwargs):
        edge_x, cond_x, cond_edge_x = kwargs['edge_x'], kwargs.get('cond_x'), kwargs.get('cond_edge_x')
        noise_level = kwargs['noise_level']
        if not xh.is_cuda:
            raise RuntimeError("jodo_amd DGT runs on an MI355X only (got a %s tensor); there is no CPU fallback — "
                               "use oracle/dgt_oracle.py for CPU checks" % xh.device)
        if self.conditional and context is None:
            raise ValueError("cond_DGT_concat needs `context`")
        if torch.is_grad_enabled() and any(t_ is not None and t_.requires_grad for t_ in (xh, edge_x, cond_x, cond_edge_x, noise_level, context)):
            raise RuntimeError("the HIP DGT returns parameter gradients only: detach the inputs (the reference's loss detaches the "
                               "self-conditioning inputs, losses.py:339, and needs no input gradient)")
        wants_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if wants_grad or (self.training and self.dropout_p > 0):
            # training path (csrc/dgt_train.hip): activations kept for loss.backward(); under model.train() dropout is active
            # in the no-grad self-conditioning forward too (losses.py:335-339), which is why that call comes here as well
            return self._forward_train(xh, edge_x, cond_x, cond_edge_x, noise_level, context, node_mask, edge_mask, wants_grad)
        dev = xh.device
        d = self.dims
        B, N, dims = xh.shape
        if dims != 3 + d.nd or edge_x.shape != (B, N, N, d.ch):
            raise ValueError("shape mismatch: xh %s edge_x %s" % (tuple(xh.shape), tuple(edge_x.shape)))
        f32 = lambda x: None if x is None else x.detach().to(torch.float32).contiguous()
        xh_, ex_, cx_, cex_, nl_ = f32(xh), f32(edge_x), f32(cond_x), f32(cond_edge_x), f32(noise_level)
        ctx_ = f32(context) if self.conditional else None
        streams = min(int(getattr(self, 'n_streams', 1)), 4)     # the plan cache holds 8 plans: every sub-batch plan of a call stays alive
        if streams > 1 and B >= 2 * streams:
            return self._forward_split(streams, node_mask, edge_mask, xh_, ex_, cx_, cex_, nl_, ctx_, dev)
        # plan first: building a plan for a new batch re-checks the weight fingerprint (`.data` updates such as the
        # reference's EMA copy_to / restore) and drops a stale blob BEFORE this call fetches it
        plan = self._plan(node_mask, edge_mask, dev)
        wkey, blob, woff_c, n_woff = self._weights(dev)
        if plan.get('split_tape') is not None and plan.get('split_key') != wkey:
            self._hand_over_split(plan)                          # the weights changed under a pinned plan: the split tape follows the blob
        out_x = torch.empty_like(xh_)
        out_e = torch.empty_like(ex_)
        self._launch(plan, blob, woff_c, n_woff, xh_, ex_, cx_, cex_, nl_, ctx_, out_x, out_e)
        self.last_flags = plan['flags']
        self._last_plan = plan
        self._last_plans = [plan]
        return out_x, out_e

    # -- training path ---------------------------------------------------------------------------
    def _train_engine(self, node_mask, edge_mask, device):
        """One TrainEngine (jodo_train handle + device tables) per batch of atom counts, keyed by the counts themselves.  A shuffling
        loader practically never repeats a key, so a real training step CREATES its engine (host tables of ~4 R ints, one upload):
        tools/train_bench.py times that case (`fresh_batches`) as its headline; the 16-entry cache only serves repeated evaluation of
        one batch (tests, the self-conditioning double forward of a step).  All engines of a module share its activation workspaces
        (TrainEngine.new_pool: two, so that a second grad-enabled forward may run before the first one's backward)."""
        from ..train import TrainEngine
        B, N = node_mask.shape[0], node_mask.shape[1]
        # the same mask tensors as the previous call (the two forwards of a self-conditioned training step): the same engine, no host work
        last = self.__dict__.get('_train_last')
        if (last is not None and last[0] is node_mask and last[1] == node_mask._version and last[2] is edge_mask and last[3] == edge_mask._version
                and last[4] == tuple(sorted((getattr(self, 'train_options', None) or {}).items()))):
            return last[5]
        eng = self._train_engine_lookup(node_mask, edge_mask, device, B, N)
        self.__dict__['_train_last'] = (node_mask, node_mask._version, edge_mask, edge_mask._version, tuple(sorted((getattr(self, 'train_options', None) or {}).items())), eng)
        return eng

    def _train_engine_lookup(self, node_mask, edge_mask, device, B, N):
        from ..train import TrainEngine
        # atom counts that the caller already has on the host (process_edge_batch, jodo_amd/losses.py: the loader's batch is a CPU
        # dict, losses.py:470-476 of the reference, so counts and the mask check cost nothing there): no device round trip, and a
        # training step then runs without any host synchronisation
        hint = getattr(node_mask, '_jodo_counts', None)
        if hint is not None and len(hint) == B:
            return self._train_engine_for(np.ascontiguousarray(hint, dtype=np.int32), N, device)
        nm = node_mask.reshape(B, N)
        n_nodes = nm.sum(1).round().to(torch.int32)
        # mask validation and the atom counts reach the host in ONE transfer (one stream sync per new batch; round 4 paid three:
        # two torch.equal and the counts).  model.validate_masks = False drops the comparison kernels as well (a loader that is
        # known to build prefix masks, like the samplers' build_masks).
        if getattr(self, 'validate_masks', True):
            prefix = (torch.arange(N, device=nm.device).unsqueeze(0) < n_nodes.unsqueeze(1))
            em = edge_mask.reshape(B, N, N)
            want = prefix.unsqueeze(1) & prefix.unsqueeze(2) & (~torch.eye(N, dtype=torch.bool, device=nm.device))
            ok = torch.stack([(prefix.to(nm.dtype) == nm).all(), (want.to(em.dtype) == em).all()]).to(torch.int32)
            host = torch.cat([n_nodes, ok]).cpu().numpy()
            if not host[B]:
                raise ValueError("node_mask must be a prefix mask (real atoms first)")
            if not host[B + 1]:
                raise ValueError("edge_mask must be node_mask x node_mask with the diagonal removed")
            n_host = np.ascontiguousarray(host[:B])
        else:
            n_host = n_nodes.cpu().numpy()
        return self._train_engine_for(n_host, N, device)

    def _train_engine_for(self, n_host, N, device):
        from ..train import TrainEngine
        opts = dict(getattr(self, 'train_options', None) or {})        # {jodo_train_set_option: value}, e.g. {0: 0}: op-by-op forward (tests)
        key = (str(device), N, tuple(sorted(opts.items()))) + tuple(int(v) for v in n_host)
        cache = self.__dict__.setdefault('_train_engines', {})
        eng = cache.pop(key, None)
        if eng is None:
            # (the 351-entry name / shape table of the state_dict is built once per module, not once per batch)
            named = self.__dict__.get('_train_named')
            if named is None:
                named = self.__dict__['_train_named'] = TrainEngine.named_table([(k, tuple(v.shape)) for k, v in self.state_dict().items()])
            pool = self.__dict__.setdefault('_train_pool', TrainEngine.new_pool())     # the activation workspace(s) of this module
            eng = TrainEngine(self._cfg(), n_host, N, named, device, pool=pool, options=opts)
            while len(cache) >= 16:                             # handles are small (index tables); the workspace is shared
                cache.pop(next(iter(cache)))
        cache[key] = eng                                        # most recently used last
        return eng

    def _forward_train(self, xh, edge_x, cond_x, cond_edge_x, noise_level, context, node_mask, edge_mask, wants_grad):
        from ..train import dgt_autograd
        d = self.dims
        B, N, dims = xh.shape
        if dims != 3 + d.nd or edge_x.shape != (B, N, N, d.ch):
            raise ValueError("shape mismatch: xh %s edge_x %s" % (tuple(xh.shape), tuple(edge_x.shape)))
        f32 = lambda x: None if x is None else x.detach().to(torch.float32).contiguous()
        eng = self._train_engine(node_mask, edge_mask, xh.device)
        p = float(self.dropout_p) if self.training else 0.0
        # dropout masks: counter-based, keyed by a seed drawn from torch's generator (so torch.manual_seed reproduces a step)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0
        ctx_ = f32(context) if self.conditional else None
        # parameters in state_dict order = the order the engine was created with (parameters() of this tree, no buffers)
        params = list(self.parameters())
        args = (eng, p, seed, f32(xh), f32(edge_x), f32(cond_x), f32(cond_edge_x), f32(noise_level), ctx_)
        if wants_grad:
            out_x, out_e = dgt_autograd(*args, params)
        else:
            # (nobody differentiates this call — the self-conditioning forward of a training step: backward-only stores are skipped)
            out_x, out_e = eng.forward([q.detach().contiguous() for q in params], *args[3:], p, seed,
                                       save_activations=bool(getattr(self, 'train_save_always', False)))      # (True: A/B runs of tools/train_bench.py)
        self.last_flags = eng.flags
        return out_x, out_e

    def _launch(self, plan, blob, woff_c, n_woff, xh_, ex_, cx_, cex_, nl_, ctx_, out_x, out_e):
        capi.check(capi.lib().jodo_dgt_forward(plan['handle'], capi.ptr(plan['desc']), capi.ptr(blob), woff_c, n_woff,
                                               capi.ptr(xh_), capi.ptr(ex_), capi.ptr(cx_), capi.ptr(cex_), capi.ptr(nl_),
                                               capi.ptr(ctx_), capi.ptr(out_x), capi.ptr(out_e), capi.ptr(plan['flags']),
                                               capi.ptr(plan['ws']), capi.current_stream_ptr()), 'jodo_dgt_forward')

    # -- stream-interleaved evaluation -----------------------------------------------------------
    def _split_of(self, node_mask, edge_mask, k):
        """Contiguous split of the batch into k sub-batches of about equal pair work (sum n^2), cached per mask tensor:
        [(lo, hi, node_mask[lo:hi], edge_mask rows of lo..hi)] — the sub-mask tensors are kept so that their plans stay cached."""
        key = (id(node_mask), id(edge_mask), k)
        ent = self._splits.get(key)
        if ent is not None and ent[0] is node_mask and ent[1] == node_mask._version and ent[3] is edge_mask and ent[4] == edge_mask._version:
            return ent[2]
        B, N = node_mask.shape[0], node_mask.shape[1]
        n = node_mask.reshape(B, N).sum(1).round().long().cpu()                 # one sync per new batch
        cum = torch.cumsum(n.double() ** 2, 0)
        cuts = [0]
        for j in range(1, k):
            c = int(torch.searchsorted(cum, cum[-1] * j / k).item()) + 1
            cuts.append(min(max(c, cuts[-1] + 1), B - (k - j)))
        cuts.append(B)
        em = edge_mask.reshape(B, N * N, -1)
        parts = [(lo, hi, node_mask[lo:hi], em[lo:hi].reshape((hi - lo) * N * N, -1)) for lo, hi in zip(cuts[:-1], cuts[1:])]
        if len(self._splits) >= 4:
            self._splits.pop(next(iter(self._splits)))
        self._splits[key] = (node_mask, node_mask._version, parts, edge_mask, edge_mask._version)
        return parts

    def _forward_split(self, k, node_mask, edge_mask, xh_, ex_, cx_, cex_, nl_, ctx_, dev):
        """n_streams = k > 1: the batch is cut into k contiguous sub-batches (molecules are independent) that are evaluated
        concurrently, sub-batch j on HIP stream j (stream 0 = the caller's).  Every launch of this path is one wave per SIMD
        with item counts that are not multiples of the chip (1409 node strips, 12 666 pair items at QM9 B = 2500), so a
        single stream leaves the tail round of every launch partly idle and serialises the latency-bound prologue /
        epilogue chains; with two streams the other sub-batch's workgroups fill those slots.  The C library still only ever
        sees the stream it is handed; the side streams belong to this module, fork from and join the caller's stream (works
        under HIP-graph capture).  Batch-global semantics of the reference: the NaN guard (mol_gnn.py:587-589) is applied
        across sub-batches below; the first-step switch (:544) is decided per sub-batch (identical unless a whole
        sub-batch's self-conditioning positions coincide — the same caveat as sharding over GPUs)."""
        parts = self._split_of(node_mask, edge_mask, k)
        plans = [self._plan(nm, em, dev) for _, _, nm, em in parts]
        _, blob, woff_c, n_woff = self._weights(dev)
        out_x = torch.empty_like(xh_)
        out_e = torch.empty_like(ex_)
        if getattr(self, '_side_streams', None) is None or len(self._side_streams) < k - 1:
            self._side_streams = [torch.cuda.Stream(device=dev) for _ in range(k - 1)]
        cur = torch.cuda.current_stream()
        cut = lambda t, lo, hi: None if t is None else t[lo:hi]
        for j, ((lo, hi, _, _), plan) in enumerate(zip(parts, plans)):
            args = (plan, blob, woff_c, n_woff, xh_[lo:hi], ex_[lo:hi], cut(cx_, lo, hi), cut(cex_, lo, hi), nl_[lo:hi],
                    cut(ctx_, lo, hi), out_x[lo:hi], out_e[lo:hi])
            if j == 0:
                self._launch(*args)
            else:
                side = self._side_streams[j - 1]
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    self._launch(*args)
        for side in self._side_streams[:k - 1]:
            cur.wait_stream(side)
        flags = torch.stack([p['flags'] for p in plans]).amax(0)
        out_x[:, :, :3] *= (flags[0] == 0).to(out_x.dtype)           # NaN guard is batch-global: any sub-batch resets all positions
        self.last_flags = flags
        self._last_plan = plans[0]
        self._last_plans = plans
        return out_x, out_e

    def take_nan_count(self):
        """Number of evaluations since the last call in which the NaN guard fired (sticky device counter, flags[5] of
        every cached plan); one host sync, so call it once per sampling round, not per step."""
        plans = list(self._plans.values())
        if not plans:
            return 0
        both = torch.stack([p['flags'][5:7] for p in plans]).sum(0).tolist()        # one host sync for all cached plans
        total, violated = int(both[0]), int(both[1])
        if total or violated:
            for p in plans:
                p['flags'][5:7] = 0
        if violated:
            raise RuntimeError("a call on a plan with pinned kernel paths (pin_paths) had inputs that take another path "
                               "(asymmetric edge tensors or per-molecule noise levels): its outputs are invalid")
        return total

    def pin_paths(self):
        """Pin the kernel variants of the plan of the last call to the ones that call used (device flags [2] shared
        modulation row, [4] asymmetric inputs; one host sync): later calls on the same masks launch only those variants
        instead of every variant + device-side early exits (24 idle dispatches per forward at 8 blocks).  For callers that
        know the inputs keep their structure — the samplers inside one round, which unpin when the round ends
        (unpin_paths).  A violating call is detected on the device and reported by take_nan_count(); a state that went NaN
        is not a violation (k_check_sym counts NaN == NaN), it ends in the NaN guard like the reference's."""
        plans = [p for p in getattr(self, '_last_plans', []) if not p.get('pinned')]
        if not plans:
            return
        fl = torch.stack([p['flags'] for p in plans]).cpu().tolist()
        L = capi.lib()
        for plan, f in zip(plans, fl):
            capi.check(L.jodo_plan_set_option(plan['handle'], 4, 2 if f[4] else 1), 'jodo_plan_set_option')
            capi.check(L.jodo_plan_set_option(plan['handle'], 5, 1 if f[2] else 2), 'jodo_plan_set_option')
            plan['pinned'] = True
            if self.split_bf16 and self.dims.D in (256, 384) and not self.conditional and not f[4] and f[2]:
                self._hand_over_split(plan)
                capi.check(L.jodo_plan_set_option(plan['handle'], 13, 2 if self.split_bf16 == 'attention' else 1), 'jodo_plan_set_option')

    def _hand_over_split(self, plan):
        """The split-bf16 weight tape of the CURRENT parameters -> this plan (also called by forward when the packed blob was rebuilt
        under a pinned plan: an optimiser step or load_state_dict between two calls on the same masks)."""
        tape = self._split_weights(plan['ws'].device)
        capi.check(capi.lib().jodo_plan_set_split_weights(plan['handle'], capi.ptr(tape), ctypes.c_size_t(tape.numel())), 'jodo_plan_set_split_weights')
        plan['split_tape'] = tape                             # keeps the device copy alive as long as the plan may use it
        plan['split_key'] = self._split_tape[0]

    def _split_weights(self, device):
        """Device copy of the split-bf16 weight tape of the current parameters (re-packed when the packed blob is)."""
        key = self._weights(device)[0]
        if self._split_tape is None or self._split_tape[0] != key:
            self._split_tape = (key, capi.pack_split_tape(self._cfg(), self.state_dict(), device))
        return self._split_tape[1]

    def unpin_paths(self):
        """End of the scope of pin_paths(): every cached plan goes back to launching all kernel variants and letting the device
        flags decide.  The samplers call it when a round ends (also when it ends in an exception), so a later caller that
        reuses the same mask tensors with other inputs — asymmetric edge tensors, per-molecule noise levels: a likelihood or
        loss evaluation — gets the right kernels instead of a pin violation."""
        L = capi.lib()
        for plan in self._plans.values():
            if plan.get('pinned'):
                capi.check(L.jodo_plan_set_option(plan['handle'], 4, 0), 'jodo_plan_set_option')
                capi.check(L.jodo_plan_set_option(plan['handle'], 5, 0), 'jodo_plan_set_option')
                capi.check(L.jodo_plan_set_option(plan['handle'], 13, 0), 'jodo_plan_set_option')
                plan['pinned'] = False

    # -- measurement plumbing (bench.py): HIP-event class timers and the executed-work model of the last call's plan(s) --
    def profile_enable(self, mode):
        for p in self._last_plans:
            capi.check(capi.lib().jodo_profile_enable(p['handle'], int(mode)), 'jodo_profile_enable')

    def profile_read(self):
        """Summed milliseconds and launch counts per class (8 each) since the last read, over the last call's plans."""
        ms_t, cnt_t = [0.0] * 8, [0] * 8
        for p in self._last_plans:
            ms, cnt = (ctypes.c_float * 8)(), (ctypes.c_int32 * 8)()
            capi.check(capi.lib().jodo_profile_read(p['handle'], ms, cnt), 'jodo_profile_read')
            ms_t = [a + b for a, b in zip(ms_t, ms)]
            cnt_t = [a + b for a, b in zip(cnt_t, cnt)]
        return ms_t, cnt_t

    def work_model(self):
        """Executed MFMA flops per launch class of ONE evaluation like the last one (jodo_plan_work, summed over its plans)."""
        tot = [0.0] * 8
        for p, f in zip(self._last_plans, torch.stack([p['flags'] for p in self._last_plans]).cpu().tolist()):
            w = (ctypes.c_double * 8)()
            capi.check(capi.lib().jodo_plan_work(p['handle'], int(f[2]), int(not f[4]), w), 'jodo_plan_work')
            tot = [a + b for a, b in zip(tot, w)]
        return tot

    def invalidate_packed_weights(self):
        """Drop the packed kernel weights; the next forward re-packs from the current parameters.  Needed after
        in-place updates that do not bump tensor versions (`param.data.copy_(...)`, e.g. the reference's
        ExponentialMovingAverage.copy_to / restore, models/ema.py:44-66)."""
        self._packed = None

    def nan_guard_fired(self):
        """Lazy read of the device NaN-guard flag of the last call (the reference prints a warning and
        zeroes all positions, mol_gnn.py:587-589; the zeroing already happened on the device)."""
        return self.last_flags is not None and bool(self.last_flags[0].item())


@utils.register_model(name='DGT_concat')
class DGT_concat(_DGTBase):
    """Diffusion Graph Transformer with self-conditioning (HIP)."""
    conditional = False


@utils.register_model(name='cond_DGT_concat')
class Cond_DGT_concat(_DGTBase):
    """Conditional Diffusion Graph Transformer with self-conditioning (HIP)."""
    conditional = True
