"""HIP-graph replay of the ancestral sampling loop ("enables HIP-graph capture of a whole
step").  One denoising step — noise level from a device table, score network (~80 launches through the C
ABI), three normal draws, fused update, state hand-over — is captured ONCE with torch.cuda.graph and
replayed for every remaining step, which removes the per-step Python / launch overhead (≈ 0.4 ms per step:
2 % at QM9 B = 2500, 7 % at B = 313).

What makes the step replayable: every per-step scalar lives in device memory (`coef` table
[steps][4] = c_x, c_pred, sigma, noise_level, indexed by a device step counter that the graph itself
increments), the state lives in static buffers, and the model's plan / packed weights / workspace are created
by the eager first steps before capture.  Step 0 (no self-conditioning input yet: a different kernel path)
and one warm-up step run eagerly.

Semantics are those of AncestralSampler.sampling (sampling.py:530-596): same update, same noise construction;
the random stream differs from an eager run (graph-captured generators advance their Philox offset per
replay), which is why parity tests compare each replayed step against the framework formula on the step's
own recorded draws rather than against an eager trajectory.
"""
import torch

from . import capi, fused
from .sampling import posterior_coefficients


class GraphedAncestralRound:
    def __init__(self, sampler, model, node_mask, edge_mask, context=None, record=False):
        if not (sampler.self_cond and sampler.model_pred_data and sampler.pred_edge):
            raise NotImplementedError("graph replay covers the self-conditioned data-prediction sampler (the JODO configs)")
        if sampler.noise_fn is not None:
            raise ValueError("noise_fn replay and graph capture are mutually exclusive")
        self.sampler, self.model = sampler, model
        self.node_mask, self.edge_mask, self.context = node_mask, edge_mask, context
        self.record = record                    # keep per-step snapshots (tests)
        self.history = []
        ns = sampler.noise_scheduler
        steps = len(sampler.t_array)
        tab = torch.empty(steps, 4, dtype=torch.float32)
        for i in range(steps):                  # host scalars, as the eager sampler computes them
            c_x, c_pred, sigma, alpha_t, sigma_t, _, _ = posterior_coefficients(ns, sampler.t_array[i], sampler.s_array[i])
            tab[i, 0], tab[i, 1], tab[i, 2] = float(c_x), float(c_pred), float(sigma)
            tab[i, 3] = float(torch.log(alpha_t ** 2 / sigma_t ** 2))
        self.steps = steps
        self.tab_host = tab
        self.graph = None

    # one step on static buffers; identical code runs eagerly (warm-up) and under capture
    def _one_step(self):
        L = capi.lib()
        B, N, F = self.x.shape
        ch = self.e.shape[-1]
        st = capi.current_stream_ptr()
        capi.check(L.jodo_step_begin(B, capi.ptr(self.tab), capi.ptr(self.step), capi.ptr(self.nl), st), 'jodo_step_begin')
        pred, epred = self.model(self.nl, self.x, self.node_mask, self.edge_mask, edge_x=self.e, noise_level=self.nl,
                                 cond_x=self.cx, cond_edge_x=self.cex, context=self.context)
        if self.record:
            self.pred_keep.copy_(pred); self.epred_keep.copy_(epred)
            self.x_prev.copy_(self.x); self.e_prev.copy_(self.e)
        # self-conditioning post-process BEFORE the update, as the eager sampler and the reference do
        # (sampling.py:553-571): 'clamp' clamps the atom / charge channels of pred in place, and the update uses them
        cxn, cexn = self.sampler.cond_process_fn(pred, epred)
        self.cx.copy_(cxn); self.cex.copy_(cexn)
        self.eps_pos.normal_()                     # draw order of models/utils.py:67-99: positions, features, edges
        self.eps_feat.normal_()
        self.eps_edge.normal_()
        capi.check(L.jodo_sampler_step_tab(B, N, F, ch, capi.ptr(self.n_nodes), capi.ptr(self.tab), capi.ptr(self.step),
                                           capi.ptr(self.x), capi.ptr(self.e), capi.ptr(pred), capi.ptr(epred),
                                           capi.ptr(self.eps_pos), capi.ptr(self.eps_feat), capi.ptr(self.eps_edge),
                                           capi.ptr(self.x_next), capi.ptr(self.e_next), capi.ptr(self.x_mean),
                                           capi.ptr(self.e_mean), st), 'jodo_sampler_step_tab')
        self.x.copy_(self.x_next); self.e.copy_(self.e_next)
        capi.check(L.jodo_step_end(capi.ptr(self.step), st), 'jodo_step_end')

    def _snapshot(self):
        if self.record:
            torch.cuda.synchronize()
            self.history.append({k: getattr(self, k).clone() for k in
                                 ('x_prev', 'e_prev', 'pred_keep', 'epred_keep', 'eps_pos', 'eps_feat', 'eps_edge', 'x', 'e',
                                  'x_mean', 'e_mean')} | {'step': int(self.step.item()) - 1})

    def prepare(self, z_T, edge_z_T):
        """Eager step 0 and warm-up step 1, then capture.  Afterwards `replay()` advances one step."""
        smp, dev = self.sampler, z_T.device
        st = smp.init_state(z_T, edge_z_T)
        st = smp.step(self.model, 0, st, self.node_mask, self.edge_mask, self.context)      # step 0: eager, cond = None
        self.done = 1
        self.last = (st['x_mean'], st['edge_x_mean'])
        if self.steps == 1:
            return
        new = lambda t: torch.empty(t.shape, dtype=torch.float32, device=dev)
        self.x, self.e = st['x'].contiguous().clone(), st['edge_x'].contiguous().clone()
        self.cx, self.cex = st['cond_x'].contiguous().clone(), st['cond_edge_x'].contiguous().clone()
        self.x_next, self.e_next, self.x_mean, self.e_mean = new(self.x), new(self.e), new(self.x), new(self.e)
        B, N, F = self.x.shape
        ch = self.e.shape[-1]
        self.eps_pos, self.eps_feat = torch.empty(B, N, 3, device=dev), torch.empty(B, N, F - 3, device=dev)
        self.eps_edge = torch.empty(B, ch, N, N, device=dev)
        self.nl = torch.empty(B, device=dev)
        self.tab = self.tab_host.to(dev)
        self.step = torch.ones(1, dtype=torch.int32, device=dev)          # next step to run
        self.n_nodes = fused.n_nodes_from_mask(self.node_mask)
        if self.record:
            self.pred_keep, self.epred_keep, self.x_prev, self.e_prev = new(self.x), new(self.e), new(self.x), new(self.e)
        # warm-up step on a side stream (torch.cuda.graph requirement), then capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._one_step()
        torch.cuda.current_stream().wait_stream(side)
        self._snapshot()
        self.done = 2
        self.last = (self.x_mean, self.e_mean)
        if self.done < self.steps:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._one_step()

    def replay(self):
        if self.graph is None or self.done >= self.steps:
            raise RuntimeError("no step left to replay")
        self.graph.replay()
        self.done += 1
        self._snapshot()

    def run(self, z_T, edge_z_T):
        """Returns (x_mean, edge_x_mean) of the last step, like AncestralSampler.sampling."""
        self.prepare(z_T, edge_z_T)
        while self.done < self.steps:
            self.replay()
        return self.last
