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
import ctypes

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
        if self.rng is not None:                   # draws generated inside the update kernel: draw index = step index
            capi.check(L.jodo_sampler_step_rng(B, N, F, ch, capi.ptr(self.n_nodes), ctypes.c_float(0), ctypes.c_float(0),
                                               ctypes.c_float(0), capi.ptr(self.tab), capi.ptr(self.step),
                                               ctypes.c_uint64(self.rng.seed), ctypes.c_uint32(self.rng_base),
                                               capi.ptr(self.x), capi.ptr(self.e), capi.ptr(pred), capi.ptr(epred),
                                               capi.ptr(self.x_next), capi.ptr(self.e_next), capi.ptr(self.x_mean),
                                               capi.ptr(self.e_mean), st), 'jodo_sampler_step_rng')
        else:
            self.eps_pos.normal_()                 # draw order of models/utils.py:67-99: positions, features, edges
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
        self.rng = smp.device_noise
        if self.rng is not None:
            self.rng_base = self.rng.draw - 1            # step 0 took draw `rng_base`; step i takes rng_base + i
            self.rng.draw += self.steps - 1
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
        from .models.utils import model_hook
        pin = model_hook(self.model, 'pin_paths')      # the captured step launches only the variants this round uses
        if pin is not None:
            pin()
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


class GraphedDPMRound:
    """HIP-graph replay of the hybrid DPM-solver loop (BASELINE config 5: single-step, order 2, `steps` = NFE): ONE outer
    step — two score-network evaluations, two position-noise draws, two fused solver updates (jodo_dpm_update), state
    hand-over — is captured once and replayed for the remaining outer steps.  All per-step scalars sit in a device table
    [K][16] (two rows of jodo_dpm_update coefficients incl. the evaluation's noise level) indexed by a device step
    counter.  The first outer step (no self-conditioning input yet) and one warm-up step run eagerly.  Semantics:
    DPM_Solver_hybrid.sampling, mix_dpm_solver.py:304-376, with singlestep_dpm_solver_second_update :96-149."""

    def __init__(self, solver, model, node_mask, edge_mask, context=None):
        if solver.method != 'singlestep_fixed' or solver.order != 2:
            raise NotImplementedError("graph replay covers the single-step second-order solver (BASELINE config 5)")
        if solver.noise_fn is not None:
            raise ValueError("noise_fn replay and graph capture are mutually exclusive")
        self.solver, self.model = solver, model
        self.node_mask, self.edge_mask, self.context = node_mask, edge_mask, context
        ns = solver.noise_schedule
        K = solver.steps // 2
        outer = solver.get_time_steps('time_uniform', ns.T, 1. / ns.total_N, K, 'cpu')
        tab = torch.zeros(K, 16, dtype=torch.float32)
        for k in range(K):
            ts, te = outer[k], outer[k + 1]
            inner = solver.get_time_steps('time_uniform', ts.item(), te.item(), 2, 'cpu')
            lam = ns.marginal_lambda(inner)
            r1 = (lam[1] - lam[0]) / (lam[-1] - lam[0])
            c = solver.second_order_coefficients(ts, te, r1)
            cx1, cp1, sg1 = solver.position_coefficients(ts, c['s1'])
            cx2, cp2, sg2 = solver.position_coefficients(c['s1'], te)
            tab[k, 0:8] = torch.tensor([float(cx1), float(cp1), float(sg1), float(c['a1']), float(c['b1']), 0.0, 1.0,
                                        float(ns.get_noiseLevel(ts))])
            tab[k, 8:16] = torch.tensor([float(cx2), float(cp2), 0.0 if k == K - 1 else float(sg2), float(c['a2']), float(c['b2']),
                                         float(c['c2']), 1.0, float(ns.get_noiseLevel(c['s1']))])
        self.K, self.tab_host, self.graph = K, tab, None

    def _outer_step(self):
        L = capi.lib()
        B, N, F = self.x.shape
        ch = self.e.shape[-1]
        st = capi.current_stream_ptr()
        def up(col, x_pos, P, DA, DB, PP, xo, eo):
            tens = [capi.ptr(t) for t in (x_pos, self.x, self.e, P[0], P[1], DA[0], DA[1], DB[0], DB[1], PP[0])]
            if self.rng is not None:                       # position noise drawn in the kernel: draws 2k and 2k + 1 of outer step k
                capi.check(L.jodo_dpm_update_rng(B, N, F, ch, capi.ptr(self.n_nodes), None, capi.ptr(self.tab), capi.ptr(self.step),
                                                 16, col, ctypes.c_uint64(self.rng.seed), ctypes.c_uint32(self.rng_base + col // 8),
                                                 ctypes.c_uint32(2), *tens, capi.ptr(xo), capi.ptr(eo), st), 'jodo_dpm_update_rng')
            else:
                self.eps.normal_()
                capi.check(L.jodo_dpm_update(B, N, F, ch, capi.ptr(self.n_nodes), None, capi.ptr(self.tab), capi.ptr(self.step), 16,
                                             col, *tens, capi.ptr(self.eps), capi.ptr(xo), capi.ptr(eo), st), 'jodo_dpm_update')
        capi.check(L.jodo_step_begin_at(B, capi.ptr(self.tab), capi.ptr(self.step), 16, 0, capi.ptr(self.nl), st), 'jodo_step_begin_at')
        p0 = self.model(self.nl, self.x, self.node_mask, self.edge_mask, edge_x=self.e, noise_level=self.nl, cond_x=self.cx,
                        cond_edge_x=self.cex, context=self.context)
        self.cx.copy_(p0[0]); self.cex.copy_(p0[1])            # self-conditioning input of the next evaluation (:296-302)
        c0 = (self.cx, self.cex)
        up(0, self.x, c0, c0, c0, c0, self.x1, self.e1)
        capi.check(L.jodo_step_begin_at(B, capi.ptr(self.tab), capi.ptr(self.step), 16, 8, capi.ptr(self.nl), st), 'jodo_step_begin_at')
        p1 = self.model(self.nl, self.x1, self.node_mask, self.edge_mask, edge_x=self.e1, noise_level=self.nl, cond_x=self.cx,
                        cond_edge_x=self.cex, context=self.context)
        up(8, self.x1, c0, p1, c0, p1, self.x2, self.e2)
        self.cx.copy_(p1[0]); self.cex.copy_(p1[1])
        self.x.copy_(self.x2); self.e.copy_(self.e2)
        capi.check(L.jodo_step_end(capi.ptr(self.step), st), 'jodo_step_end')

    def run(self, x, edge_x):
        """Returns (x, edge_x) at t_end like DPM_Solver_hybrid.sampling."""
        sv, dev = self.solver, x.device
        ns = sv.noise_schedule
        sv.cond_x = sv.cond_edge_x = None
        sv._noise_calls = 0
        sv._n_nodes_dev = fused.n_nodes_from_mask(self.node_mask)      # this round's atom counts (the solver serves many rounds)
        sv._pinned = False
        self.rng = sv.device_noise
        model_fn = sv.get_model_fn(self.model)
        outer = sv.get_time_steps('time_uniform', ns.T, 1. / ns.total_N, self.K, 'cpu')
        inner = sv.get_time_steps('time_uniform', outer[0].item(), outer[1].item(), 2, 'cpu')
        lam = ns.marginal_lambda(inner)
        r1 = (lam[1] - lam[0]) / (lam[-1] - lam[0])
        x, edge_x = sv.singlestep_dpm_solver_update(model_fn, x, self.node_mask, self.edge_mask, edge_x, self.context, outer[0],
                                                    outer[1], self.K == 1, order=2, r1=r1)      # outer step 0: eager, cond = None
        if self.K > 1:
            new = lambda t: torch.empty(t.shape, dtype=torch.float32, device=dev)
            self.x, self.e = x.contiguous().clone(), edge_x.contiguous().clone()
            self.cx, self.cex = sv.cond_x.contiguous().clone(), sv.cond_edge_x.contiguous().clone()
            self.x1, self.e1, self.x2, self.e2 = new(self.x), new(self.e), new(self.x), new(self.e)
            self.eps = torch.empty(self.x.shape[0], self.x.shape[1], 3, device=dev)
            self.nl = torch.empty(self.x.shape[0], device=dev)
            self.tab = self.tab_host.to(dev)
            self.step = torch.ones(1, dtype=torch.int32, device=dev)
            self.n_nodes = sv._n_nodes_dev
            if self.rng is not None:
                self.rng_base = self.rng.draw - 2                # outer step 0 took two draws; outer step k takes 2k, 2k + 1
                self.rng.draw += 2 * (self.K - 1) - 1
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._outer_step()                               # outer step 1: eager warm-up on the static buffers
            torch.cuda.current_stream().wait_stream(side)
            from .models.utils import model_hook
            pin = model_hook(self.model, 'pin_paths')
            if pin is not None:
                pin()
            if self.K > 2:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self._outer_step()
                for _ in range(self.K - 2):                # capture records, it does not run: K - 2 outer steps are left
                    self.graph.replay()
            x, edge_x = self.x.clone(), self.e.clone()
        from .models.utils import assert_mean_zero_with_mask
        assert_mean_zero_with_mask(x[:, :, :3], self.node_mask)
        return x, edge_x
