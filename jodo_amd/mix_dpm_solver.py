"""Hybrid DPM-Solver++ sampler (BASELINE config 5): atom-type / charge / bond channels follow
DPM-Solver++ data-prediction updates of order 1-3 (single-step) or 2 (multi-step); positions always
take an ancestral (stochastic) step.  Self-conditioning state is carried between model calls.

Behaviour of /root/reference/mix_dpm_solver.py: `DPM_Solver_hybrid.sampling` (:304-376), position
update (:44-59), first/second/third-order single-step updates (:61-227), multistep second order
(:230-265), `get_model_fn` (:296-302).  The score-network call signature is unchanged.
`steps` counts model evaluations (NFE): K = steps // order outer steps of `order` evaluations each.
"""
import torch

from .models.utils import assert_mean_zero_with_mask, model_hook, sample_center_gravity_zero_gaussian_with_mask


def split_x(x):
    return x[:, :, :3], x[:, :, 3:]


def merge_x(pos, feat):
    return torch.cat([pos, feat], dim=-1)


class DPM_Solver_hybrid:
    def __init__(self, noise_schedule, config, noise_fn=None, fused=True, device_noise=None):
        self.noise_schedule = noise_schedule
        self.device_noise = device_noise  # fused.DeviceNoise: position noise drawn inside jodo_dpm_update_rng (Philox)
        self._n_nodes_dev = None          # int32 [B] atom counts of the round being sampled (set by `sampling`)
        self._pinned = False
        self.cond_x = None
        self.cond_edge_x = None
        self.order = config.sampling.dpm_solver_order
        self.steps = config.sampling.steps
        self.method = config.sampling.dpm_solver_method
        self.noise_fn = noise_fn          # optional replay hook: noise_fn(call_index, 'pos', like) -> tensor
        self._noise_calls = 0
        self.fused = fused                # GPU tensors: one fused kernel per tensor and update (jodo_dpm_update)
        assert config.model.pred_data, "Not support in current version."
        assert config.model.self_cond, "Not support in current version."

    def noise_draws_per_round(self):
        """Position-noise draws of one `sampling` call: every update draws once except the last."""
        if self.method == 'singlestep_fixed':
            return (self.steps // self.order) * self.order - 1
        return self.steps - 1

    @staticmethod
    def get_time_steps(skip_type, t_T, t_0, N, device):
        if skip_type != 'time_uniform':
            raise ValueError("Unsupported skip_type {}".format(skip_type))
        return torch.linspace(t_T, t_0, N + 1).to(device)

    # -- positions: one ancestral step t_start -> t_end --------------------------------------
    def position_coefficients(self, t_start, t_end):
        """(c_x, c_pred, sigma) of the ancestral position step (:44-59)."""
        ns = self.noise_schedule
        alpha_t, sigma_t = ns.marginal_prob(t_start)
        alpha_s, sigma_s = ns.marginal_prob(t_end)
        a_ts = alpha_t / alpha_s
        var_ts = sigma_t ** 2 - a_ts ** 2 * sigma_s ** 2
        sigma = torch.sqrt(var_ts) * sigma_s / sigma_t
        return a_ts * sigma_s ** 2 / sigma_t ** 2, alpha_s * var_ts / sigma_t ** 2, sigma

    def ancestral_position_update(self, position_x, position_pred, node_mask, t_start, t_end, last_step=False):
        c_x, c_pred, sigma = self.position_coefficients(t_start, t_end)
        position = c_x * position_x + c_pred * position_pred
        if not last_step:
            if self.noise_fn is not None:
                eps = self.noise_fn(self._noise_calls, 'pos', position_x)
            else:
                eps = sample_center_gravity_zero_gaussian_with_mask(position_x.size(), position_x.device, node_mask)
            self._noise_calls += 1
            position = position + sigma * eps
        return position

    def _update(self, x_pos, x_base, edge_base, P, DA, DB, PP, node_mask, t_from, t_to, last_step, a, b, c=None, c2=None):
        """One solver update of the whole state:
            positions   ancestral step t_from -> t_to driven by the position channels of PP           (:44-59)
            the rest    a * base - b * P - c * (c2 * (DA - DB))     (node feature channels and the edge tensor)
        P / DA / DB / PP are (node prediction, edge prediction) pairs.  On GPU tensors (and when no noise is being
        replayed) this is ONE fused kernel per tensor (jodo_dpm_update, csrc/sampler_kernels.hip) instead of ~25
        framework launches; the op order of the products is the same in both paths."""
        if self.fused and x_base.is_cuda:
            from . import fused
            c_x, c_pred, sigma = self.position_coefficients(t_from, t_to)
            coef = [float(c_x), float(c_pred), 0.0 if last_step else float(sigma), float(a), float(b),
                    0.0 if c is None else float(c), 1.0 if c2 is None else float(c2), 0.0]
            eps = None
            if not last_step:
                if self.noise_fn is not None:         # replayed draw (masked, CoM-free: the kernel's own steps are idempotent)
                    eps = self.noise_fn(self._noise_calls, 'pos', split_x(x_pos)[0])
                self._noise_calls += 1
            if self._n_nodes_dev is None or self._n_nodes_dev.shape[0] != x_base.shape[0]:
                self._n_nodes_dev = fused.n_nodes_from_mask(node_mask)      # direct callers of the update methods
            return fused.dpm_update(self, coef, x_pos, x_base, edge_base, P, DA, DB, PP, self._n_nodes_dev, eps=eps,
                                    rng=self.device_noise if self.noise_fn is None else None)
        pos_base, atom_base = split_x(x_base)
        _, atom_p = split_x(P[0])
        atom = a * atom_base - b * atom_p
        edge = a * edge_base - b * P[1]
        if c is not None:
            d_atom, d_edge = split_x(DA[0])[1] - split_x(DB[0])[1], DA[1] - DB[1]
            if c2 is not None:
                d_atom, d_edge = c2 * d_atom, c2 * d_edge
            atom = atom - c * d_atom
            edge = edge - c * d_edge
        pos = self.ancestral_position_update(split_x(x_pos)[0], split_x(PP[0])[0], node_mask, t_from, t_to, last_step)
        return merge_x(pos, atom), edge

    def _predict(self, model_fn, x, node_mask, edge_mask, edge_x, context, t):
        bs = x.size(0)
        nl = self.noise_schedule.get_noiseLevel(t)
        ones = torch.ones(bs, device=x.device)
        return model_fn(x, node_mask, edge_mask, edge_x, context, ones * t, ones * nl)

    # -- order 1 (DDIM on the discrete channels) ---------------------------------------------
    def dpm_solver_first_update(self, model_fn, x, node_mask, edge_mask, edge_x, context, t_start, t_end,
                                last_step, pred_start=None, edge_pred_start=None):
        ns = self.noise_schedule
        h = ns.marginal_lambda(t_end) - ns.marginal_lambda(t_start)
        sigma_start, sigma_end = ns.marginal_std(t_start), ns.marginal_std(t_end)
        alpha_end = torch.exp(ns.marginal_log_mean_coeff(t_end))
        phi_1 = torch.expm1(-h)
        if pred_start is None and edge_pred_start is None:
            pred_start, edge_pred_start = self._predict(model_fn, x, node_mask, edge_mask, edge_x, context, t_start)
        p0 = (pred_start, edge_pred_start)
        return self._update(x, x, edge_x, p0, p0, p0, p0, node_mask, t_start, t_end, last_step,
                            sigma_end / sigma_start, alpha_end * phi_1)

    # -- order 2, single step ------------------------------------------------------------------
    def second_order_coefficients(self, t_start, t_end, r1):
        ns = self.noise_schedule
        lam0, lam1 = ns.marginal_lambda(t_start), ns.marginal_lambda(t_end)
        h = lam1 - lam0
        s1 = ns.inverse_lambda(lam0 + r1 * h).reshape(())     # scalar (0-dim): broadcasts across devices
        sigma_start, sigma_s1, sigma_end = ns.marginal_std(t_start), ns.marginal_std(s1), ns.marginal_std(t_end)
        alpha_s1 = torch.exp(ns.marginal_log_mean_coeff(s1))
        alpha_end = torch.exp(ns.marginal_log_mean_coeff(t_end))
        phi_11 = torch.expm1(-r1 * h)
        phi_1 = torch.expm1(-h)
        return dict(s1=s1, a1=sigma_s1 / sigma_start, b1=alpha_s1 * phi_11, a2=sigma_end / sigma_start, b2=alpha_end * phi_1,
                    c2=(0.5 / r1) * (alpha_end * phi_1))

    def singlestep_dpm_solver_second_update(self, model_fn, x, node_mask, edge_mask, edge_x, context,
                                            t_start, t_end, last_step, r1=0.5):
        r1 = 0.5 if r1 is None else r1
        k = self.second_order_coefficients(t_start, t_end, r1)
        s1 = k['s1']
        p0 = self._predict(model_fn, x, node_mask, edge_mask, edge_x, context, t_start)
        x_s1, edge_s1 = self._update(x, x, edge_x, p0, p0, p0, p0, node_mask, t_start, s1, False, k['a1'], k['b1'])
        p1 = self._predict(model_fn, x_s1, node_mask, edge_mask, edge_s1, context, s1)
        return self._update(x_s1, x, edge_x, p0, p1, p0, p1, node_mask, s1, t_end, last_step, k['a2'], k['b2'], k['c2'])

    # -- order 3, single step ------------------------------------------------------------------
    def singlestep_dpm_solver_third_update(self, model_fn, x, node_mask, edge_mask, edge_x, context,
                                           t_start, t_end, last_step, r1=1. / 3., r2=2. / 3.):
        r1 = 1. / 3. if r1 is None else r1
        r2 = 2. / 3. if r2 is None else r2
        ns = self.noise_schedule
        lam0, lam1 = ns.marginal_lambda(t_start), ns.marginal_lambda(t_end)
        h = lam1 - lam0
        s1 = ns.inverse_lambda(lam0 + r1 * h).reshape(())     # scalar (0-dim): broadcasts across devices
        s2 = ns.inverse_lambda(lam0 + r2 * h).reshape(())
        sigma_start, sigma_s1, sigma_s2, sigma_end = (ns.marginal_std(t_start), ns.marginal_std(s1),
                                                      ns.marginal_std(s2), ns.marginal_std(t_end))
        alpha_s1 = torch.exp(ns.marginal_log_mean_coeff(s1))
        alpha_s2 = torch.exp(ns.marginal_log_mean_coeff(s2))
        alpha_end = torch.exp(ns.marginal_log_mean_coeff(t_end))
        phi_11 = torch.expm1(-r1 * h)
        phi_12 = torch.expm1(-r2 * h)
        phi_1 = torch.expm1(-h)
        phi_22 = torch.expm1(-r2 * h) / (r2 * h) + 1.
        phi_2 = phi_1 / h + 1.
        p0 = self._predict(model_fn, x, node_mask, edge_mask, edge_x, context, t_start)
        x_s1, edge_s1 = self._update(x, x, edge_x, p0, p0, p0, p0, node_mask, t_start, s1, False,
                                     sigma_s1 / sigma_start, alpha_s1 * phi_11)
        p1 = self._predict(model_fn, x_s1, node_mask, edge_mask, edge_s1, context, s1)
        # the reference ADDS the difference terms here (:199-202, :216-219): c is the negated coefficient
        x_s2, edge_s2 = self._update(x_s1, x, edge_x, p0, p1, p0, p1, node_mask, s1, s2, False,
                                     sigma_s2 / sigma_start, alpha_s2 * phi_12, -(r2 / r1 * (alpha_s2 * phi_22)))
        p2 = self._predict(model_fn, x_s2, node_mask, edge_mask, edge_s2, context, s2)
        return self._update(x_s2, x, edge_x, p0, p2, p0, p2, node_mask, s2, t_end, last_step,
                            sigma_end / sigma_start, alpha_end * phi_1, -((1. / r2) * (alpha_end * phi_2)))

    # -- order 2, multistep --------------------------------------------------------------------
    def multistep_dpm_solver_second_update(self, model_fn, x, node_mask, edge_mask, edge_x, context,
                                           model_prev_list, t_prev_list, t, last_step):
        ns = self.noise_schedule
        p_m1, p_m0 = model_prev_list[-2], model_prev_list[-1]
        t_m1, t_m0 = t_prev_list[-2], t_prev_list[-1]
        lam_m1, lam_m0, lam_t = ns.marginal_lambda(t_m1), ns.marginal_lambda(t_m0), ns.marginal_lambda(t)
        sigma_m0, sigma_t = ns.marginal_std(t_m0), ns.marginal_std(t)
        alpha_t = torch.exp(ns.marginal_log_mean_coeff(t))
        h_0 = lam_m0 - lam_m1
        h = lam_t - lam_m0
        r0 = h_0 / h
        phi_1 = torch.expm1(-h)
        return self._update(x, x, edge_x, p_m0, p_m0, p_m1, p_m0, node_mask, t_prev_list[-1], t, last_step,
                            sigma_t / sigma_m0, alpha_t * phi_1, 0.5 * (alpha_t * phi_1), 1. / r0)

    def singlestep_dpm_solver_update(self, model_fn, x, node_mask, edge_mask, edge_x, context, t_start, t_end,
                                     last_step, order, r1=None, r2=None):
        a = (model_fn, x, node_mask, edge_mask, edge_x, context, t_start, t_end, last_step)
        if order == 1:
            return self.dpm_solver_first_update(*a)
        if order == 2:
            return self.singlestep_dpm_solver_second_update(*a, r1=r1)
        if order == 3:
            return self.singlestep_dpm_solver_third_update(*a, r1=r1, r2=r2)
        raise ValueError("Solver order Error")

    def multistep_dpm_solver_update(self, model_fn, x, node_mask, edge_mask, edge_x, context, model_prev_list,
                                    t_prev_list, t, last_step, order):
        if order == 1:
            return self.dpm_solver_first_update(model_fn, x, node_mask, edge_mask, edge_x, context,
                                                t_prev_list[-1], t, last_step,
                                                pred_start=model_prev_list[-1][0],
                                                edge_pred_start=model_prev_list[-1][1])
        if order == 2:
            return self.multistep_dpm_solver_second_update(model_fn, x, node_mask, edge_mask, edge_x, context,
                                                           model_prev_list, t_prev_list, t, last_step)
        raise ValueError("Solver order Error")

    def get_model_fn(self, model):
        def model_fn(x, node_mask, edge_mask, edge_x, context, vec_t, noise_level):
            pred_t, edge_pred_t = model(vec_t, x, node_mask, edge_mask, edge_x=edge_x, noise_level=noise_level,
                                        cond_x=self.cond_x, cond_edge_x=self.cond_edge_x, context=context)
            if self.cond_x is not None and not self._pinned and x.is_cuda:
                # first self-conditioned evaluation of the round: pin the HIP model's kernel variants (sampling.py, step 1)
                pin = model_hook(model, 'pin_paths')
                if pin is not None:
                    pin()
                self._pinned = True
            self.cond_x, self.cond_edge_x = pred_t, edge_pred_t
            return pred_t, edge_pred_t
        return model_fn

    @torch.no_grad()
    def sampling(self, model, x, node_mask, edge_mask, edge_x, context=None, t_start=None, t_end=None,
                 skip_type='time_uniform'):
        try:
            return self._sampling_round(model, x, node_mask, edge_mask, edge_x, context, t_start, t_end, skip_type)
        finally:
            unpin = model_hook(model, 'unpin_paths')       # the pin of the first self-conditioned evaluation is scoped to this round
            if unpin is not None:
                unpin()

    def _sampling_round(self, model, x, node_mask, edge_mask, edge_x, context, t_start, t_end, skip_type):
        steps, order = self.steps, self.order
        self.cond_x = self.cond_edge_x = None
        self._noise_calls = 0
        self._n_nodes_dev = None
        self._pinned = False
        if self.fused and x.is_cuda:
            from . import fused
            self._n_nodes_dev = fused.n_nodes_from_mask(node_mask)      # per ROUND: a solver object serves many rounds
        model_fn = self.get_model_fn(model)
        ns = self.noise_schedule
        t_0 = 1. / ns.total_N if t_end is None else t_end
        t_T = ns.T if t_start is None else t_start
        assert t_0 > 0 and t_T > 0, "Time range needs to be greater than 0."
        device = 'cpu'       # schedule scalars on the host (see sampling.get_sampling_fn); 0-dim CPU tensors
                             # broadcast against device tensors

        if self.method == 'singlestep_fixed':
            K = steps // order
            outer = self.get_time_steps(skip_type, t_T, t_0, K, device)
            for step in range(K):
                ts, te = outer[step], outer[step + 1]
                inner = self.get_time_steps(skip_type, ts.item(), te.item(), order, device)
                lam = ns.marginal_lambda(inner)
                h = lam[-1] - lam[0]
                r1 = None if order <= 1 else (lam[1] - lam[0]) / h
                r2 = None if order <= 2 else (lam[2] - lam[0]) / h
                x, edge_x = self.singlestep_dpm_solver_update(model_fn, x, node_mask, edge_mask, edge_x, context,
                                                              ts, te, step == K - 1, order=order, r1=r1, r2=r2)
        elif self.method == 'multistep':
            ts = self.get_time_steps(skip_type, t_T, t_0, steps, device)
            assert ts.shape[0] - 1 == steps
            t = ts[0]
            t_prev = [t]
            m_prev = [self._predict(model_fn, x, node_mask, edge_mask, edge_x, context, t)]
            for step in range(1, order):                 # warm-up with lower orders
                t = ts[step]
                x, edge_x = self.multistep_dpm_solver_update(model_fn, x, node_mask, edge_mask, edge_x, context,
                                                             m_prev, t_prev, t, last_step=False, order=step)
                t_prev.append(t)
                m_prev.append(self._predict(model_fn, x, node_mask, edge_mask, edge_x, context, t))
            for step in range(order, steps + 1):
                t = ts[step]
                x, edge_x = self.multistep_dpm_solver_update(model_fn, x, node_mask, edge_mask, edge_x, context,
                                                             m_prev, t_prev, t, last_step=step == steps,
                                                             order=order)
                for i in range(order - 1):
                    t_prev[i] = t_prev[i + 1]
                    m_prev[i] = m_prev[i + 1]
                t_prev[-1] = t
                if step < steps:                         # the final model value is never needed
                    m_prev[-1] = self._predict(model_fn, x, node_mask, edge_mask, edge_x, context, t)
        else:
            raise ValueError("Get wrong method {}".format(self.method))

        assert_mean_zero_with_mask(x[:, :, :3], node_mask)
        if self.fused and x.is_cuda:
            x, edge_x = x.clone(), edge_x.clone()        # the fused updates write into buffers the solver reuses
        return x, edge_x
