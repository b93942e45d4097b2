"""Hybrid DPM-Solver++ sampler (BASELINE config 5): atom-type / charge / bond channels follow
DPM-Solver++ data-prediction updates of order 1-3 (single-step) or 2 (multi-step); positions always
take an ancestral (stochastic) step.  Self-conditioning state is carried between model calls.

Behaviour of /root/reference/mix_dpm_solver.py: `DPM_Solver_hybrid.sampling` (:304-376), position
update (:44-59), first/second/third-order single-step updates (:61-227), multistep second order
(:230-265), `get_model_fn` (:296-302).  The score-network call signature is unchanged.
`steps` counts model evaluations (NFE): K = steps // order outer steps of `order` evaluations each.
"""
import torch

from .models.utils import assert_mean_zero_with_mask, sample_center_gravity_zero_gaussian_with_mask


def split_x(x):
    return x[:, :, :3], x[:, :, 3:]


def merge_x(pos, feat):
    return torch.cat([pos, feat], dim=-1)


class DPM_Solver_hybrid:
    def __init__(self, noise_schedule, config, noise_fn=None):
        self.noise_schedule = noise_schedule
        self.cond_x = None
        self.cond_edge_x = None
        self.order = config.sampling.dpm_solver_order
        self.steps = config.sampling.steps
        self.method = config.sampling.dpm_solver_method
        self.noise_fn = noise_fn          # optional replay hook: noise_fn(call_index, 'pos', like) -> tensor
        self._noise_calls = 0
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
    def ancestral_position_update(self, position_x, position_pred, node_mask, t_start, t_end, last_step=False):
        ns = self.noise_schedule
        alpha_t, sigma_t = ns.marginal_prob(t_start)
        alpha_s, sigma_s = ns.marginal_prob(t_end)
        a_ts = alpha_t / alpha_s
        var_ts = sigma_t ** 2 - a_ts ** 2 * sigma_s ** 2
        sigma = torch.sqrt(var_ts) * sigma_s / sigma_t
        position = (a_ts * sigma_s ** 2 / sigma_t ** 2) * position_x + (alpha_s * var_ts / sigma_t ** 2) * position_pred
        if not last_step:
            if self.noise_fn is not None:
                eps = self.noise_fn(self._noise_calls, 'pos', position_x)
            else:
                eps = sample_center_gravity_zero_gaussian_with_mask(position_x.size(), position_x.device, node_mask)
            self._noise_calls += 1
            position = position + sigma * eps
        return position

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
        pos0, atom0 = split_x(x)
        if pred_start is None and edge_pred_start is None:
            pred_start, edge_pred_start = self._predict(model_fn, x, node_mask, edge_mask, edge_x, context, t_start)
        pos_pred0, atom_pred0 = split_x(pred_start)
        atom_end = sigma_end / sigma_start * atom0 - alpha_end * phi_1 * atom_pred0
        edge_end = sigma_end / sigma_start * edge_x - alpha_end * phi_1 * edge_pred_start
        pos_end = self.ancestral_position_update(pos0, pos_pred0, node_mask, t_start, t_end, last_step)
        return merge_x(pos_end, atom_end), edge_end

    # -- order 2, single step ------------------------------------------------------------------
    def singlestep_dpm_solver_second_update(self, model_fn, x, node_mask, edge_mask, edge_x, context,
                                            t_start, t_end, last_step, r1=0.5):
        r1 = 0.5 if r1 is None else r1
        ns = self.noise_schedule
        lam0, lam1 = ns.marginal_lambda(t_start), ns.marginal_lambda(t_end)
        h = lam1 - lam0
        s1 = ns.inverse_lambda(lam0 + r1 * h).reshape(())     # scalar (0-dim): broadcasts across devices
        sigma_start, sigma_s1, sigma_end = ns.marginal_std(t_start), ns.marginal_std(s1), ns.marginal_std(t_end)
        alpha_s1 = torch.exp(ns.marginal_log_mean_coeff(s1))
        alpha_end = torch.exp(ns.marginal_log_mean_coeff(t_end))
        phi_11 = torch.expm1(-r1 * h)
        phi_1 = torch.expm1(-h)
        pos0, atom0 = split_x(x)

        pred0, edge_pred0 = self._predict(model_fn, x, node_mask, edge_mask, edge_x, context, t_start)
        pos_pred0, atom_pred0 = split_x(pred0)
        atom_s1 = (sigma_s1 / sigma_start) * atom0 - (alpha_s1 * phi_11) * atom_pred0
        edge_s1 = (sigma_s1 / sigma_start) * edge_x - (alpha_s1 * phi_11) * edge_pred0
        pos_s1 = self.ancestral_position_update(pos0, pos_pred0, node_mask, t_start, s1)
        x_s1 = merge_x(pos_s1, atom_s1)

        pred1, edge_pred1 = self._predict(model_fn, x_s1, node_mask, edge_mask, edge_s1, context, s1)
        pos_pred1, atom_pred1 = split_x(pred1)
        atom_end = ((sigma_end / sigma_start) * atom0 - (alpha_end * phi_1) * atom_pred0
                    - (0.5 / r1) * (alpha_end * phi_1) * (atom_pred1 - atom_pred0))
        edge_end = ((sigma_end / sigma_start) * edge_x - (alpha_end * phi_1) * edge_pred0
                    - (0.5 / r1) * (alpha_end * phi_1) * (edge_pred1 - edge_pred0))
        pos_end = self.ancestral_position_update(pos_s1, pos_pred1, node_mask, s1, t_end, last_step)
        return merge_x(pos_end, atom_end), edge_end

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
        pos0, atom0 = split_x(x)

        pred0, edge_pred0 = self._predict(model_fn, x, node_mask, edge_mask, edge_x, context, t_start)
        pos_pred0, atom_pred0 = split_x(pred0)
        atom_s1 = (sigma_s1 / sigma_start) * atom0 - (alpha_s1 * phi_11) * atom_pred0
        edge_s1 = (sigma_s1 / sigma_start) * edge_x - (alpha_s1 * phi_11) * edge_pred0
        pos_s1 = self.ancestral_position_update(pos0, pos_pred0, node_mask, t_start, s1)
        x_s1 = merge_x(pos_s1, atom_s1)

        pred1, edge_pred1 = self._predict(model_fn, x_s1, node_mask, edge_mask, edge_s1, context, s1)
        pos_pred1, atom_pred1 = split_x(pred1)
        atom_s2 = ((sigma_s2 / sigma_start) * atom0 - (alpha_s2 * phi_12) * atom_pred0
                   + r2 / r1 * (alpha_s2 * phi_22) * (atom_pred1 - atom_pred0))
        edge_s2 = ((sigma_s2 / sigma_start) * edge_x - (alpha_s2 * phi_12) * edge_pred0
                   + r2 / r1 * (alpha_s2 * phi_22) * (edge_pred1 - edge_pred0))
        pos_s2 = self.ancestral_position_update(pos_s1, pos_pred1, node_mask, s1, s2)
        x_s2 = merge_x(pos_s2, atom_s2)

        pred2, edge_pred2 = self._predict(model_fn, x_s2, node_mask, edge_mask, edge_s2, context, s2)
        pos_pred2, atom_pred2 = split_x(pred2)
        atom_end = ((sigma_end / sigma_start) * atom0 - (alpha_end * phi_1) * atom_pred0
                    + (1. / r2) * (alpha_end * phi_2) * (atom_pred2 - atom_pred0))
        edge_end = ((sigma_end / sigma_start) * edge_x - (alpha_end * phi_1) * edge_pred0
                    + (1. / r2) * (alpha_end * phi_2) * (edge_pred2 - edge_pred0))
        pos_end = self.ancestral_position_update(pos_s2, pos_pred2, node_mask, s2, t_end, last_step)
        return merge_x(pos_end, atom_end), edge_end

    # -- order 2, multistep --------------------------------------------------------------------
    def multistep_dpm_solver_second_update(self, model_fn, x, node_mask, edge_mask, edge_x, context,
                                           model_prev_list, t_prev_list, t, last_step):
        ns = self.noise_schedule
        (pred_m1, edge_pred_m1), (pred_m0, edge_pred_m0) = model_prev_list[-2], model_prev_list[-1]
        _, atom_pred_m1 = split_x(pred_m1)
        pos_pred_m0, atom_pred_m0 = split_x(pred_m0)
        pos_m0, atom_m0 = split_x(x)
        t_m1, t_m0 = t_prev_list[-2], t_prev_list[-1]
        lam_m1, lam_m0, lam_t = ns.marginal_lambda(t_m1), ns.marginal_lambda(t_m0), ns.marginal_lambda(t)
        sigma_m0, sigma_t = ns.marginal_std(t_m0), ns.marginal_std(t)
        alpha_t = torch.exp(ns.marginal_log_mean_coeff(t))
        h_0 = lam_m0 - lam_m1
        h = lam_t - lam_m0
        r0 = h_0 / h
        phi_1 = torch.expm1(-h)
        d_atom = (1. / r0) * (atom_pred_m0 - atom_pred_m1)
        d_edge = (1. / r0) * (edge_pred_m0 - edge_pred_m1)
        atom_t = (sigma_t / sigma_m0) * atom_m0 - (alpha_t * phi_1) * atom_pred_m0 - 0.5 * (alpha_t * phi_1) * d_atom
        edge_t = (sigma_t / sigma_m0) * edge_x - (alpha_t * phi_1) * edge_pred_m0 - 0.5 * (alpha_t * phi_1) * d_edge
        pos_t = self.ancestral_position_update(pos_m0, pos_pred_m0, node_mask, t_prev_list[-1], t, last_step)
        return merge_x(pos_t, atom_t), edge_t

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
            self.cond_x, self.cond_edge_x = pred_t, edge_pred_t
            return pred_t, edge_pred_t
        return model_fn

    @torch.no_grad()
    def sampling(self, model, x, node_mask, edge_mask, edge_x, context=None, t_start=None, t_end=None,
                 skip_type='time_uniform'):
        steps, order = self.steps, self.order
        self.cond_x = self.cond_edge_x = None
        self._noise_calls = 0
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
        return x, edge_x
