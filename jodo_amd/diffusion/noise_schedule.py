"""VP noise schedule used by the samplers (host-side scalar math, stays Python).

Mirrors the behaviour of /root/reference/diffusion/noise_schedule.py:6-122 (`NoiseScheduleVP`) for the
continuous 'linear' and 'cosine' schedules (every JODO config on the hot path uses 'cosine').
The discrete schedules ('discrete' is broken upstream — undefined name at :30 — and 'discrete_poly'
is not used by any in-scope config) are rejected loudly.
"""
import math

import torch


class NoiseScheduleVP:
    def __init__(self, schedule='cosine', betas=None, alphas_cumprod=None, continuous_beta_0=0.1,
                 continuous_beta_1=20., dtype=torch.float32):
        if schedule not in ('linear', 'cosine'):
            raise ValueError("Unsupported noise schedule {} (hot path supports 'linear', 'cosine')"
                             .format(schedule))
        self.schedule = schedule
        self.total_N = 1000
        self.beta_0 = continuous_beta_0
        self.beta_1 = continuous_beta_1
        self.cosine_s = 0.008
        self.cosine_beta_max = 999.
        self.cosine_t_max = (math.atan(self.cosine_beta_max * (1. + self.cosine_s) / math.pi) * 2.
                             * (1. + self.cosine_s) / math.pi - self.cosine_s)
        self.cosine_log_alpha_0 = math.log(math.cos(self.cosine_s / (1. + self.cosine_s) * math.pi / 2.))
        # T = 1 is numerically unstable for the cosine schedule; the reference fixes 0.9946 (:51).
        self.T = 0.9946 if schedule == 'cosine' else 1.

    def marginal_log_mean_coeff(self, t):
        """log(alpha_t)"""
        if self.schedule == 'linear':
            return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        ang = (t + self.cosine_s) / (1. + self.cosine_s) * math.pi / 2.
        return torch.log(torch.cos(ang)) - self.cosine_log_alpha_0

    def marginal_alpha(self, t):
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return torch.sqrt(1. - torch.exp(2. * self.marginal_log_mean_coeff(t)))

    def marginal_prob(self, t):
        la = self.marginal_log_mean_coeff(t)
        return torch.exp(la), torch.sqrt(1. - torch.exp(2. * la))

    def marginal_lambda(self, t):
        """half log-SNR: log(alpha_t) - log(sigma_t)"""
        la = self.marginal_log_mean_coeff(t)
        return la - 0.5 * torch.log(1. - torch.exp(2. * la))

    def inverse_lambda(self, lamb):
        if self.schedule == 'linear':
            tmp = 2. * (self.beta_1 - self.beta_0) * torch.logaddexp(-2. * lamb, torch.zeros((1,)).to(lamb))
            delta = self.beta_0 ** 2 + tmp
            return tmp / (torch.sqrt(delta) + self.beta_0) / (self.beta_1 - self.beta_0)
        la = -0.5 * torch.logaddexp(-2. * lamb, torch.zeros((1,)).to(lamb))
        return (torch.arccos(torch.exp(la + self.cosine_log_alpha_0)) * 2. * (1. + self.cosine_s) / math.pi
                - self.cosine_s)

    def get_noiseLevel(self, t):
        a, s = self.marginal_alpha(t), self.marginal_std(t)
        return torch.log(a ** 2 / s ** 2)
