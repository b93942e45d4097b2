"""Exponential moving average of model parameters — the checkpoint-side contract of
/root/reference/models/ema.py:4-85.

Only what sampling needs is reproduced: the reference's evaluation path builds an EMA over
`model.parameters()`, loads `loaded_state['ema']` (a dict with `decay`, `num_updates` and
`shadow_params`, a *positional* list in parameter-registration order) and calls `copy_to` before
sampling (run_lib.py evaluation flow).  Because the shadow list is positional, the HIP-backed modules
keep the reference's parameter order (tests/test_host_logic.py checks it name by name).
"""
import torch


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=True):
        if not 0.0 <= decay <= 1.0:
            raise ValueError('Decay must be between 0 and 1')
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.detach().clone() for p in parameters if p.requires_grad]
        self.collected_params = []

    @staticmethod
    def _trainable(parameters):
        return [p for p in parameters if p.requires_grad]

    def update(self, parameters):
        """shadow <- shadow - (1 - d) (shadow - p), d warmed up as min(decay, (1 + k) / (10 + k))."""
        d = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            d = min(d, (1 + self.num_updates) / (10 + self.num_updates))
        # the reference's per-tensor `s.sub_((1 - d) * (s - p))` (models/ema.py:38-40) as three multi-tensor launches instead of
        # 3 x 351: the same three roundings per element, bit-identical (tests/test_losses_host.py), 1 050 launches fewer per step.
        # Parameters that are slices of one flat buffer (jodo_amd/optim.py FlatAdam): the shadow is flattened the same way once and
        # the three operations run on the two flat tensors — no per-tensor host work at all (same roundings again).
        with torch.no_grad():
            params = self._trainable(parameters)
            if len(params) != len(self.shadow_params):
                raise ValueError(f'EMA holds {len(self.shadow_params)} tensors, the model has {len(params)} trainable parameters')
            flat_p = self._flat_params(params)
            if flat_p is not None:
                delta = self._flat_shadow - flat_p
                delta.mul_(1.0 - d)
                self._flat_shadow.sub_(delta)
                return
            delta = torch._foreach_sub(self.shadow_params, params)
            torch._foreach_mul_(delta, 1.0 - d)
            torch._foreach_sub_(self.shadow_params, delta)

    def _flat_params(self, params):
        """The parameters' flat buffer when they have one AND the shadow could be flattened to match it (done on first use)."""
        if not params or not params[0].is_cuda:
            return None
        from ..optim import flat_parameters, flatten_parameters, flat_total
        total = getattr(self, '_total', None)
        if total is None:
            total = self._total = flat_total(params)
        flat_p = flat_parameters(params, total)
        if flat_p is None:
            return None
        if getattr(self, '_flat_shadow', None) is None or self._flat_shadow_of is not self.shadow_params:
            if any(s.shape != p.shape or s.device != p.device or s.dtype != torch.float32 for s, p in zip(self.shadow_params, params)):
                return None
            self._flat_shadow = flatten_parameters(self.shadow_params)       # (shadow tensors are plain tensors: `.data` re-pointing works on them too)
            self._flat_shadow_of = self.shadow_params
        return flat_p

    def copy_to(self, parameters):
        params = self._trainable(parameters)
        if len(params) != len(self.shadow_params):
            raise ValueError(f'EMA holds {len(self.shadow_params)} tensors, the model has {len(params)} trainable parameters')
        with torch.no_grad():
            for s, p in zip(self.shadow_params, params):
                if s.shape != p.shape:
                    raise ValueError(f'EMA tensor of shape {tuple(s.shape)} does not fit parameter of shape {tuple(p.shape)}')
                p.copy_(s.to(p.device))            # in-place: bumps the version the packed-weight cache keys on

    def store(self, parameters):
        self.collected_params = [p.detach().clone() for p in parameters]

    def restore(self, parameters):
        with torch.no_grad():
            for c, p in zip(self.collected_params, parameters):
                p.copy_(c)

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

    def load_state_dict(self, state_dict):
        self.decay = state_dict['decay']
        self.num_updates = state_dict['num_updates']
        self.shadow_params = state_dict['shadow_params']
