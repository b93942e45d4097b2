"""Machine-independent deterministic weight initialisation.

Trained JODO checkpoints are external downloads and unavailable offline, so tests and benchmarks
need identical random weights in the build container (where the real reference is initialised with
them to produce fixtures) and on the GPU box (where our module is).  torch's default init depends on
construction order and generator state; this one depends only on (seed, parameter name, shape):
numpy PCG64 streams keyed by crc32(name).  Distributions follow torch's defaults for each layer kind
(U(+-1/sqrt(fan_in)) for Linear weight and bias, U(0,3) for the GBF means/stds, N(0,1) for the
sinusoid frequencies); `gain` scales weight matrices to push activations out of the linear regime.
"""
import zlib

import numpy as np
import torch


@torch.no_grad()
def deterministic_init_(module, seed=0, gain=1.0, coord_scale=None):
    sd = module.state_dict()
    for name, t in sd.items():
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
        shape = tuple(t.shape)
        if name.endswith('means.weight') or name.endswith('stds.weight'):
            v = rng.uniform(0.0, 3.0, size=shape)
        elif name.endswith('coord_norm.scale'):
            v = np.full(shape, 1e-2 if coord_scale is None else coord_scale)
        elif name.endswith('.weights'):                       # learned sinusoid frequencies
            v = rng.standard_normal(size=shape)
        elif name.endswith('.weight'):
            bound = 1.0 / np.sqrt(shape[1])
            v = rng.uniform(-bound, bound, size=shape) * gain
        elif name.endswith('.bias'):
            w = sd[name[:-4] + 'weight']
            bound = 1.0 / np.sqrt(w.shape[1])
            v = rng.uniform(-bound, bound, size=shape)
        else:
            raise KeyError("unexpected parameter " + name)
        t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
    return module
