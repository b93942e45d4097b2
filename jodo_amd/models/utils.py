"""Model registry and masked-tensor helpers: the drop-in boundary of the hot path.

Same public names, argument meaning and error behaviour as /root/reference/models/utils.py:
`register_model` / `create_model` (:2-28), `remove_mean_with_mask` (:38-45), the mask assertions
(:48-64) and the noise samplers (:67-99).  The models registered here are the MI355X-native ones
(jodo_amd/models/dgt.py), under the reference's names 'DGT_concat' and 'cond_DGT_concat'.

`create_model` differs from the reference in one deliberate way: the reference wraps the model in
`torch.nn.DataParallel` (:27, single process, per-forward parameter broadcast).  Here multi-GPU is
one process per GPU with the batch sharded by the sampler (jodo_amd/dist.py), so the model is
wrapped by default (`wrap='dataparallel_keys'`) in a thin module exposing the same `module.`-prefixed
state_dict keys so reference checkpoints load with strict=True; `wrap='dataparallel'` is the reference's
own `torch.nn.DataParallel` on the model's single device (supported: DataParallel over one device calls
the module in place); over several devices the modules raise instead of being replicated.
"""
import torch

_MODELS = {}


def register_model(cls=None, *, name=None):
    """Decorator: `@register_model` or `@register_model(name='DGT_concat')`.  Re-registering a name
    raises ValueError, as the reference does."""

    def _add(c):
        key = c.__name__ if name is None else name
        if key in _MODELS:
            raise ValueError("Already registerd model")
        _MODELS[key] = c
        return c

    return _add if cls is None else _add(cls)


def get_model_class(name):
    return _MODELS[name]


class _ModulePrefix(torch.nn.Module):
    """Gives `module.<key>` state_dict names (what a DataParallel-saved checkpoint holds) without
    DataParallel's replicate/scatter/gather."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def model_hook(model, name):
    """Optional method `name` of a score network or of the module it wraps (DataParallel-style `.module`), else None."""
    fn = getattr(model, name, None)
    if fn is None:
        fn = getattr(getattr(model, 'module', None), name, None)
    return fn


def create_model(config, wrap='dataparallel_keys'):
    model = _MODELS[config.model.name](config)
    model = model.to(config.device)
    if wrap == 'dataparallel_keys':
        model = _ModulePrefix(model)
    elif wrap == 'dataparallel':
        # exactly the reference's wrapper (models/utils.py:27), on the ONE device the model lives on: DataParallel then calls the
        # module in place (no replicas); over several devices the DGT modules refuse to be replicated (_replicate_for_data_parallel)
        dev = torch.device(config.device)
        model = torch.nn.DataParallel(model, device_ids=[dev.index if dev.index is not None else torch.cuda.current_device()])
    elif wrap is not None:
        raise ValueError("wrap in {'dataparallel_keys', 'dataparallel', None}")
    return model


# ---------------------------------------------------------------------------------------------
# masked helpers
# ---------------------------------------------------------------------------------------------
def remove_mean_with_mask(x, node_mask):
    """x [B,N,3], node_mask [B,N,1]: subtract the per-molecule mean (sum over all N rows / #real)."""
    n_real = node_mask.sum(1, keepdims=True)
    return x - torch.sum(x, dim=1, keepdim=True) / n_real * node_mask


def assert_correctly_masked(variable, node_mask):
    assert (variable * (1 - node_mask)).abs().max().item() < 1e-4, 'Variables not masked properly.'


def assert_mean_zero_with_mask(x, node_mask, eps=1e-10):
    assert_correctly_masked(x, node_mask)
    largest = x.abs().max().item()
    err = torch.sum(x, dim=1, keepdim=True).abs().max().item()
    rel = err / (largest + eps)
    assert rel < 1e-2, f'Mean is not zero, relative_error {rel}'


def sample_center_gravity_zero_gaussian_with_mask(size, device, node_mask, generator=None):
    assert len(size) == 3
    x = torch.randn(size, device=device, generator=generator) * node_mask
    return remove_mean_with_mask(x, node_mask)


def sample_gaussian_with_mask(size, device, node_mask, generator=None):
    return torch.randn(size, device=device, generator=generator) * node_mask


def sample_combined_position_feature_noise(n_samples, n_nodes, in_node_nf, node_mask, generator=None):
    """CoM-free N(0,1) for the 3 position channels, masked N(0,1) for the feature channels.
    RNG draw order (positions first, then features) matches the reference so seeded runs line up."""
    z_x = sample_center_gravity_zero_gaussian_with_mask((n_samples, n_nodes, 3), node_mask.device, node_mask,
                                                        generator)
    z_h = sample_gaussian_with_mask((n_samples, n_nodes, in_node_nf), node_mask.device, node_mask, generator)
    return torch.cat([z_x, z_h], dim=2)


def sample_symmetric_edge_feature_noise(n_samples, n_nodes, edge_ch, edge_mask, generator=None):
    """Symmetric edge noise: draw [B,ch,N,N], keep the strict lower triangle, mirror it, mask."""
    z = torch.randn((n_samples, edge_ch, n_nodes, n_nodes), device=edge_mask.device, generator=generator)
    z = torch.tril(z, -1)
    z = z + z.transpose(-1, -2)
    return z.permute(0, 2, 3, 1) * edge_mask.reshape(n_samples, n_nodes, n_nodes, 1)


def coord2dist(x, edge_index):
    row, col = edge_index
    d = x[row] - x[col]
    return torch.sum(d ** 2, 1).unsqueeze(1)
