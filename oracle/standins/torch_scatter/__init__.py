"""Pure-torch restatement of torch_scatter.scatter (2.0.9 semantics) for reduce in {'add','sum'}."""
import torch


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce='sum'):
    assert reduce in ('add', 'sum') and out is None
    dim = dim if dim >= 0 else src.dim() + dim
    if dim_size is None:
        dim_size = int(index.max()) + 1
    shape = list(src.shape)
    shape[dim] = dim_size
    view = [1] * src.dim()
    view[dim] = -1
    idx = index.view(view).expand_as(src)
    return torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(dim, idx, src)
