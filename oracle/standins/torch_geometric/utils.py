"""Pure-torch restatements of torch_geometric.utils.{softmax,dense_to_sparse} (PyG 2.1 semantics).

softmax(src, index, ptr, num_nodes): per-segment softmax over rows sharing `index`:
    m = segment_max(src)[index]; e = exp(src - m); e / (segment_sum(e)[index] + 1e-16)
dense_to_sparse(adj) for a 3-D [B,N,N] tensor: nonzero entries in row-major order,
    row = b*N + i, col = b*N + j, plus the edge values.
"""
import torch


def softmax(src, index=None, ptr=None, num_nodes=None, dim=0):
    assert ptr is None and dim == 0
    n = int(index.max()) + 1 if num_nodes is None else int(num_nodes)
    shape = (n,) + tuple(src.shape[1:])
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    seg_max = torch.full(shape, float('-inf'), dtype=src.dtype, device=src.device)
    seg_max = seg_max.scatter_reduce(0, idx, src, reduce='amax', include_self=True)
    out = (src - seg_max.gather(0, idx)).exp()
    seg_sum = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(0, idx, out)
    return out / (seg_sum.gather(0, idx) + 1e-16)


def dense_to_sparse(adj):
    assert adj.dim() in (2, 3)
    if adj.dim() == 2:
        nz = adj.nonzero().t()
        return nz, adj[nz[0], nz[1]]
    nz = adj.nonzero().t()
    n = adj.size(-1)
    row = nz[1] + nz[0] * n
    col = nz[2] + nz[0] * n
    return torch.stack([row, col], dim=0), adj[nz[0], nz[1], nz[2]]
