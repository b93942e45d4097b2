"""Stand-in for torch_geometric.nn: only names the reference needs to be importable."""
import torch
from .conv import MessagePassing  # noqa: F401


class Linear(torch.nn.Linear):
    """PyG Linear; only imported by models/cdgs.py (off the hot path)."""


class GINEConv(torch.nn.Module):      # import-only placeholder (CDGS model, out of scope)
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("CDGS is out of scope")


class GATConv(GINEConv):
    pass
