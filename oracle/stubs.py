"""Test infrastructure (checker side, like everything under oracle/): deterministic stand-ins for the pieces of the reference's
conditional evaluation that are outside the hot path — the pretrained property classifier (cond_gen/, an EGNN), the dataset's
atom-count and property distributions.  Used by oracle/make_golden.py (cond_eval fixture, driving the REAL reference's
get_cond_sampling_eval_fn) and by tests/test_oracle_golden.py (driving the mirror in jodo_amd/sampling.py)."""
import torch


class StubClassifier(torch.nn.Module):
    """A deterministic stand-in for the reference's pretrained property classifier (cond_gen/: an EGNN, out of this path's scope)
    with its call signature (sampling.py:365-367): one number per molecule from the sampled atoms and positions.  Shared by the
    fixture generator and tests/test_oracle_golden.py."""

    def forward(self, h0, x, edges, edge_attr, node_mask, edge_mask, n_nodes):
        assert edge_attr is None and edges[0].numel() == (h0.shape[0] // n_nodes) * n_nodes * n_nodes
        w = torch.arange(1, h0.shape[1] + 1, dtype=h0.dtype, device=h0.device)
        per_atom = (h0 * w).sum(1, keepdim=True) * 0.1 + x.square().sum(1, keepdim=True)
        return (per_atom * node_mask).reshape(-1, n_nodes).sum(1) / node_mask.reshape(-1, n_nodes).sum(1)


class FixedNodes:
    def __init__(self, n_nodes):
        self.n = torch.as_tensor(n_nodes)

    def sample(self, k):
        assert k == self.n.numel()
        return self.n.clone()


class NormalContext:
    def sample_batch(self, n_nodes):
        return torch.randn(len(n_nodes), 1)


