"""TEST INFRASTRUCTURE (checker only; never imported by the product path).

CPU restatements for the training-side slice:

  sde_graph_loss        the data-prediction loss of get_sde_graph_loss_fn, /root/reference/losses.py:350-385, given the
                        model's prediction and the targets the reference built (scaled data, Kabsch-aligned positions)
  edge_ffn_phase        phase D of EquivariantMixBlock.forward, /root/reference/models/mol_gnn.py:313-317: edge residual with
                        the message gate, LayerNorm2 + modulate, edge FFN with its gate — the two lines of
                        oracle/dgt_oracle.forward_dense that produce `en` and `e`, as a function of its inputs, so that
                        torch.autograd gives the local vector-Jacobian products the HIP backward kernel is checked against

Pinned by tests/golden/grad_qm9.npz: the reference's own loss value and `loss.backward()` gradients (oracle/make_golden.py
grad_fixture) are reproduced by autograd through oracle.dgt_oracle.forward_dense + sde_graph_loss
(tests/test_oracle_golden.py::test_oracle_gradients_match_reference_backward).
"""
import torch
from torch.nn import functional as F


def sde_graph_loss(pred, edge_pred, xh, edge_x, align_pos, node_mask, edge_mask, alpha_t, sigma_t, loss_weights=(1., 0.25, 0.1),
                   reduce_mean=False):
    """losses.py:350-385 (pred_data = True): per-molecule squared errors of positions (against the aligned target), atom
    features and edge features, weighted, scaled by sqrt(alpha_t / sigma_t), mean over the batch."""
    B = xh.shape[0]
    l_pos = torch.square(pred[:, :, :3] - align_pos).mean(-1).sum(-1)
    l_atom = torch.square(pred[:, :, 3:] - xh[:, :, 3:]).mean(-1).sum(-1)
    l_edge = torch.square(edge_x - edge_pred).mean(-1).reshape(B, -1).sum(-1)
    if reduce_mean:
        n_nodes = node_mask.reshape(B, -1).sum(-1)
        l_pos, l_atom = l_pos / n_nodes, l_atom / n_nodes
        l_edge = l_edge / (edge_mask.reshape(B, -1).sum(-1) + 1e-8)
    losses = loss_weights[0] * l_pos + loss_weights[1] * l_atom + loss_weights[2] * l_edge
    return (torch.sqrt(alpha_t / sigma_t) * losses).mean()


def _ln(x, eps=1e-6):
    return F.layer_norm(x, x.shape[-1:], eps=eps)


def edge_ffn_phase(e_in, ehat, eg1, es2, ec2, eg2, W3, b3, W4, b4):
    """mol_gnn.py:313-317 with cond_time: h_edge = h_in_edge + edge_gate_msa * node2edge(h_i + h_j);
    h_edge = modulate(norm2_edge(h_edge), shift_mlp, scale_mlp); out = h_edge + edge_gate_mlp * ff4(SiLU(ff3(h_edge))).
    e_in, ehat [R, De]; the four modulation vectors broadcast against them ([R, De] or [1, De])."""
    en = _ln(e_in + eg1 * ehat) * (1 + ec2) + es2
    return en + eg2 * F.linear(F.silu(F.linear(en, W3, b3)), W4, b4)
