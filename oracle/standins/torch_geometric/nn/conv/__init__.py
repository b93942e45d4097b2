"""Pure-torch restatement of torch_geometric.nn.conv.MessagePassing.propagate (PyG 2.1 semantics,
flow='source_to_target', aggr='add', node_dim=0), just enough for the reference's attention layers.

For every parameter of `message` named `X_i` / `X_j` the tensor kwargs['X'] is gathered along dim 0
with edge_index[1] (targets) / edge_index[0] (sources); `index` = edge_index[1], `ptr` = None,
`size_i` = number of nodes; other parameters are passed through.  The result of `message` is
summed into its target node (scatter-add over edge_index[1]); `update` is the identity.
"""
import inspect
import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr='add', flow='source_to_target', node_dim=0, **kwargs):
        super().__init__()
        assert aggr == 'add' and flow == 'source_to_target' and node_dim == 0
        self._msg_params = None

    def propagate(self, edge_index, size=None, **kwargs):
        if self._msg_params is None:
            self._msg_params = list(inspect.signature(self.message).parameters)
        n_nodes = None
        for v in kwargs.values():
            if torch.is_tensor(v) and v.dim() > 0:
                n_nodes = v.size(0)
                break
        src, dst = edge_index[0], edge_index[1]
        args = {}
        for name in self._msg_params:
            if name == 'size_i':
                args[name] = None          # filled in below
            elif name.endswith('_i'):
                args[name] = kwargs[name[:-2]].index_select(0, dst)
                n_nodes = kwargs[name[:-2]].size(0)
            elif name.endswith('_j'):
                args[name] = kwargs[name[:-2]].index_select(0, src)
            elif name == 'index':
                args[name] = dst
            elif name == 'ptr':
                args[name] = None
            else:
                args[name] = kwargs[name]
        if 'size_i' in args:
            args['size_i'] = n_nodes
        msg = self.message(**args)
        out = torch.zeros((n_nodes,) + tuple(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
        out.index_add_(0, dst, msg)
        return out
