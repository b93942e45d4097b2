"""Which host call enqueues the bursts of __amd_rocclr_copyBuffer seen behind a training forward?  Variants of one forward under rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from jodo_amd import configs
from jodo_amd.models import get_model_class, deterministic_init_, load_dataset_info, get_node_dist
from jodo_amd.sampling import build_masks
mode = sys.argv[1]
cfg = configs.get('vpsde_qm9_uncond_jodo'); dev = torch.device('cuda:0'); cfg.device = dev
B = 32
torch.manual_seed(42)
n_nodes = get_node_dist(load_dataset_info('qm9_with_h')).sample(B).tolist()
model = deterministic_init_(get_model_class(cfg.model.name)(cfg), seed=42).to(dev)
N = max(n_nodes); nm, em = build_masks(n_nodes, N, dev)
xh = torch.randn(B, N, 3 + model.dims.nd, device=dev) * nm
ex = torch.randn(B, N, N, model.dims.ch, device=dev); ex = (ex + ex.transpose(1, 2)) * em.reshape(B, N, N, 1)
nl = torch.randn(B, device=dev)
model.train()
for i in range(4):
    if mode == 'nograd':
        with torch.no_grad():
            model(nl, xh, nm, em, edge_x=ex, cond_x=None, cond_edge_x=None, noise_level=nl)
    elif mode == 'grad_nobwd':
        model(nl, xh, nm, em, edge_x=ex, cond_x=None, cond_edge_x=None, noise_level=nl)
    else:
        ox, oe = model(nl, xh, nm, em, edge_x=ex, cond_x=None, cond_edge_x=None, noise_level=nl)
        model.zero_grad()
        (ox.sum() + oe.sum()).backward()
torch.cuda.synchronize()
