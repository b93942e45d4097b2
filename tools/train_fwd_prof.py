"""Only grad-enabled forwards of the training path (QM9 config batch), for `rocprofv3 --kernel-trace --stats`: which launches the
forward's time is in.   (cd /tmp && rocprofv3 --kernel-trace --stats -d OUT -- python tools/train_fwd_prof.py [reps] [fused 0|1])"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jodo_amd import configs
from jodo_amd.models import get_model_class, deterministic_init_, load_dataset_info, get_node_dist
from jodo_amd.sampling import build_masks

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1
with_bwd = len(sys.argv) > 3 and sys.argv[3] == 'bwd'        # also run loss.backward(): the kernel table then holds forward + backward
cfg = configs.get('vpsde_qm9_uncond_jodo')
dev = torch.device('cuda:0')
cfg.device = dev
B = int(os.environ.get("JODO_PROF_BATCH", cfg.training.batch_size))
torch.manual_seed(42)
n_nodes = get_node_dist(load_dataset_info('qm9_with_h')).sample(B).tolist()
model = deterministic_init_(get_model_class(cfg.model.name)(cfg), seed=42).to(dev)
model.train_options = {0: fused}
N = max(n_nodes)
nm, em = build_masks(n_nodes, N, dev)
xh = torch.randn(B, N, 3 + model.dims.nd, device=dev) * nm
ex = torch.randn(B, N, N, model.dims.ch, device=dev)
ex = (ex + ex.transpose(1, 2)) * em.reshape(B, N, N, 1)
nl = torch.randn(B, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for i in range(reps + 1):
    if i == 1:
        torch.cuda.synchronize(); ev[0].record()
    ox, oe = model(nl, xh, nm, em, edge_x=ex, cond_x=None, cond_edge_x=None, noise_level=nl)
    if with_bwd:
        model.zero_grad()
        (ox.square().sum() + oe.square().sum()).backward()
ev[1].record(); torch.cuda.synchronize()
print('%s ms (fused=%d): %.3f' % ('forward + backward' if with_bwd else 'forward', fused, ev[0].elapsed_time(ev[1]) / reps))
