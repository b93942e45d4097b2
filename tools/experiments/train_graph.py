"""Does replaying the training forward + backward as a HIP graph shorten it?  (2 200 launches of ~11 us)   python tools/experiments/train_graph.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jodo_amd import configs
from jodo_amd.models import get_model_class, deterministic_init_, load_dataset_info, get_node_dist
from jodo_amd.sampling import build_masks
from jodo_amd.train import TrainEngine

cfg = configs.get('vpsde_qm9_uncond_jodo')
dev = torch.device('cuda:0'); cfg.device = dev
torch.manual_seed(42)
B = int(sys.argv[1]) if len(sys.argv) > 1 else int(cfg.training.batch_size)
n_nodes = get_node_dist(load_dataset_info('qm9_with_h')).sample(B).tolist()
model = deterministic_init_(get_model_class(cfg.model.name)(cfg), seed=42).to(dev)
N = max(n_nodes)
nm, em = build_masks(n_nodes, N, dev)
xh = torch.randn(B, N, 3 + model.dims.nd, device=dev) * nm
ex = torch.randn(B, N, N, model.dims.ch, device=dev); ex = (ex + ex.transpose(1, 2)) * em.reshape(B, N, N, 1)
nl = torch.randn(B, device=dev)
named = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
eng = TrainEngine(model._cfg(), n_nodes, N, named, dev)
params = [v.detach().float().contiguous() for v in model.state_dict().values()]
dx, de = torch.randn_like(xh), torch.randn_like(ex)

def step():
    ox, oe = eng.forward(params, xh, ex, None, None, nl, None, 0.1, 7)
    return ox, oe, eng.backward(params, nl, dx, de, 0.1, 7)

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print('eager forward + backward: %.2f ms' % timeit(step))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
print('graph replay:             %.2f ms' % timeit(g.replay))
# (timing experiment only: the gradients of a replay are NOT validated — TrainEngine.backward allocates its gradient buffer per call,
# which a capture turns into graph-pool memory; result on MI355X: 22.2 ms eager, 22.2 ms replayed — the step is bound by kernel time)
