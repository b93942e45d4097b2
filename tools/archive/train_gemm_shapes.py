"""Which products a training forward + backward issues, and what each shape costs in isolation:
    JODO_TRAIN_GEMM_LOG=/tmp/g.log python tools/train_gemm_shapes.py [--workload qm9]
(the library appends one line per jodo_train gemm() call to the log; this script runs ONE grad-enabled forward + backward of the
config's training batch, groups the log by (layout, M, N, K, leading dimensions) and times every distinct shape with jodo_train_gemm)."""
import argparse, collections, ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='qm9')
args = ap.parse_args()
log = os.environ.setdefault('JODO_TRAIN_GEMM_LOG', '/tmp/jodo_gemm_shapes.log')
if os.path.exists(log):
    os.remove(log)
from jodo_amd import capi, configs
from jodo_amd.models import get_model_class, deterministic_init_, load_dataset_info, get_node_dist
from jodo_amd.sampling import build_masks
name, info = dict(qm9=('vpsde_qm9_uncond_jodo', 'qm9_with_h'), geom=('vpsde_geom_uncond_jodo', 'geom_with_h_1'))[args.workload]
cfg = configs.get(name)
dev = torch.device('cuda:0')
cfg.device = dev
torch.manual_seed(42)
B = int(cfg.training.batch_size)
n_nodes = get_node_dist(load_dataset_info(info)).sample(B).tolist()
model = deterministic_init_(get_model_class(cfg.model.name)(cfg), seed=42).to(dev)
N = max(n_nodes)
nm, em = build_masks(n_nodes, N, dev)
xh = torch.randn(B, N, 3 + model.dims.nd, device=dev) * nm
ex = torch.randn(B, N, N, model.dims.ch, device=dev)
ex = (ex + ex.transpose(1, 2)) * em.reshape(B, N, N, 1)
nl = torch.randn(B, device=dev)
ox, oe = model(nl, xh, nm, em, edge_x=ex, cond_x=None, cond_edge_x=None, noise_level=nl)
torch.cuda.synchronize()
n_fwd = sum(1 for _ in open(log))
(ox.square().sum() + oe.square().sum()).backward()
torch.cuda.synchronize()
lines = [tuple(int(v) for v in l.split()) for l in open(log)]
os.environ.pop('JODO_TRAIN_GEMM_LOG')
count = collections.Counter(lines)
fwd_count = collections.Counter(lines[:n_fwd])
L = capi.lib()
ws = torch.empty(64 << 20, device=dev)
rows = []
for (tA, tB, M, Nc, K, lda, ldb, ldc), c in count.items():
    A = torch.randn((K if tA else M), lda, device=dev)
    Bm = torch.randn((Nc if tB else K), ldb, device=dev)
    C = torch.empty(M, ldc, device=dev)
    call = lambda: capi.check(L.jodo_train_gemm(tA, tB, M, Nc, K, capi.ptr(A), lda, capi.ptr(Bm), ldb, capi.ptr(C), ldc, None, 0, capi.ptr(ws),
                                                 ctypes.c_size_t(ws.numel()), capi.current_stream_ptr()), 'gemm')
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    rows.append((c * dt, c, fwd_count[(tA, tB, M, Nc, K, lda, ldb, ldc)], tA, tB, M, Nc, K, lda, ldb, ldc, dt))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('%d products (%d forward), %d shapes, %.2f ms in isolation' % (len(lines), n_fwd, len(rows), tot * 1e3))
for t, c, cf, tA, tB, M, Nc, K, lda, ldb, ldc, dt in rows[:40]:
    print('%5.2f ms %4.1f%%  x%3d (fwd %3d)  tA %d tB %d  M %6d N %5d K %6d  ld %5d %5d %5d  %7.1f us  %5.1f TF  %5.2f TB/s' % (
        t * 1e3, 100 * t / tot, c, cf, tA, tB, M, Nc, K, lda, ldb, ldc, dt * 1e6, 2.0 * M * Nc * K / dt / 1e12, 4.0 * (M * K + K * Nc + M * Nc) / dt / 1e12))
