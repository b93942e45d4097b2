"""Phase-cycle breakdown of k_edge_update_sym (needs a build with -DJODO_PHASE_TIMING)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from helpers import *      # noqa
from jodo_amd import capi
from jodo_amd.models import load_dataset_info, get_node_dist
cfg = make_config('vpsde_qm9_uncond_jodo')
model = make_model(cfg, 8, 'cuda:0')
hp = O.Hyper.from_config(cfg)
torch.manual_seed(42)
n_nodes = get_node_dist(load_dataset_info('qm9_with_h')).sample(2500).tolist()
xh, ex, nl, ctx, nm, em = random_inputs(hp, n_nodes, seed=7)
nl[:] = 0.5
d = lambda x: x.cuda()
args = (d(nl), d(xh), d(nm), d(em))
kw = dict(edge_x=d(ex), cond_x=None, cond_edge_x=None, noise_level=d(nl))
SPLIT = 'split' in sys.argv[2:]                      # attn split: the two-launch split attention (model.split_bf16 = 'attention', pinned paths)
with torch.no_grad():
    if SPLIT:
        model.split_bf16 = 'attention'
    model(*args, **kw)
    if SPLIT:
        model.pin_paths()
        model(*args, **kw)
    buf = torch.zeros(16, dtype=torch.int64, device='cuda')
    L = capi.lib()
    capi.check(L.jodo_debug_set_timing_buffer(model._last_plan['handle'], capi.ptr(buf)), 'set')
    model(*args, **kw)
    torch.cuda.synchronize()
b = buf.cpu().tolist()
n = max(b[15], 1)
names = ['0 top + LN(en)', '1 FFN', '2 readout', '3 input_lin S: MFMA blocks', '4 LN statistics riding on S', '5 folded coord_mlp.0 (Z): MFMA blocks', '6 item end', '7 SiLU / coord_mlp.2 tails riding on Z']   # hoisted kernel
if len(sys.argv) > 1 and sys.argv[1] == 'attn':      # build with -DJODO_PHASE_TIMING_ATTN; counts are per pair offset
    names = ['0 item prologue (weights to LDS, own rows)', '1 edge input: GBF, edge_emb, LN', '2 scores: lin_edge0, tanh, q.k', '3 hand-over barrier + read', '4 softmax update', '5 messages: lin_edge1, tanh, v, hand-over', '6 item epilogue (partials out)', '7 -']
if SPLIT:
    names[7] = '7 (inside the others) ring chunk boundaries: commit + barrier'
tot = sum(b[:7]) if SPLIT else sum(b[:8])
print('instrumented waves (x8 blocks):', n, ' total cycles/wave-item: %.0f' % (tot / n))
for i in range(8):
    print('%-28s %10.0f cycles  %5.1f %%' % (names[i], b[i] / n, 100.0 * b[i] / tot))
