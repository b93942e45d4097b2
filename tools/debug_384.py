"""nf=384 fixture: HIP vs fp32 reference vs an fp64 evaluation of the dense oracle (is the gap rounding noise?)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import load_fixture, make_config, make_model, masks, state_dict_cpu
from oracle import dgt_oracle as O

fname = sys.argv[1] if len(sys.argv) > 1 else 'fwd_geom384.npz'
layout = sys.argv[2] if len(sys.argv) > 2 else 'auto'
fx = load_fixture(fname)
over = dict(kernel_layout=layout)
if 'nf' in fx: over['nf'] = int(fx['nf'])
cfg = make_config(str(fx['cfg_name']), **over)
model = make_model(cfg, int(fx['seed']), 'cuda:0')
hp = O.Hyper.from_config(cfg)
nm, em = masks(fx['n_nodes'].tolist())
t = lambda k: torch.from_numpy(fx[k])
ctx = t('context') if hp.cond_ch else None
sd = state_dict_cpu(model)
sd64 = {k: v.double() for k, v in sd.items()}
d = lambda x: None if x is None else x.to('cuda:0')
for step, (cx, cex, wx, we) in enumerate(((None, None, t('out1_x'), t('out1_e')), (t('out1_x'), t('out1_e'), t('out2_x'), t('out2_e')))):
    with torch.no_grad():
        o = model(d(t('noise_level')), d(t('xh')), d(nm), d(em), edge_x=d(t('edge_x')), cond_x=d(cx), cond_edge_x=d(cex),
                  noise_level=d(t('noise_level')), context=d(ctx))
        r64 = O.forward_dense(sd64, hp, t('xh').double(), nm.double(), em.double(), t('edge_x').double(),
                              None if cx is None else cx.double(), None if cex is None else cex.double(),
                              t('noise_level').double(), None if ctx is None else ctx.double())
    hx, he = o[0].cpu().double(), o[1].cpu().double()
    e = lambda a, b: (a - b).abs().max().item()
    print('step', step, 'pos: hip-ref %.2e hip-f64 %.2e ref-f64 %.2e | logits: hip-ref %.2e hip-f64 %.2e ref-f64 %.2e | edges: hip-ref %.2e hip-f64 %.2e ref-f64 %.2e' % (
        e(hx[..., :3], wx[..., :3].double()), e(hx[..., :3], r64[0][..., :3]), e(wx[..., :3].double(), r64[0][..., :3]),
        e(hx[..., 3:], wx[..., 3:].double()), e(hx[..., 3:], r64[0][..., 3:]), e(wx[..., 3:].double(), r64[0][..., 3:]),
        e(he, we.double()), e(he, r64[1]), e(we.double(), r64[1])))
