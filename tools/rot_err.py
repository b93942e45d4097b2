"""Error of the rotated-statistics path and of the plain folded path against the oracle in float64 (and float32).
    gpurun -- 'python tools/rot_err.py'"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from helpers import make_config, make_model, random_inputs, state_dict_cpu
from oracle import dgt_oracle as O

DEV = 'cuda:0'
CASES = [('vpsde_qm9_uncond_jodo', [6, 11, 20, 29, 1, 2, 29, 29, 17, 3], {}),
         ('vpsde_geom_uncond_jodo', [70, 33, 12, 150, 1, 2], {}),
         ('vpsde_geom_uncond_jodo', [70, 33, 12, 1, 2], dict(nf=384))]


def run(model, xh, ex, nl, nm, em, cx=None, cex=None):
    d = lambda t: None if t is None else t.to(DEV)
    with torch.no_grad():
        o = model(d(nl), d(xh), d(nm), d(em), edge_x=d(ex), cond_x=d(cx), cond_edge_x=d(cex), noise_level=d(nl))
    return [t.cpu() for t in o]


for cfg_name, n_nodes, over in CASES:
    cfg = make_config(cfg_name, **over)
    hp = O.Hyper.from_config(cfg)
    xh, ex, nl, ctx, nm, em = random_inputs(hp, n_nodes, seed=17)
    nl = torch.full_like(nl, -0.8)
    outs = {}
    for rot in (1, 0):
        model = make_model(cfg, 11, DEV, gain=1.5, coord_scale=0.05)
        model.plan_options = {6: rot}
        outs[rot] = run(model, xh, ex, nl, nm, em)
    sd = state_dict_cpu(model)
    with torch.no_grad():
        r32 = O.forward_dense(sd, hp, xh, nm, em, ex, None, None, nl)
        try:
            sd64 = {k: v.double() for k, v in sd.items()}
            r64 = O.forward_dense(sd64, hp, xh.double(), nm.double(), em.double(), ex.double(), None, None, nl.double())
        except Exception as e:      # noqa
            print('float64 oracle failed:', e)
            r64 = r32
    for k, name in ((0, 'nodes'), (1, 'edges')):
        e = lambda a, b: (a.double() - b.double()).abs().max().item()
        print(cfg_name, over, name, 'rot-vs-f64 %.2e plain-vs-f64 %.2e oracle32-vs-f64 %.2e rot-vs-plain %.2e  |x|max %.2f' % (
            e(outs[1][k], r64[k]), e(outs[0][k], r64[k]), e(r32[k], r64[k]), e(outs[1][k], outs[0][k]), r64[k].abs().max().item()))
