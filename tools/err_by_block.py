"""Where the kernels' float32 error comes from: after every block, the state (h, e, pos) of the HIP forward — rotated statistics and
plain fold — and of the float32 oracle, each against the float64 oracle.  Self-conditioned evaluation of the cases the tolerance
regimes of tests/helpers.py were written for (n = 150 at nf 256; nf 384).   gpurun -- 'python tools/err_by_block.py'"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import make_config, make_model, random_inputs, state_dict_cpu, debug_fetch
from jodo_amd import capi
from oracle import dgt_oracle as O

DEV = 'cuda:0'
CASES = [('vpsde_geom_uncond_jodo', [70, 33, 12, 150, 1, 2], {}), ('vpsde_geom_uncond_jodo', [70, 33, 12, 1, 2], dict(nf=384))]


_MASKS = {}


def run(model, xh, ex, nl, nm, em, cx=None, cex=None):
    d = lambda t: None if t is None else t.to(DEV)
    if id(nm) not in _MASKS:                       # the plan cache is keyed by the mask tensors: keep ONE device copy per batch
        _MASKS[id(nm)] = (nm, d(nm), d(em))
    _, nmd, emd = _MASKS[id(nm)]
    with torch.no_grad():
        o = model(d(nl), d(xh), nmd, emd, edge_x=d(ex), cond_x=d(cx), cond_edge_x=d(cex), noise_level=d(nl))
    torch.cuda.synchronize()
    return [t.cpu() for t in o]


for cfg_name, n_nodes, over in CASES:
    cfg = make_config(cfg_name, **over)
    hp = O.Hyper.from_config(cfg)
    xh, ex, nl, ctx, nm, em = random_inputs(hp, n_nodes, seed=17)
    nl = torch.full_like(nl, -0.8)
    model = make_model(cfg, 11, DEV, gain=1.5, coord_scale=0.05)
    sd = state_dict_cpu(model)
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        f1 = O.forward_dense(sd, hp, xh, nm, em, ex, None, None, nl)
        r32 = O.forward_dense(sd, hp, xh, nm, em, ex, f1[0], f1[1], nl, return_intermediates=True)
        r64 = O.forward_dense(sd64, hp, xh.double(), nm.double(), em.double(), ex.double(), f1[0].double(), f1[1].double(), nl.double(),
                              return_intermediates=True)
    order = sorted(range(len(n_nodes)), key=lambda b: -n_nodes[b])
    Nn, rows, D, De = sum(n_nodes), sum(n * n for n in n_nodes), hp.nf, hp.de
    print('==', cfg_name, over, n_nodes)
    for rot in (1, 0):
        model = make_model(cfg, 11, DEV, gain=1.5, coord_scale=0.05)
        model.plan_options = {6: rot}
        out = run(model, xh, ex, nl, nm, em, f1[0], f1[1])
        handle = model._last_plan['handle']
        for l in range(hp.n_layers):
            capi.check(capi.lib().jodo_debug_set_max_blocks(handle, l + 1), 'set_max_blocks')
            run(model, xh, ex, nl, nm, em, f1[0], f1[1])
            h = debug_fetch(model, 0, Nn * D).reshape(Nn, D)
            e = debug_fetch(model, 1, rows * De).reshape(rows, De)
            pos = debug_fetch(model, 2, Nn * 4).reshape(Nn, 4)[:, :3]
            w = dict(h=[0, 0], e=[0, 0], pos=[0, 0])
            no = eo = 0
            for b in order:
                n = n_nodes[b]
                offd = ~torch.eye(n, dtype=torch.bool)
                gp = pos[no:no + n] - pos[no:no + n].mean(0, keepdim=True)
                ge = e[eo:eo + n * n].reshape(n, n, De)
                t64, t32 = r64[2][b][l], r32[2][b][l]
                for name, got, k in (('h', h[no:no + n], 'h'), ('pos', gp, 'pos')):
                    w[name][0] = max(w[name][0], float((got.double() - t64[k]).abs().max()))
                    w[name][1] = max(w[name][1], float((t32[k].double() - t64[k]).abs().max()))
                if n > 1:
                    w['e'][0] = max(w['e'][0], float((ge.double() - t64['e'])[offd].abs().max()))
                    w['e'][1] = max(w['e'][1], float((t32['e'].double() - t64['e'])[offd].abs().max()))
                no += n; eo += n * n
            print('rot %d block %2d  HIP-vs-f64 (oracle32-vs-f64):  h %.2e (%.2e)   e %.2e (%.2e)   pos %.2e (%.2e)' % (
                rot, l, w['h'][0], w['h'][1], w['e'][0], w['e'][1], w['pos'][0], w['pos'][1]))
        capi.check(capi.lib().jodo_debug_set_max_blocks(handle, -1), 'set_max_blocks')
        ex_, ea = (out[0][..., :3].double() - r64[0][..., :3]).abs().max(), (out[0][..., 3:].double() - r64[0][..., 3:]).abs().max()
        o32x, o32a = (r32[0][..., :3].double() - r64[0][..., :3]).abs().max(), (r32[0][..., 3:].double() - r64[0][..., 3:]).abs().max()
        print('rot %d outputs: positions %.2e (oracle32 %.2e)  atom logits %.2e (oracle32 %.2e)  edges %.2e (oracle32 %.2e)' % (
            rot, ex_, o32x, ea, o32a, (out[1].double() - r64[1]).abs().max(), (r32[1].double() - r64[1]).abs().max()))
