"""Per-step teacher-forced comparison HIP vs oracle along the golden ancestral trajectory."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from helpers import *          # noqa
from jodo_amd.diffusion import NoiseScheduleVP
from jodo_amd.sampling import AncestralSampler
from jodo_amd.utils import get_self_cond_fn

fx = load_fixture('traj_qm9_anc5.npz')
cfg = make_config('vpsde_qm9_uncond_jodo')
hg = float(sys.argv[1]) if len(sys.argv) > 1 else float(fx['head_gain'])
model = make_model(cfg, int(fx['seed']), 'cuda:0', head_gain=hg)
hp = O.Hyper.from_config(cfg)
sd = state_dict_cpu(model)
nm, em = masks(fx['n_nodes'].tolist())
ns = NoiseScheduleVP('cosine')
noise = {'node': torch.from_numpy(fx['node_noise']), 'edge': torch.from_numpy(fx['edge_noise'])}
calls = []


class Both:
    def eval(self): return self
    def __call__(self, t, xh, node_mask, edge_mask, context=None, **kw):
        with torch.no_grad():
            ref = O.forward_dense(sd, hp, xh, node_mask, edge_mask, kw['edge_x'], kw.get('cond_x'), kw.get('cond_edge_x'), kw['noise_level'])
            d = lambda x: None if x is None else x.cuda()
            got = model(d(t), d(xh), d(node_mask), d(edge_mask), edge_x=d(kw['edge_x']), cond_x=d(kw.get('cond_x')),
                        cond_edge_x=d(kw.get('cond_edge_x')), noise_level=d(kw['noise_level']))
        gx, ge = got[0].cpu(), got[1].cpu()
        print("step %d  nl %.3f  pos %.2e  atom %.2e  edge %.2e   | ref mag atom %.2f edge %.2f" % (
            len(calls), kw['noise_level'][0].item(), (gx[..., :3] - ref[0][..., :3]).abs().max(), (gx[..., 3:] - ref[0][..., 3:]).abs().max(),
            (ge - ref[1]).abs().max(), ref[0][..., 3:].abs().max(), ref[1].abs().max()))
        calls.append(1)
        return ref


sampler = AncestralSampler(ns, torch.linspace(ns.T, 1e-3, 5), True, True, True, get_self_cond_fn(cfg),
                           noise_fn=lambda i, kind, like: noise[kind][i])
x_mean, e_mean = sampler.sampling(Both(), torch.from_numpy(fx['z']), nm, em, torch.from_numpy(fx['edge_z']), None)
print('final vs fixture', (x_mean - torch.from_numpy(fx['x_mean'])).abs().max().item())
