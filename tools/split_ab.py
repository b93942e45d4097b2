"""A/B of the opt-in split-bf16 pair update (JODO_OPT_SPLIT_BF16) against the exact-fp32 default on a bench workload: ms per denoise
step, HIP-event class times (pair update us per block), and the distance of the split path's outputs from the default path's on the
full batch.   gpurun -- 'python tools/split_ab.py [--workload qm9|geom] [--steps 30]'  -> gpurun_out/split_ab_<workload>.txt"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jodo_amd import configs, fused
from jodo_amd.diffusion import NoiseScheduleVP
from jodo_amd.models import get_model_class, deterministic_init_, load_dataset_info, get_node_dist
from jodo_amd.sampling import AncestralSampler, build_masks
from jodo_amd.models.utils import sample_combined_position_feature_noise, sample_symmetric_edge_feature_noise
from jodo_amd.utils import get_self_cond_fn

WL = {'qm9': ('vpsde_qm9_uncond_jodo', 'qm9_with_h', 2500), 'geom': ('vpsde_geom_uncond_jodo', 'geom_with_h_1', 512),
      'geom384': ('vpsde_geom_uncond_jodo', 'geom_with_h_1', 1250)}
CLS = ['prologue', 'node_pre', 'attention', 'unused3', 'unused4', 'node_post', 'pair_update', 'heads']
ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='qm9', choices=sorted(WL))
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--batch', type=int, default=0)
ap.add_argument('--round', type=int, default=0, help='also run a complete round of that many ancestral steps with both paths on identical in-kernel noise and compare the decoded molecules')
ap.add_argument('--attention', action='store_true', help="also time split_bf16 = 'attention' (option value 2: the two-launch split attention on top)")
args = ap.parse_args()
cfg_name, info, B = WL[args.workload]
B = args.batch or B
dev = torch.device('cuda:0')
cfg = configs.get(cfg_name)
cfg.device = dev
if args.workload == 'geom384':
    cfg.model.nf = 384                         # README.md:168 `--config.model.nf 384`
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


torch.manual_seed(cfg.seed)
n_nodes = get_node_dist(load_dataset_info(info)).sample(B).tolist()
N = max(n_nodes)
nm, em = build_masks(n_nodes, N, dev)
node_nf = cfg.data.atom_types + int(cfg.model.include_fc_charge)
z = sample_combined_position_feature_noise(B, N, node_nf, nm)
ez = sample_symmetric_edge_feature_noise(B, N, cfg.model.edge_ch, em)
ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
ts = torch.linspace(ns.T, 1e-3, 1000)
say('# split-bf16 pair update A/B, %s B = %d, %s' % (args.workload, B, torch.cuda.get_device_name(0)))
res, state5 = {}, {}
for split in (False, True, False, True) + (('attention', 'attention') if args.attention else ()):
    model = deterministic_init_(get_model_class(cfg.model.name)(cfg), seed=cfg.seed).to(dev).eval()
    model.split_bf16 = split
    sampler = AncestralSampler(ns, ts, True, True, True, get_self_cond_fn(cfg), device_noise=fused.DeviceNoise.for_rank(cfg.seed, 0))
    with torch.no_grad():
        st = sampler.init_state(z, ez)
        for i in range(5):
            st = sampler.step(model, i, st, nm, em, None)
        torch.cuda.synchronize()
        if split not in state5:
            state5[split] = (st['x'].clone() if 'x' in st else None, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()})
        model.profile_enable(1)
        t0 = time.perf_counter()
        for i in range(5, 5 + args.steps):
            st = sampler.step(model, i, st, nm, em, None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        ms, cnt = model.profile_read()
        model.profile_enable(0)
    nb = cfg.model.n_layers
    say('split_bf16=%-5s  %.3f ms/step (class timers on)   pair update %.1f us/block   attention %.1f   node class %.1f us/block   nan %d' % (
        split, dt * 1e3, ms[6] / args.steps / nb * 1e3, ms[2] / args.steps / nb * 1e3, ms[5] / args.steps / nb * 1e3, model.take_nan_count()))
    res.setdefault(split, []).append((dt * 1e3, ms[6] / args.steps / nb * 1e3, ms[2] / args.steps / nb * 1e3))
a, b = state5[False][1], state5[True][1]
for k in a:
    if torch.is_tensor(a[k]) and a[k].dtype.is_floating_point and a[k].shape == b[k].shape:
        say('after 5 identical-noise steps, state[%s]: max |split - default| %.3e (max |x| %.2f)' % (k, float((a[k] - b[k]).abs().max()), float(a[k].abs().max())))
bd, bs = min(r[1] for r in res[False]), min(r[1] for r in res[True])
say('pair update: exact fp32 %.1f us/block -> split bf16x3 %.1f us/block (x%.2f);  step %.3f -> %.3f ms' % (bd, bs, bd / bs, min(r[0] for r in res[False]), min(r[0] for r in res[True])))
if args.attention:
    c = state5['attention'][1]
    for k in a:
        if torch.is_tensor(a[k]) and a[k].dtype.is_floating_point and a[k].shape == c[k].shape:
            say("after 5 identical-noise steps, state[%s]: max |split 'attention' - default| %.3e" % (k, float((a[k] - c[k]).abs().max())))
    say("attention: exact fp32 %.1f us/block -> two split launches %.1f us/block;  step with all three split kernels %.3f ms" % (
        min(r[2] for r in res[False]), min(r[2] for r in res['attention']), min(r[0] for r in res['attention'])))
if args.round > 0:
    # a complete round, both paths on the SAME in-kernel noise stream (Philox keyed by (seed, rank, round)): what SURVEY.md 8c asks of long
    # runs is a statistical comparison; with identical noise the two paths can be compared molecule by molecule as well
    import numpy as np
    ts_r = torch.linspace(ns.T, 1e-3, args.round)
    dec = {}
    for split in (False, True):
        model = deterministic_init_(get_model_class(cfg.model.name)(cfg), seed=cfg.seed).to(dev).eval()
        model.split_bf16 = split
        sampler = AncestralSampler(ns, ts_r, True, True, True, get_self_cond_fn(cfg), device_noise=fused.DeviceNoise.for_rank(cfg.seed, 0))
        with torch.no_grad():
            t0 = time.perf_counter()
            x_mean, e_mean = sampler.sampling(model, z, nm, em, ez, None)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            pos, at, fc, et = fused.decode(cfg, x_mean, e_mean, fused.n_nodes_from_mask(nm))
        dec[split] = (x_mean.cpu(), e_mean.cpu(), pos.cpu(), at.cpu(), fc.cpu(), et.cpu())
        say('%d-step round, split_bf16=%-5s %.2f s = %.1f molecules/s' % (args.round, split, dt, B / dt))
    a, b = dec[False], dec[True]
    real = (nm[..., 0] > 0).cpu()
    emk = (em.reshape(B, N, N) > 0).cpu()
    say('end of the %d-step round, split vs default on identical noise: max |x_mean diff| %.3e, max |edge_x_mean diff| %.3e; atom types equal on %.4f of the atoms, '
        'charges on %.4f, bond types on %.4f of the edges; atom-type histogram default %s split %s' % (
            args.round, float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max()), float((a[3] == b[3])[real].float().mean()),
            float((a[4] == b[4])[real].float().mean()), float((a[5] == b[5])[emk].float().mean()),
            np.bincount(a[3][real].numpy(), minlength=cfg.data.atom_types).tolist(), np.bincount(b[3][real].numpy(), minlength=cfg.data.atom_types).tolist()))
os.makedirs('gpurun_out', exist_ok=True)
open('gpurun_out/split_ab_%s.txt' % args.workload, 'w').write('\n'.join(lines) + '\n')
