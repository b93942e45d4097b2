"""CPU-baseline calibration (build container only): BASELINE config 1 in full —
vpsde_qm9_uncond_jodo, batch 64, 50 ancestral steps, PyTorch CPU — timed for

  (a) the REAL reference (imported from /root/reference under oracle/standins): its own
      `AncestralSampler.sampling` (sampling.py:530-596) driving its own `DGT_concat`;
  (b) the port that bench.py's `cpu_baseline` leg times on the GPU box (where the reference cannot go):
      jodo_amd's host sampler driving `oracle.dgt_oracle.forward_faithful`,

on the same seeded inputs, same weights (deterministic_init_), same thread count.  Prints one JSON line with
both wall times, their ratio and the max difference of the two end states; the numbers are copied into
BASELINE.md.  TEST INFRASTRUCTURE ONLY.

    python oracle/calibrate_cpu.py [--steps 50] [--batch 64] [--threads N]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_import import load_reference, reference_config          # noqa: E402
from oracle import dgt_oracle as O                                      # noqa: E402
from jodo_amd.models.init_utils import deterministic_init_              # noqa: E402


def config1_inputs(batch, seed=42):
    """n_nodes ~ the QM9 training histogram, masks, z_T, edge_z_T — identical for both legs."""
    from jodo_amd.models import load_dataset_info, get_node_dist
    from jodo_amd.sampling import build_masks
    from jodo_amd.models.utils import sample_combined_position_feature_noise, sample_symmetric_edge_feature_noise
    torch.manual_seed(seed)
    n_nodes = get_node_dist(load_dataset_info('qm9_with_h')).sample(batch).tolist()
    N = max(n_nodes)
    nm, em = build_masks(n_nodes, N, 'cpu')
    z = sample_combined_position_feature_noise(batch, N, 6, nm)
    ez = sample_symmetric_edge_feature_noise(batch, N, 2, em)
    return n_nodes, nm, em, z, ez


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--threads', type=int, default=os.cpu_count())
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    ref = load_reference()
    cfg = reference_config('vpsde_qm9_uncond_jodo')
    cfg.device = torch.device('cpu')
    model = ref.models.utils._MODELS[cfg.model.name](cfg).eval()
    deterministic_init_(model, seed=42)
    n_nodes, nm, em, z, ez = config1_inputs(args.batch)
    S = ref.sampling
    ns = ref.diffusion.noise_schedule.NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0,
                                                      continuous_beta_1=cfg.sde.continuous_beta_1)
    ts = torch.linspace(ns.T, 1e-3, args.steps)
    out = {}
    # (a) the reference
    smp = S.AncestralSampler(ns, ts, cfg.model.pred_data, cfg.pred_edge, cfg.model.self_cond, ref.utils.get_self_cond_fn(cfg))
    torch.manual_seed(1)
    t0 = time.perf_counter()
    with torch.no_grad():
        xa, ea = smp.sampling(model, z, nm, em, ez, None)
    out['reference_s'] = time.perf_counter() - t0
    # (b) the port: our host sampler + forward_faithful (what bench.py times on the GPU box)
    from jodo_amd import configs
    from jodo_amd.diffusion import NoiseScheduleVP
    from jodo_amd.sampling import AncestralSampler
    from jodo_amd.utils import get_self_cond_fn
    ours = configs.get('vpsde_qm9_uncond_jodo')
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    hp = O.Hyper.from_config(ours)

    class Port:
        def __call__(self, t, xh, node_mask, edge_mask, context=None, **kw):
            return O.forward_faithful(sd, hp, xh, node_mask, edge_mask, kw['edge_x'], kw.get('cond_x'), kw.get('cond_edge_x'),
                                      kw['noise_level'], context)

    ns2 = NoiseScheduleVP(ours.sde.schedule, continuous_beta_0=ours.sde.continuous_beta_0, continuous_beta_1=ours.sde.continuous_beta_1)
    smp2 = AncestralSampler(ns2, torch.linspace(ns2.T, 1e-3, args.steps), True, True, True, get_self_cond_fn(ours))
    torch.manual_seed(1)
    t0 = time.perf_counter()
    with torch.no_grad():
        xb, eb = smp2.sampling(Port(), z, nm, em, ez, None)
    out['port_s'] = time.perf_counter() - t0
    out.update(config='vpsde_qm9_uncond_jodo, batch %d, %d ancestral steps, PyTorch CPU' % (args.batch, args.steps),
               threads=args.threads, cpu=cpu_model(), port_over_reference=out['port_s'] / out['reference_s'],
               reference_s_per_step=out['reference_s'] / args.steps, port_s_per_step=out['port_s'] / args.steps,
               reference_molecules_per_s=args.batch / out['reference_s'], port_molecules_per_s=args.batch / out['port_s'],
               end_state_max_diff=max((xa - xb).abs().max().item(), (ea - eb).abs().max().item()))
    print(json.dumps(out))


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


if __name__ == '__main__':
    main()
