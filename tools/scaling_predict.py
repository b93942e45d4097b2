"""Host-side multi-GPU balance prediction for the BASELINE configs (no GPU needed: plans and the executed-work model are host C++).

    python tools/scaling_predict.py [--out profiles/r05_scaling_prediction.json]

* weak scaling as `bench.py --gpus N` runs it: rank r samples its own B molecules drawn from seed 42 + r (N = 1, 2, 4, 8);
* BASELINE configs[3] (GEOM nf 384, 10 000 molecules over 8 GPUs) and configs[4] (QM9 conditional + DPM-solver, 10 000 over 8):
  the molecules dealt to 8 / 4 / 2 ranks as get_sampling_fn(shard=..., shard_mode='perf') deals them, `contiguous` and `lpt`.
See jodo_amd/scaling.py for what the figure means (work balance only)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    from jodo_amd import configs, scaling
    from jodo_amd.models import get_model_class, load_dataset_info, get_node_dist
    out = {'what': 'predicted scaling efficiency = mean / max over ranks of the executed MFMA flops of a rank\'s share (jodo_plan_work); '
                   'host-side, no GPU; see jodo_amd/scaling.py'}
    specs = {'qm9': ('vpsde_qm9_uncond_jodo', 'qm9_with_h', {}, 2500), 'geom': ('vpsde_geom_uncond_jodo', 'geom_with_h_1', {}, 512),
             'geom384': ('vpsde_geom_uncond_jodo', 'geom_with_h_1', {'nf': 384}, 1250), 'cond': ('vpsde_qm9_cond_jodo', 'qm9_second_half', {}, 1250)}
    for name, (cfgn, info, over, B) in specs.items():
        cfg = configs.get(cfgn)
        for k, v in over.items():
            cfg.model[k] = v
        model = get_model_class(cfg.model.name)(cfg)
        cs = model._cfg()
        uni = 0 if model.dims.cond_ch else 1
        dist_ = get_node_dist(load_dataset_info(info))
        weak = {}
        draws = []
        for r in range(8):
            torch.manual_seed(cfg.seed + r)
            draws.append(dist_.sample(B).tolist())
        for n in (2, 4, 8):
            weak[str(n)] = scaling.predict_weak(cs, draws[:n], uni)
        out[name] = {'config': cfgn, 'batch_per_gpu': B, 'weak_scaling_as_bench_runs_it': weak}
        if name in ('geom384', 'cond'):
            torch.manual_seed(cfg.seed)
            n_all = dist_.sample(10000).tolist()
            out[name]['dealt_10000'] = {str(w): scaling.predict_dealt(cs, n_all, w, B, uni) for w in (2, 4, 8)}
        print(name, 'weak x8: %.4f' % weak['8']['predicted_efficiency'],
              {k: round(v['predicted_efficiency'], 4) for k, v in out[name].get('dealt_10000', {}).get('8', {}).items()}, file=sys.stderr)
    s = json.dumps(out, indent=1)
    if args.out:
        open(args.out, 'w').write(s + '\n')
    else:
        print(s)


if __name__ == '__main__':
    main()
