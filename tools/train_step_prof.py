"""Whole optimiser steps of the training path for `rocprofv3 --kernel-trace`: which kernels (library and torch) a step's time is in and
how much of it the card idles.   (cd /tmp && rocprofv3 --kernel-trace -d OUT -o ts -- python tools/train_step_prof.py [steps] [batch];
tools/train_step_prof.py --report OUT/ts_results.db [steps])"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def report(db_path, steps):
    import sqlite3, re, collections
    c = sqlite3.connect(db_path).cursor()
    rows = c.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start").fetchall()
    # the timed steps are bracketed by two marker launches of a fill kernel with a telltale size: find the longest run between the last two big gaps instead:
    # simply take the last `steps` / (warm-up + steps) fraction by launch count
    n = len(rows)
    frac = steps / float(steps + 3)
    rows = rows[int(n * (1 - frac)):]
    span = rows[-1][2] - rows[0][1]
    busy = 0; last_end = rows[0][1]; gaps = []
    for name, st, en in rows:
        if st > last_end:
            gaps.append((st - last_end, name))
        busy += max(0, en - max(st, last_end))
        last_end = max(last_end, en)
    agg = collections.defaultdict(lambda: [0, 0])
    for name, st, en in rows:
        k = re.sub(r'\(.*', '', name)[:100]
        agg[k][0] += en - st; agg[k][1] += 1
    print('launches/step %.0f  span/step %.3f ms  busy/step %.3f ms  idle/step %.3f ms' % (len(rows) / steps, span / steps / 1e6, busy / steps / 1e6, (span - busy) / steps / 1e6))
    lib = sum(v[0] for k, v in agg.items() if 'jt' in k or 'jodo' in k or '_GLOBAL__N_' in k or 'jd' in k)
    print('library kernels/step %.3f ms, others %.3f ms' % (lib / steps / 1e6, (sum(v[0] for v in agg.values()) - lib) / steps / 1e6))
    print('-- non-library kernels')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        if not ('jt' in k or 'jodo' in k or '_GLOBAL__N_' in k):
            print('%8.3f ms/step %7.1f calls/step %7.1f us  %s' % (v[0] / steps / 1e6, v[1] / steps, v[0] / v[1] / 1e3, k))
    print('-- largest gaps')
    for g, nm in sorted(gaps, reverse=True)[:12]:
        print('%8.1f us before %s' % (g / 1e3, re.sub(r'\(.*', '', nm)[:90]))


if len(sys.argv) > 1 and sys.argv[1] == '--report':
    report(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 6)
    sys.exit(0)

import random
import torch
from tools.train_bench import synthetic_batch
from jodo_amd import configs, losses as L
from jodo_amd.diffusion import NoiseScheduleVP
from jodo_amd.models import get_model_class, deterministic_init_, load_dataset_info, get_node_dist
from jodo_amd.models.ema import ExponentialMovingAverage
from jodo_amd.utils import get_data_scaler

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = configs.get('vpsde_qm9_uncond_jodo')
dev = torch.device('cuda:0')
cfg.device = dev
B = int(sys.argv[2]) if len(sys.argv) > 2 else int(cfg.training.batch_size)
torch.manual_seed(42); random.seed(42)
dist_ = get_node_dist(load_dataset_info('qm9_with_h'))
model = deterministic_init_(get_model_class(cfg.model.name)(cfg), seed=42).to(dev)
ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
state = dict(model=model, optimizer=L.get_optimizer(cfg, model.parameters()), ema=ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_decay), step=1)
step_fn = L.get_step_fn(ns, True, L.optimization_manager(cfg), get_data_scaler(cfg), cfg)
batches = [synthetic_batch(cfg, dist_.sample(B).tolist(), 43 + i) for i in range(3 + steps)]      # CPU dicts, as the reference's loader hands them over
for i in range(3):
    step_fn(state, batches[i])
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(3, 3 + steps):
    step_fn(state, batches[i])
torch.cuda.synchronize()
print('ms per step: %.3f' % ((time.perf_counter() - t0) / steps * 1e3))
