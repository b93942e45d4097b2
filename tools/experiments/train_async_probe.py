"""Where does an un-synchronised training step spend its host time?  Host timestamps around the sections of step_fn, fixed device batch."""
import os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tools.train_bench import synthetic_batch
from jodo_amd import configs, losses as L
from jodo_amd.diffusion import NoiseScheduleVP
from jodo_amd.models import get_model_class, deterministic_init_, load_dataset_info, get_node_dist
from jodo_amd.models.ema import ExponentialMovingAverage
from jodo_amd.utils import get_data_scaler

mode = sys.argv[1] if len(sys.argv) > 1 else 'async'
cfg = configs.get('vpsde_qm9_uncond_jodo')
dev = torch.device('cuda:0'); cfg.device = dev
B = 128
torch.manual_seed(1); random.seed(1)
dist_ = get_node_dist(load_dataset_info('qm9_with_h'))
model = deterministic_init_(get_model_class(cfg.model.name)(cfg), seed=1).to(dev)
ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
opt = L.get_optimizer(cfg, model.parameters())
ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_decay)
loss_fn = L.get_sde_graph_loss_fn(ns, True, get_data_scaler(cfg), cfg, None)
opt_fn = L.optimization_manager(cfg)
batch = {k: v.to(dev) for k, v in synthetic_batch(cfg, dist_.sample(B).tolist(), 50).items()}
names = ['zero_grad', 'loss_fn', 'backward', 'optimize', 'ema', 'wait']
acc = {n: 0.0 for n in names}
steps = 12
for it in range(4 + steps):
    if it == 4:
        torch.cuda.synchronize(); t_all = time.perf_counter(); acc = {n: 0.0 for n in names}
    t = [time.perf_counter()]
    opt.zero_grad(); t.append(time.perf_counter())
    loss = loss_fn(model, batch); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt_fn(opt, model.parameters(), step=it + 1); t.append(time.perf_counter())
    ema.update(model.parameters()); t.append(time.perf_counter())
    if mode == 'sync_end':
        torch.cuda.synchronize()
    elif mode == 'sync_mid':
        pass
    t.append(time.perf_counter())
    for n, a, b in zip(names, t, t[1:]):
        acc[n] += (b - a) * 1e3 / steps
torch.cuda.synchronize()
print(mode, 'ms/step %.2f' % ((time.perf_counter() - t_all) / steps * 1e3), {k: round(v, 2) for k, v in acc.items()})
