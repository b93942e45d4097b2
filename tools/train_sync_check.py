"""Which operations of a training step synchronise the host with the device?  Runs steps on loader-style (CPU) batches under
torch.cuda.set_sync_debug_mode('warn') and prints every warning with the Python line that triggered it."""
import os, sys, random, warnings, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.train_bench import synthetic_batch
from jodo_amd import configs, losses as L
from jodo_amd.diffusion import NoiseScheduleVP
from jodo_amd.models import get_model_class, deterministic_init_, load_dataset_info, get_node_dist
from jodo_amd.models.ema import ExponentialMovingAverage
from jodo_amd.utils import get_data_scaler

cfg = configs.get('vpsde_qm9_uncond_jodo')
dev = torch.device('cuda:0'); cfg.device = dev
B = 32
torch.manual_seed(1); random.seed(1)
dist_ = get_node_dist(load_dataset_info('qm9_with_h'))
model = deterministic_init_(get_model_class(cfg.model.name)(cfg), seed=1).to(dev)
ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
state = dict(model=model, optimizer=L.get_optimizer(cfg, model.parameters()), ema=ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_decay), step=1)
step_fn = L.get_step_fn(ns, True, L.optimization_manager(cfg), get_data_scaler(cfg), cfg)
batches = [synthetic_batch(cfg, dist_.sample(B).tolist(), 50 + i) for i in range(8)]
for i in range(4):
    step_fn(state, batches[i])
torch.cuda.synchronize()


def show(message, category, filename, lineno, file=None, line=None):
    stack = [f for f in traceback.extract_stack() if '/root/repo/' in f.filename or 'jodo_amd' in f.filename][-4:]
    print('SYNC:', str(message)[:80], '|', ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(stack)))


warnings.showwarning = show
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
for i in range(4, 8):
    step_fn(state, batches[i])
torch.cuda.set_sync_debug_mode('default')
torch.cuda.synchronize()
print('done')
