"""One complete sampling round through the public entry points (get_sampling_fn -> sampler -> fused decode):
wall time of the whole round incl. decode and the device->host copies ("time one full round").
Usage: python tools/full_round.py [qm9|geom|cond] [batch] [steps] [ancestral|fast]   (fast = hybrid DPM-solver, steps = NFE)"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jodo_amd import configs
from jodo_amd.diffusion import NoiseScheduleVP
from jodo_amd.models import get_model_class, deterministic_init_, load_dataset_info, get_node_dist
from jodo_amd.sampling import get_sampling_fn
from jodo_amd.utils import get_data_inverse_scaler

which = sys.argv[1] if len(sys.argv) > 1 else 'qm9'
cfg_name, info = {'qm9': ('vpsde_qm9_uncond_jodo', 'qm9_with_h'), 'geom': ('vpsde_geom_uncond_jodo', 'geom_with_h_1'),
                  'cond': ('vpsde_qm9_cond_jodo', 'qm9_second_half')}[which]
cfg = configs.get(cfg_name)
B = int(sys.argv[2]) if len(sys.argv) > 2 else {'qm9': 2500, 'geom': 512, 'cond': 313}[which]
if len(sys.argv) > 3:
    cfg.sampling.steps = int(sys.argv[3])
if len(sys.argv) > 4:
    cfg.sampling.method = sys.argv[4]
    if sys.argv[4] == 'fast':                        # keys the reference's cond config lacks (values of the uncond config, :102-103)
        cfg.sampling['dpm_solver_method'] = 'singlestep_fixed'
        cfg.sampling['dpm_solver_order'] = 2
dev = torch.device('cuda:0')
cfg.device = dev
torch.manual_seed(cfg.seed)
model = deterministic_init_(get_model_class(cfg.model.name)(cfg), seed=42).to(dev).eval()
ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
nodes_dist = get_node_dist(load_dataset_info(info))


class _Prop:                                           # synthetic property sampler for the conditional config
    def sample_batch(self, n_nodes):
        return torch.randn(len(n_nodes), 1)


fn = get_sampling_fn(cfg, ns, nodes_dist, B, B, get_data_inverse_scaler(cfg),
                     prop_dist=_Prop() if which == 'cond' else None, return_raw=True,
                     hip_graph=os.environ.get('HIP_GRAPH', '0') == '1')
if os.environ.get('WARM', '1') == '1':               # first call pays weight packing + first-touch costs
    fn(model)
torch.cuda.synchronize()
t0 = time.perf_counter()
mols = fn(model)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({'workload': which, 'batch': B, 'steps': int(cfg.sampling.steps), 'method': cfg.sampling.method,
                  'round_seconds': dt, 'molecules': len(mols), 'molecules_per_s': len(mols) / dt, 'hip_graph': os.environ.get('HIP_GRAPH', '0') == '1',
                  'ms_per_step': dt / int(cfg.sampling.steps) * 1e3, 'nan_guard': bool(model.nan_guard_fired()),
                  'first_molecule_atoms': int(mols[0][0].shape[0])}))
