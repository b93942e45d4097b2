"""Shared builder for the JODO experiment configs.

Keys and default values follow /root/reference/configs/vpsde_qm9_uncond_jodo.py:7-119 (QM9 values);
the per-experiment files override what differs (reference: vpsde_geom_uncond_jodo.py,
vpsde_qm9_cond_jodo.py).  Written as nested plain dicts and converted, so that the same keys are
available through either ml_collections.ConfigDict (if installed) or jodo_amd.config_dict.ConfigDict.
"""
import copy

import torch

from ..config_dict import get_config_dict_class

_QM9 = dict(
    exp_type='vpsde_edge', pred_edge=True, only_2D=False,
    data=dict(root='data/QM9', name='QM9', processed_file='', transform='EdgeCom', collate='collate_edge',
              info_name='qm9_with_h', num_workers=16, compress_edge=True, centered=True,
              include_aromatic=False, atom_types=5, bond_types=4, fc_scale=[-1., 1.], max_node=29),
    sde=dict(schedule='cosine', continuous_beta_0=0.1, continuous_beta_1=20.),
    model=dict(name='DGT_concat', pred_data=True, include_fc_charge=True, normalize_factors='1, 4, 4, 1',
               ema_decay=0.999, edge_ch=2, nf=256, n_layers=8, n_heads=16, dropout=0.1, cond_time=True,
               dist_gbf=True, gbf_name='CondGaussianLayer', self_cond=True, self_cond_type='ori',
               edge_quan_th=0., n_extra_heads=2, CoM=True, mlp_ratio=2, spatial_cut_off=2.,
               softmax_inf=True, trans_name='TransMixLayer', loss_weights='1., 0.25, 0.1',
               noise_align=True),
    training=dict(reduce_mean=False, batch_size=128, eval_batch_size=128, eval_samples=128, log_freq=500,
                  n_iters=1500000, snapshot_freq=50000, snapshot_freq_for_preemption=10000,
                  snapshot_sampling=True),
    optim=dict(weight_decay=0, optimizer='AdamW', lr=2e-4, beta1=0.9, eps=1e-8, warmup=100000,
               grad_clip=10., disable_grad_log=True),
    sampling=dict(method='ancestral', steps=1000, vis_row=4, vis_col=4,
                  dpm_solver_method='singlestep_fixed', dpm_solver_order=2),
    eval=dict(enable_sampling=True, batch_size=2500, num_samples=10000, begin_ckpt=30, end_ckpt=30,
              ckpts='', save_graph=False, sub_geometry=True),
    seed=42,
)


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def _wrap(d, cls):
    out = cls()
    for k, v in d.items():
        out[k] = _wrap(v, cls) if isinstance(v, dict) else v
    return out


def build(overrides=None, drop=()):
    d = copy.deepcopy(_QM9)
    if overrides:
        _merge(d, overrides)
    for section, key in drop:
        d[section].pop(key, None)
    cfg = _wrap(d, get_config_dict_class())
    cfg.device = torch.device('cuda:0') if torch.cuda.is_available() else torch.device('cpu')
    return cfg
