"""Conditional JODO on QM9, single property (BASELINE config 5).

The reference's file has no `sampling.dpm_solver_*` keys: they are dropped here too,
so callers wanting the hybrid DPM-solver must add them, exactly as with the reference.
"""
from ._common import build


def get_config():
    cfg = build(dict(
        exp_type='vpsde_edge_cond', cond_property='alpha',
        data=dict(transform='EdgeComCond', collate='collate_cond', info_name='qm9_second_half'),
        model=dict(name='cond_DGT_concat', cond_ch=1),
        training=dict(n_iters=2000000),
        eval=dict(begin_ckpt=40, end_ckpt=40, sub_geometry=False),
    ), drop=(('sampling', 'dpm_solver_method'), ('sampling', 'dpm_solver_order')))
    return cfg
