"""JODO on GEOM-Drugs, unconditional (BASELINE configs 3 and 4; `model.nf = 384` for "large")."""
from ._common import build


def get_config():
    return build(dict(
        data=dict(root='data/geom', name='GeomDrug', processed_file='data_geom_drug_1.pt',
                  info_name='geom_with_h_1', include_aromatic=True, atom_types=16, bond_types=5,
                  fc_scale=[-2., 3.], max_node=181),
        model=dict(edge_ch=3, nf=256, n_layers=10, mlp_ratio=4, spatial_cut_off=3.,
                   loss_weights='1, 0.25, 0.1'),
        training=dict(batch_size=16, eval_batch_size=16),
        optim=dict(grad_clip=20.),
        eval=dict(batch_size=1000),
    ))
