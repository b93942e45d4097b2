"""Caller-side glue kept as host Python: data (inverse) scaling, self-conditioning post-process,
checkpoint I/O.  Behaviour of /root/reference/utils.py:7-30 (checkpoints), :71-105 (inverse scaler),
:108-150 (self-cond fn)."""
import logging
import os

import numpy as np
import torch


def _norm_factors(config):
    nf = config.model.normalize_factors
    if isinstance(nf, str):
        nf = [int(s) for s in nf.split(',')]
    return list(nf)


def get_data_scaler(config):
    """Training normalisation (/root/reference/utils.py:33-68): fn(pos, atom_type, fc_charge, node_mask, edge_type=None,
    edge_mask=None) divides by the factors of config.model.normalize_factors, maps one-hots to [-1, 1] when
    config.data.centered, and masks."""
    nf = _norm_factors(config)
    pos_norm, atom_norm, fc_norm = nf[0], nf[1], nf[2]
    edge_norm = nf[3] if len(nf) > 3 else 1
    centered = config.data.centered

    def scale_fn(pos, atom_type, fc_charge, node_mask, edge_type=None, edge_mask=None):
        if centered:
            atom_type = atom_type * 2. - 1.
        if pos is not None:
            pos = pos / pos_norm * node_mask
        atom_type = atom_type / atom_norm * node_mask
        fc_charge = fc_charge / fc_norm * node_mask
        if edge_type is None:
            return pos, atom_type, fc_charge
        if centered:
            edge_type = edge_type * 2. - 1.
        n = node_mask.size(1)
        edge_type = edge_type / edge_norm * edge_mask.reshape(node_mask.size(0), n, n, 1)
        return pos, atom_type, fc_charge, edge_type

    return scale_fn


def get_data_inverse_scaler(config):
    """Returns fn(pos, atom_type, fc_charge, node_mask, edge_type=None, edge_mask=None) that undoes
    the training normalisation: multiply by the factors, and map centred [-1,1] one-hots to [0,1]."""
    nf = _norm_factors(config)
    pos_norm, atom_norm, fc_norm = nf[0], nf[1], nf[2]
    edge_norm = nf[3] if len(nf) > 3 else 1
    centered = config.data.centered

    def inverse_scale_fn(pos, atom_type, fc_charge, node_mask, edge_type=None, edge_mask=None):
        if pos is not None:
            pos = pos * pos_norm * node_mask
        atom_type = atom_type * atom_norm
        fc_charge = fc_charge * fc_norm * node_mask
        if centered:
            atom_type = (atom_type + 1.) / 2. * node_mask
        if edge_type is None:
            return pos, atom_type, fc_charge
        edge_type = edge_type * edge_norm
        if centered:
            edge_type = (edge_type + 1.) / 2.
        n = node_mask.size(1)
        edge_type = edge_type * edge_mask.reshape(node_mask.size(0), n, n, 1)
        return pos, atom_type, fc_charge, edge_type

    inverse_scale_fn.from_config = True       # lets the sampler use the equivalent device-side decode (fused.decode)
    return inverse_scale_fn


def get_self_cond_fn(config):
    """'ori': identity; 'clamp': clamp predicted atom/charge/edge channels to their data range."""
    kind = config.model.self_cond_type
    atom_types = config.data.atom_types
    include_fc = config.model.include_fc_charge
    _, atom_norm, fc_norm, edge_norm = _norm_factors(config)
    atom_rng = np.array([0., 1.])
    edge_rng = np.array([0., 1.])
    fc_rng = np.array(config.data.fc_scale, dtype=np.float64)
    if config.data.centered:
        atom_rng = atom_rng * 2. - 1.
        edge_rng = edge_rng * 2. - 1.
    atom_rng, fc_rng, edge_rng = atom_rng / atom_norm, fc_rng / fc_norm, edge_rng / edge_norm

    def process_self_cond(cond_x, cond_edge_x):
        if kind == 'ori':
            return cond_x, cond_edge_x
        if kind == 'clamp':
            cond_x[:, :, 3:3 + atom_types] = cond_x[:, :, 3:3 + atom_types].clamp(atom_rng[0], atom_rng[1])
            if include_fc:
                cond_x[:, :, -1:] = cond_x[:, :, -1:].clamp(fc_rng[0], fc_rng[1])
            return cond_x, cond_edge_x.clamp(edge_rng[0], edge_rng[1])
        raise ValueError("Self-condition data process error.")

    return process_self_cond


def expand_dims(v, dims):
    return v[(...,) + (None,) * (dims - 1)]


def restore_checkpoint(ckpt_dir, state, device):
    """state = {'optimizer', 'model', 'ema', 'step'}; same on-disk format as the reference
    (torch.save of the four state_dicts, model keys carry the DataParallel 'module.' prefix)."""
    if not os.path.exists(ckpt_dir):
        os.makedirs(os.path.dirname(ckpt_dir), exist_ok=True)
        logging.warning(f"No checkpoint found at {ckpt_dir}. Returned the same state as input")
        return state
    loaded = torch.load(ckpt_dir, map_location=device)
    if state.get('optimizer') is not None and 'optimizer' in loaded:
        state['optimizer'].load_state_dict(loaded['optimizer'])
    state['model'].load_state_dict(loaded['model'], strict=True)
    if state.get('ema') is not None:
        state['ema'].load_state_dict(loaded['ema'])
    state['step'] = loaded['step']
    return state


def save_checkpoint(ckpt_dir, state):
    torch.save({'optimizer': state['optimizer'].state_dict() if state.get('optimizer') is not None else None,
                'model': state['model'].state_dict(),
                'ema': state['ema'].state_dict() if state.get('ema') is not None else None,
                'step': state['step']}, ckpt_dir)


def load_for_sampling(ckpt_path, config, device=None, use_ema=True):
    """The reference's evaluation prologue in one call (run_lib.py:175-176, 221-222): build the model
    from `config`, restore a reference-format checkpoint and overwrite the parameters with the EMA
    shadow weights.  Returns (model, ema, step).  The packed kernel weights are rebuilt lazily on the
    next forward (the cache is keyed on parameter versions)."""
    from .models import utils as mutils
    from .models.ema import ExponentialMovingAverage
    if not os.path.exists(ckpt_path):
        raise FileNotFoundError(ckpt_path)         # the reference silently continues with random weights
    device = device or config.device
    model = mutils.create_model(config)
    ema = ExponentialMovingAverage(model.parameters(), decay=config.model.ema_decay)
    state = dict(optimizer=None, model=model, ema=ema, step=0)
    state = restore_checkpoint(ckpt_path, state, device)
    if use_ema:
        ema.copy_to(model.parameters())
    model.eval()
    return model, ema, state['step']
