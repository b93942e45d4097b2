"""Checkpoint ingestion: reference-format files (utils.py:7-30, models/ema.py:79-85)."""
import os
import sys

import pytest
import torch

from jodo_amd import configs
from jodo_amd.models import utils as mutils
from jodo_amd.models.ema import ExponentialMovingAverage
from jodo_amd.utils import load_for_sampling, restore_checkpoint, save_checkpoint


def _cfg():
    cfg = configs.get('vpsde_qm9_uncond_jodo')
    cfg.device = torch.device('cpu')
    return cfg


def test_ema_update_rule_and_positional_state():
    p = [torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.zeros(2), requires_grad=False),
         torch.nn.Parameter(torch.full((2,), 2.0))]
    ema = ExponentialMovingAverage(p, decay=0.999)
    assert len(ema.shadow_params) == 2                       # frozen tensors are skipped
    with torch.no_grad():
        p[0].add_(1.0)
    ema.update(p)
    d = min(0.999, 2 / 11)                                   # warm-up (1 + k) / (10 + k), k = 1
    assert torch.allclose(ema.shadow_params[0], torch.full((3,), 1.0 + (1 - d)))
    sd = ema.state_dict()
    assert set(sd) == {'decay', 'num_updates', 'shadow_params'} and sd['num_updates'] == 1
    ema.store(p)
    ema.copy_to(p)
    assert torch.allclose(p[0], ema.shadow_params[0])
    ema.restore(p)
    assert torch.allclose(p[0], torch.full((3,), 2.0))
    with pytest.raises(ValueError):
        ExponentialMovingAverage(p, decay=1.5)


def test_round_trip_and_ema_overwrite(tmp_path):
    cfg = _cfg()
    torch.manual_seed(3)
    model = mutils.create_model(cfg)
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_decay)
    with torch.no_grad():
        for s in ema.shadow_params:
            s.mul_(0.5)
    path = str(tmp_path / 'ckpt' / 'checkpoint_7.pth')
    os.makedirs(os.path.dirname(path))
    save_checkpoint(path, dict(optimizer=None, model=model, ema=ema, step=7))
    raw = torch.load(path)
    assert all(k.startswith('module.') for k in raw['model'])  # DataParallel-style keys on disk
    model2, ema2, step = load_for_sampling(path, cfg, use_ema=True)
    assert step == 7
    for (k, a), b in zip(model.state_dict().items(), model2.parameters()):
        assert torch.equal(a * 0.5, b), k
    versions = [p._version for p in model2.parameters()]
    ema2.copy_to(model2.parameters())
    assert all(p._version > v for p, v in zip(model2.parameters(), versions))   # invalidates packed weights
    with pytest.raises(FileNotFoundError):
        load_for_sampling(str(tmp_path / 'missing.pth'), cfg)


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='reference tree only exists in the build container')
def test_reference_written_checkpoint_loads_strict(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'oracle'))
    from ref_import import load_reference, reference_config
    ref = load_reference()
    rcfg = reference_config('vpsde_qm9_uncond_jodo')
    rcfg.device = torch.device('cpu')
    torch.manual_seed(11)
    import importlib
    ref.models                                               # puts the stand-ins and the reference on sys.path
    rutils = importlib.import_module('utils')                # the reference's top-level utils.py
    rmodel = importlib.import_module('models.utils').create_model(rcfg)
    rema = importlib.import_module('models.ema').ExponentialMovingAverage(rmodel.parameters(), decay=0.999)
    with torch.no_grad():
        for p in rmodel.parameters():
            p.add_(0.01)
    rema.update(rmodel.parameters())
    path = str(tmp_path / 'checkpoint_ref.pth')
    rutils.save_checkpoint(
        path, dict(optimizer=torch.optim.Adam(rmodel.parameters()), model=rmodel, ema=rema, step=123))
    model, ema, step = load_for_sampling(path, _cfg(), use_ema=False)
    assert step == 123
    for (k, a), (k2, b) in zip(rmodel.state_dict().items(), model.state_dict().items()):
        assert k == k2 and torch.equal(a, b), (k, k2)
    ema.copy_to(model.parameters())
    for s, b in zip(rema.shadow_params, model.parameters()):
        assert torch.equal(s, b)
    # ... and a file written by this package restores into the reference with strict=True
    path2 = str(tmp_path / 'checkpoint_ours.pth')
    save_checkpoint(path2, dict(optimizer=torch.optim.Adam(model.parameters()), model=model, ema=ema, step=5))
    rstate = dict(optimizer=torch.optim.Adam(rmodel.parameters()), model=rmodel, ema=rema, step=0)
    rstate = rutils.restore_checkpoint(path2, rstate, torch.device('cpu'))
    assert rstate['step'] == 5
