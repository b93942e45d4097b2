"""CPU: registry, configs, schedule, decode and error behaviour of the host-side mirror."""
import math

import pytest
import torch

from jodo_amd import configs
from jodo_amd.diffusion import NoiseScheduleVP
from jodo_amd.models import utils as mutils
from jodo_amd.models import get_model_class, get_node_dist, load_dataset_info
from jodo_amd.sampling import build_masks, post_process, posterior_coefficients
from jodo_amd.utils import get_data_inverse_scaler


def test_registry_behaviour():
    assert get_model_class('DGT_concat').__name__ == 'DGT_concat'
    assert get_model_class('cond_DGT_concat').conditional
    with pytest.raises(ValueError):
        mutils.register_model(get_model_class('DGT_concat'), name='DGT_concat')     # duplicate name

    @mutils.register_model
    class _Tmp(torch.nn.Module):
        pass
    assert mutils._MODELS['_Tmp'] is _Tmp
    del mutils._MODELS['_Tmp']


def test_create_model_exposes_dataparallel_keys():
    cfg = configs.get('vpsde_qm9_uncond_jodo')
    cfg.device = 'cpu'
    m = mutils.create_model(cfg)
    keys = list(m.state_dict().keys())
    assert len(keys) == 351 and all(k.startswith('module.') for k in keys)
    assert keys[0] == 'module.node_emb.weight' and keys[-1] == 'module.time_mlp.3.bias'
    assert sum(p.numel() for p in m.parameters()) == 27538584


def test_dataparallel_replication_is_refused_with_directions():
    """The reference's create_model wraps the model in torch.nn.DataParallel (models/utils.py:27).  Over ONE device that calls the
    module in place (tests/test_dgt_gpu.py runs a trajectory that way); over several it would shallow-copy the module — plan cache,
    ctypes handles and the packed blob of cuda:0 included — into worker threads every forward: refused, naming the replacement."""
    for name in ('DGT_concat', 'cond_DGT_concat'):
        cfg = configs.get('vpsde_qm9_cond_jodo' if name.startswith('cond') else 'vpsde_qm9_uncond_jodo')
        model = get_model_class(name)(cfg)
        with pytest.raises(RuntimeError, match=r'shard=\(rank, world\)'):
            model._replicate_for_data_parallel()
        with pytest.raises(RuntimeError, match='one process per GPU'):
            model._replicate_for_data_parallel()
    cfg = configs.get('vpsde_qm9_uncond_jodo')
    cfg.device = 'cpu'
    with pytest.raises(ValueError):
        mutils.create_model(cfg, wrap='replicas')


def test_unsupported_settings_fail_loudly():
    # (n_layers = 3 at nf 256: the per-block edge readout, 2 De / 3 = 42 features, does not fit one 32-row block)
    for key, val in (('dist_gbf', False), ('cond_time', False), ('pred_data', False), ('nf', 512), ('n_layers', 3)):
        cfg = configs.get('vpsde_qm9_uncond_jodo')
        cfg.model[key] = val
        with pytest.raises(NotImplementedError):
            get_model_class('DGT_concat')(cfg)


def test_reference_model_variants_construct():
    """The README's model variants beside the three configs: GEOM Base (nf 128, 6 layers, README.md:150) and Large (nf 384,
    README.md:168); derived sizes as the C side computes them (csrc/dgt_plan.cpp dgt_dims_from_cfg)."""
    for over, want in ((dict(nf=128, n_layers=6), (64, 16, 512, 128, True)), (dict(nf=384), (96, 32, 1344, 416, True)),
                       (dict(n_layers=8), (64, 16, 768, 192, False)), (dict(), (64, 16, 896, 224, False))):
        cfg = configs.get('vpsde_geom_uncond_jodo')
        for k, v in over.items():
            cfg.model[k] = v
        d = get_model_class('DGT_concat')(cfg).dims
        assert (d.cnp, d.cep, d.KNH, d.KEH, d.wide) == want, (over, (d.cnp, d.cep, d.KNH, d.KEH, d.wide))


def test_cosine_schedule_values():
    ns = NoiseScheduleVP('cosine')
    assert ns.T == 0.9946 and ns.total_N == 1000
    t = torch.tensor([0.5])
    la = ns.marginal_log_mean_coeff(t)
    want = math.log(math.cos((0.5 + 0.008) / 1.008 * math.pi / 2)) - math.log(math.cos(0.008 / 1.008 * math.pi / 2))
    assert abs(la.item() - want) < 1e-6
    a, s = ns.marginal_prob(t)
    assert abs((a ** 2 + s ** 2).item() - 1) < 1e-6
    lam = ns.marginal_lambda(t)
    assert abs(ns.inverse_lambda(lam).item() - 0.5) < 1e-4
    assert abs(ns.get_noiseLevel(t).item() - 2 * lam.item()) < 1e-5
    with pytest.raises(ValueError):
        NoiseScheduleVP('discrete')
    c_x, c_pred, sigma, *_ = posterior_coefficients(ns, torch.tensor(0.5), torch.tensor(0.0))
    assert abs(c_x.item()) < 1e-3 and abs(c_pred.item() - 1) < 1e-3 and abs(sigma.item()) < 1e-3   # s = 0: returns x0


def test_masks_and_node_distribution():
    nm, em = build_masks([2, 3], 3, 'cpu')
    assert nm[:, :, 0].tolist() == [[1, 1, 0], [1, 1, 1]]
    e = em.reshape(2, 3, 3)
    assert e[0].tolist() == [[0, 1, 0], [1, 0, 0], [0, 0, 0]] and e[1].sum() == 6
    info = load_dataset_info('qm9_with_h')
    assert info['max_n_nodes'] == 29 and sum(info['train_n_nodes'].values()) == 100000
    torch.manual_seed(0)
    s = get_node_dist(info).sample(1000)
    assert 3 <= int(s.min()) and int(s.max()) <= 29 and abs(s.float().mean().item() - 18.0) < 0.5
    assert load_dataset_info('geom_with_h_1')['max_n_nodes'] == 181


def test_post_process_thresholds():
    cfg = configs.get('vpsde_geom_uncond_jodo')
    inv = get_data_inverse_scaler(cfg)
    nm, em = build_masks([2], 2, 'cpu')
    xh = torch.zeros(1, 2, 3 + 17)
    xh[0, 0, 3 + 4] = 1.0          # atom type 4
    xh[0, 1, 3 + 9] = 1.0
    xh[0, 0, -1] = 0.26            # charge: *4 -> 1.04 -> 1
    ex = torch.full((1, 2, 2, 3), -1.0)
    ex[0, 0, 1] = ex[0, 1, 0] = torch.tensor([1.0, 0.0, -1.0])       # exists, order (0+1)/2*3 = 1.5 -> 2
    pos, one_hot, fc, et = post_process(xh, 16, True, nm, inv, ex, em, True)
    assert one_hot.argmax(2).tolist() == [[4, 9]] and fc[0, 0, 0].item() == 1
    assert et[0].tolist() == [[0, 2], [2, 0]]
    ex[0, 0, 1] = ex[0, 1, 0] = torch.tensor([1.0, -1.0, 1.0])      # exists, no order, aromatic -> 4
    _, _, _, et = post_process(xh, 16, True, nm, inv, ex, em, True)
    assert et[0].tolist() == [[0, 4], [4, 0]]


def test_philox_known_answers():
    """oracle/philox_ref.py (the checker of the in-kernel noise draws) against the published Random123 known-answer vectors of
    philox4x32-10, and the structure of the noise it builds from them."""
    import numpy as np
    from oracle import philox_ref as PR
    kat = [([0, 0, 0, 0], [0, 0], '6627e8d5 e169c58d bc57ac4c 9b00dbd8'),
           ([0xffffffff] * 4, [0xffffffff] * 2, '408f276d 41c83b0e a20bc7c6 6d5451fd'),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], 'd16cfe09 94fdcceb 5001e420 24126ea1')]
    for ctr, key, want in kat:
        got = PR.philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32))
        assert ' '.join('%08x' % v for v in got) == want
    z = PR.normal4(12345, 3, 0, np.arange(100000))
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01 and np.isfinite(z).all()
    n_nodes = [3, 5, 1]
    x = PR.node_noise(7, 1, n_nodes, 5, 6)
    assert np.abs(x[:, :, :3].sum(1)).max() < 1e-6 and x[0, 3:].any() == False and x[2, 1:].any() == False
    e = PR.edge_noise(7, 1, n_nodes, 5, 2)
    assert np.array_equal(e, e.transpose(0, 2, 1, 3)) and np.abs(e[:, np.arange(5), np.arange(5)]).max() == 0
    assert e[2].any() == False and e[0, 3:].any() == False


def test_plan_options_are_range_checked():
    import ctypes
    import numpy as np
    from jodo_amd import capi
    from jodo_amd.models.dgt import _DGTBase
    lib = capi.lib()
    cfg = _DGTBase._Cfg(256, 8, 16, 2, 2, 6, 2, 0, 2.0, 0.0, 0)
    n = np.asarray([5, 9], dtype=np.int32)
    h = ctypes.c_void_p()
    assert lib.jodo_plan_create(ctypes.byref(cfg), 2, 9, n.ctypes.data_as(ctypes.c_void_p), 0, ctypes.byref(h)) == 0
    try:
        for opt, good, bad in ((0, (0, 1), (2, -1)), (1, (0, 1), (7,)), (2, (0, 1, 2, 4, 12, 14), (3, 5)), (3, (0, 1, 2, 3), (4, -1))):
            for v in good:
                assert lib.jodo_plan_set_option(h, opt, v) == 0
            for v in bad:
                assert lib.jodo_plan_set_option(h, opt, v) == -1 and b'plan_set_option' in lib.jodo_last_error()
        assert lib.jodo_plan_set_option(h, 99, 0) == -1
    finally:
        lib.jodo_plan_destroy(h)


def test_scaling_prediction_runs_on_the_host():
    """jodo_amd/scaling.py: per-rank executed work of a sharded run from the library's own work model, no GPU.  Equal shares
    predict efficiency 1; a share with one large molecule more predicts less; LPT dealing is never worse than contiguous slices."""
    import torch
    from jodo_amd import scaling
    from jodo_amd.models import get_model_class, load_dataset_info, get_node_dist
    from helpers import make_config
    cfg = make_config('vpsde_geom_uncond_jodo')
    cs = get_model_class(cfg.model.name)(cfg)._cfg()
    same = scaling.predict_weak(cs, [[40, 30, 20, 10]] * 4, 1)
    assert same['ranks'] == 4 and abs(same['predicted_efficiency'] - 1.0) < 1e-12
    skew = scaling.predict_weak(cs, [[40, 30, 20, 10], [150, 30, 20, 10]], 1)
    assert skew['predicted_efficiency'] < 0.9 and skew['sum_n2_per_rank'][1] > skew['sum_n2_per_rank'][0]
    torch.manual_seed(3)
    n_all = get_node_dist(load_dataset_info('geom_with_h_1')).sample(256).tolist()
    dealt = scaling.predict_dealt(cs, n_all, 4, 64, 1)
    assert dealt['lpt']['predicted_efficiency'] >= dealt['contiguous']['predicted_efficiency'] - 1e-3
    assert sum(dealt['lpt']['molecules_per_rank']) == 256 and sum(dealt['contiguous']['molecules_per_rank']) == 256


def test_flatten_parameters_keeps_values_order_and_views():
    """jodo_amd/optim.py flatten_parameters (host logic of the flat optimiser): every parameter becomes a slice of one buffer in
    registration order, values kept; flat_view recognises exactly that layout."""
    import torch
    from jodo_amd import optim as JO
    g = torch.Generator().manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in [(3, 5), (7,), (2, 2, 2)]]
    want = [p.detach().clone() for p in ps]
    assert JO.flat_view([p.data for p in ps]) is None
    flat = JO.flatten_parameters(ps)
    assert JO.slice_offsets([15, 7, 8]) == ([0, 16, 24], 32)                       # every slice on a 16-byte boundary
    assert flat.numel() == 32 and all(torch.equal(p.data, w) for p, w in zip(ps, want))
    assert torch.equal(flat[:15], want[0].reshape(-1)) and float(flat[15]) == 0.0 and torch.equal(flat[16:23], want[1]) and float(flat[23]) == 0.0
    again = JO.flat_view([p.data for p in ps])
    assert again is not None and again.data_ptr() == flat.data_ptr()
    assert JO.flatten_parameters(ps).data_ptr() == flat.data_ptr()                 # already flat: left where they are
    flat.mul_(2.0)
    assert all(torch.equal(p.data, 2.0 * w) for p, w in zip(ps, want))            # views, not copies
    assert JO.flat_view([ps[1].data, ps[0].data, ps[2].data]) is None             # order matters
    assert JO.flat_view([p.data for p in ps[:2]]) is None                         # must tile the whole storage
