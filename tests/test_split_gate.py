"""The opt-in split-bf16 (three-term) MFMA form of a projection (csrc/dgt_split.h): packing on the CPU, arithmetic on the GPU.

w = hi + mid + lo in bf16 is exact for every finite float; the six retained products accumulated in fp32 must be as good as the
exact-fp32 MFMA chain the product path runs (gate of the round-5 review: not worse than 2 x its error against float64)."""
import ctypes

import numpy as np
import pytest
import torch

from jodo_amd import capi


def _pack(W):
    n_out, n_in = W.shape
    f = np.zeros(n_out * n_in, dtype=np.float32)
    s = np.zeros(n_out * n_in * 3, dtype=np.uint16)
    Wc = np.ascontiguousarray(W, dtype=np.float32)
    capi.check(capi.lib().jodo_debug_pack_split(Wc.ctypes.data_as(ctypes.c_void_p), n_out, n_in, f.ctypes.data_as(ctypes.c_void_p),
                                                s.ctypes.data_as(ctypes.c_void_p)), 'pack_split')
    return f, s


def test_split_packing_is_exact_and_a_permutation_of_the_f32_packing():
    rng = np.random.default_rng(3)
    W = (rng.standard_normal((64, 96)) * 10.0 ** rng.uniform(-6, 6, size=(64, 96))).astype(np.float32)
    f, s = _pack(W)
    b2f = lambda u: (u.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    t = s.reshape(2, 6, 3, 64, 8)                                # [out block][K16 step][term][lane][j]
    rec = b2f(t[:, :, 0]) + b2f(t[:, :, 1]) + b2f(t[:, :, 2])    # hi + mid + lo, summed in float64: exact
    # lane l of step G, element j reads the f32 packing's k-step 8 G + j of the same lane: [block][quad = kstep / 4][lane][kstep % 4]
    f4 = f.reshape(2, 12, 64, 4)
    want = np.empty_like(rec)
    for G in range(6):
        for j in range(8):
            k = 8 * G + j
            want[:, G, :, j] = f4[:, k // 4, :, k % 4]
    assert np.array_equal(rec, want.astype(np.float64))
    assert capi.lib().jodo_debug_pack_split(W.ctypes.data_as(ctypes.c_void_p), 60, 96, None, None) != 0      # sizes are checked


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 33, 4096])
def test_split_projection_is_as_good_as_the_fp32_chain(rows):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 256, generator=g)
    W = (torch.rand(256, 256, generator=g) * 2 - 1) / 16
    y64 = x.double() @ W.double().t()
    f, s = _pack(W.numpy())
    wf, ws = torch.from_numpy(f).cuda(), torch.from_numpy(s.view(np.int16)).cuda()
    xd = x.cuda()
    err = {}
    for mode in (0, 1, 2):
        y = torch.zeros(rows, 256, device='cuda')
        capi.check(capi.lib().jodo_debug_chain(mode, 256, 1, capi.ptr(xd), rows, capi.ptr(wf), capi.ptr(ws), capi.ptr(y), 1, None,
                                               capi.current_stream_ptr()), 'debug_chain')
        torch.cuda.synchronize()
        err[mode] = float((y.cpu().double() - y64).abs().max())
    assert err[0] < 5e-6                                          # the product form: a K = 256 fp32 fma chain
    assert err[1] <= 2.0 * err[0] and err[2] <= 2.0 * err[0], err


def test_split_tape_is_the_consumption_order_of_the_pair_update():
    """jodo_dgt_pack_split_host: per block [for every hidden chunk c: ff_linear3 output blocks 2c, 2c + 1 | ff_linear4 output blocks, steps
    4c .. 4c + 3] [readout] [L blocks, shortest first] as K16 steps of 3 KiB — checked slice by slice against the generic split packing
    of the same matrices (jodo_debug_pack_split, natural maps)."""
    from jodo_amd import configs
    from jodo_amd.models import get_model_class, deterministic_init_
    cfg = configs.get('vpsde_qm9_uncond_jodo')
    model = deterministic_init_(get_model_class('DGT_concat')(cfg), seed=3)
    sd = model.state_dict()
    both = capi.pack_split_tape(model._cfg(), sd).numpy().view(np.uint16)
    De, r, L = 64, cfg.model.mlp_ratio, cfg.model.n_layers
    STEP = 3 * 64 * 8
    NCH, NSE, NE, NB2 = r * De // 64, De // 16, De // 32, 2 * (De // 32)
    steps = NCH * (2 * NSE + NE * 4) + NSE + sum(2 * (NB2 - b) for b in range(NB2))
    total, pair_b, node_b, attn_b = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    size = lambda m: capi.check(capi.lib().jodo_dgt_split_size(ctypes.byref(m._cfg()), ctypes.byref(total), ctypes.byref(pair_b), ctypes.byref(node_b),
                                                                ctypes.byref(attn_b)), 'split_size')
    size(model)
    nsteps = 2 * 16 + r * 4 * (2 * 16 + 8 * 4) + 16 * 16 + 2 * 16 + 24 * 16          # the node tape of the tuned nf 256 kernel set
    asteps = (12 + 13) * 4                                                           # the two (cyclic) attention tapes: one per launch
    assert steps == 56 and pair_b.value == steps * 3072 and node_b.value == nsteps * 3072 and attn_b.value == asteps * 3072
    assert both.size * 2 == total.value == L * (pair_b.value + node_b.value + attn_b.value)
    tape = both[:L * steps * STEP]
    # attention tape of block 1: edge_emb ([G ; e] columns as they are: G first), lin_edge0 in the tuned q / k arrangement, lin_edge1
    attn = both[L * (steps + nsteps) * STEP:].reshape(L, asteps, STEP)
    _, see = _pack(sd['e_block_1.edge_emb.weight'].numpy())
    see = see.reshape(2, 8, STEP)                                                    # G halves of both blocks first, then the e halves
    ee = np.concatenate([see[0, :4], see[1, :4], see[0, 4:], see[1, 4:]])
    _, sl1 = _pack(sd['e_block_1.attn_mpnn.lin_edge1.weight'].numpy())
    sl1 = sl1.reshape(8, 4, STEP)
    # first launch: edge_emb | lin_edge0 blocks 0 - 2 + tail | lin_edge1 blocks 0 - 3; second: edge_emb | lin_edge0 3 - 6 + tail | lin_edge1 4 - 7
    assert np.array_equal(attn[1, :16], ee) and np.array_equal(attn[1, 48:64], ee)
    assert np.array_equal(attn[1, 32:48], sl1[:4].reshape(16, STEP)) and np.array_equal(attn[1, 84:], sl1[4:].reshape(16, STEP))
    assert np.array_equal(attn[1, 28:32], attn[1, 80:84])                             # the tail block of lin_edge0 closes both score sections
    # node tape of block 0: node2edge_lin first, the next block's lin_value last; the last block's q / k / v section is zero
    node = both[L * steps * STEP:L * (steps + nsteps) * STEP].reshape(L, nsteps, STEP)
    _, sn2e = _pack(sd['e_block_0.node2edge_lin.weight'].numpy())
    assert np.array_equal(node[0, :32], sn2e.reshape(2 * 16, STEP))
    _, sv = _pack(sd['e_block_1.attn_mpnn.lin_value.weight'].numpy())
    assert np.array_equal(node[0, nsteps - 8 * 16:], sv.reshape(8 * 16, STEP))
    _, sf1 = _pack(sd['e_block_0.ff_linear1.weight'].numpy())
    _, sf2 = _pack(sd['e_block_0.ff_linear2.weight'].numpy())
    sf1, sf2 = sf1.reshape(r * 8, 16, STEP), sf2.reshape(8, r * 16, STEP)
    c = 1                                                                            # second hidden chunk: ff1 blocks 2, 3, then ff2 steps 4 .. 7 of every block
    base = 32 + c * 64
    assert np.array_equal(node[0, base:base + 16], sf1[2]) and np.array_equal(node[0, base + 16:base + 32], sf1[3])
    assert np.array_equal(node[0, base + 32 + 4 * 5:base + 32 + 4 * 6], sf2[5, 4:8])
    assert not node[L - 1, nsteps - 24 * 16:].any()
    for l in (0, L - 1):
        blk = tape[l * steps * STEP:(l + 1) * steps * STEP].reshape(steps, STEP)
        _, s3 = _pack(sd['e_block_%d.ff_linear3.weight' % l].numpy())
        _, s4 = _pack(sd['e_block_%d.ff_linear4.weight' % l].numpy())
        s3 = s3.reshape(r * De // 32, NSE, STEP)
        s4 = s4.reshape(NE, r * De // 16, STEP)
        at = 0
        for c in range(NCH):
            for b2 in range(2):
                assert np.array_equal(blk[at:at + NSE], s3[2 * c + b2]); at += NSE
            for ob in range(NE):
                assert np.array_equal(blk[at:at + 4], s4[ob, 4 * c:4 * c + 4]); at += 4
        # readout: edge_l [2 De / L = 16, De], rows padded to one 32-row block
        wro = np.zeros((32, De), dtype=np.float32)
        w = sd['edge_%d.weight' % l].numpy()
        wro[:w.shape[0]] = w
        _, sro = _pack(wro)
        assert np.array_equal(blk[at:at + NSE], sro.reshape(1, NSE, STEP)[0]); at += NSE
        assert steps - at == 20                                   # L: 2 + 4 + 6 + 8 steps
    # configurations the split form is not built for are refused by name
    cfg128 = configs.get('vpsde_geom_uncond_jodo')
    cfg128.model.nf, cfg128.model.n_layers = 128, 6
    with pytest.raises(capi.JodoHipError, match='nf = 256 / 384'):
        capi.pack_split_tape(get_model_class('DGT_concat')(cfg128)._cfg(), {})
    with pytest.raises(capi.JodoHipError, match='unconditional'):
        capi.pack_split_tape(get_model_class('cond_DGT_concat')(configs.get('vpsde_qm9_cond_jodo'))._cfg(), {})
    # nf 384: a pair tape only (its node kernels are the width-generic set): 192 steps per block at mlp_ratio 4
    cfg384 = configs.get('vpsde_geom_uncond_jodo')
    cfg384.model.nf = 384
    m384 = get_model_class('DGT_concat')(cfg384)
    size(m384)
    assert pair_b.value == 192 * 3072 and node_b.value == 0 and attn_b.value == 0 and total.value == cfg384.model.n_layers * pair_b.value


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name,n_nodes,over", [
    ('vpsde_qm9_uncond_jodo', [3, 9, 17, 29, 12, 5, 1, 2, 28, 29, 29, 18, 7] * 3, {}),                 # r = 2, several strips, idle waves
    ('vpsde_geom_uncond_jodo', [44, 45, 7, 70, 33], {}),                                                # r = 4, L = 10
    ('vpsde_geom_uncond_jodo', [44, 45, 7, 1, 2, 61], dict(nf=384)),                                    # nf 384: pair update only, one wave per SIMD
    # >= 1024 node strips: the node kernel also produces the next block's q / k / v (fuse_next), one full round + a remainder
    ('vpsde_qm9_uncond_jodo', ([29, 17, 23, 12, 9, 28, 19, 21, 18, 20] * 181)[:1805], {}),
])
@pytest.mark.parametrize("mode", [True, 'attention'])
def test_split_bf16_pair_update_against_the_default_path_and_float64(cfg_name, n_nodes, over, mode):
    """JODO_OPT_SPLIT_BF16 (opt-in, model.split_bf16 = True; takes effect under pin_paths): the folded pair update with split-bf16
    projections.  (a) it really runs (outputs differ from the exact-fp32 path in the last bits), (b) it stays within a small multiple of
    fp32 rounding of the default path, (c) it is held to the float64 oracle at the SAME stated tolerance / K64 as the default path,
    first-step and self-conditioned evaluation."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import K64, close64, make_config, make_model, oracle_32_64, random_inputs, state_dict_cpu
    from oracle import dgt_oracle as O
    DEV = 'cuda:0'
    cfg = make_config(cfg_name, **over)
    hp = O.Hyper.from_config(cfg)
    xh, ex, nl, ctx, nm, em = random_inputs(hp, n_nodes, seed=23)
    nl = torch.full_like(nl, 0.3)                              # one noise level: the shared modulation row of sampling
    d = lambda x: None if x is None else x.to(DEV)
    nmd, emd = d(nm), d(em)

    def run(model, cx, cex):
        with torch.no_grad():
            o = model(d(nl), d(xh), nmd, emd, edge_x=d(ex), cond_x=d(cx), cond_edge_x=d(cex), noise_level=d(nl))
        torch.cuda.synchronize()
        return o[0].cpu(), o[1].cpu()

    if mode == 'attention' and (over.get('nf', 256) != 256 or len(n_nodes) > 100):
        pytest.skip("the attention variant (JODO_OPT_SPLIT_BF16 = 2, experiments: two launches by heads) is built for the tuned nf 256 set; one small case each")
    outs = {}
    for split in (False, True):
        model = make_model(cfg, 13, DEV)
        model.split_bf16 = mode if split else False
        first = run(model, None, None)
        run(model, first[0], first[1])
        model.pin_paths()                                      # what a sampler does after its first self-conditioned evaluation
        assert model._last_plan.get('pinned') and (('split_tape' in model._last_plan) == split)
        o2 = run(model, first[0], first[1])
        o1 = run(model, None, None)
        assert model.take_nan_count() == 0
        outs[split] = (o1, o2)
        if not split:
            sd = state_dict_cpu(model)
        elif mode is True and len(n_nodes) < 100:
            # weights changed under the pinned plan (an optimiser step between two calls on the same masks): the split tape follows the
            # packed blob — the perturbed model's split outputs equal a freshly built perturbed model's, not the stale tape's
            with torch.no_grad():
                for p_ in model.parameters():
                    p_.mul_(1.0 + 1e-3)
            moved = run(model, first[0], first[1])
            fresh = make_model(cfg, 13, DEV)
            with torch.no_grad():
                for p_ in fresh.parameters():
                    p_.mul_(1.0 + 1e-3)
            fresh.split_bf16 = True
            run(fresh, None, None); run(fresh, first[0], first[1]); fresh.pin_paths()
            want = run(fresh, first[0], first[1])
            assert torch.equal(moved[0], want[0]) and torch.equal(moved[1], want[1])
            with torch.no_grad():
                for p_ in model.parameters():
                    p_.div_(1.0 + 1e-3)
    for k in (0, 1):
        for j in (0, 1):
            a, b = outs[True][k][j], outs[False][k][j]
            assert not torch.equal(a, b), "the split kernel did not run"
            assert float((a - b).abs().max()) <= 2e-5 + 1e-4 * float(b.abs().max()), float((a - b).abs().max())
    # the oracle on (at most) the first 40 molecules: outputs are batch-independent
    nb = min(len(n_nodes), 40)
    Ns = max(n_nodes[:nb])
    from helpers import masks
    nms, ems = masks(n_nodes[:nb])
    cut = lambda t, k: t[:nb, :Ns] if k == 1 else t[:nb, :Ns, :Ns]
    r1 = oracle_32_64(sd, hp, cut(xh, 1), nms, ems, cut(ex, 2), None, None, nl[:nb])
    r2 = oracle_32_64(sd, hp, cut(xh, 1), nms, ems, cut(ex, 2), cut(outs[False][0][0], 1), cut(outs[False][0][1], 2), nl[:nb])
    for step, (r32, r64) in ((0, r1), (1, r2)):
        for split in (False, True):
            close64(cut(outs[split][step][0], 1), r32[0], r64[0], 'split_bf16=%s step %d nodes' % (split, step + 1), k=K64)
            close64(cut(outs[split][step][1], 2), r32[1], r64[1], 'split_bf16=%s step %d edges' % (split, step + 1), k=K64)
