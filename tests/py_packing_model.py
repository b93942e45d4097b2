"""TEST INFRASTRUCTURE: independent Python restatement of the C weight packer (csrc/dgt_pack.cpp), blob for blob.

Pack a DGT state_dict (reference parameter names, models/mol_gnn.py:414-489 / :601-684) into the
single float32 blob + offset table consumed by `jodo_dgt_forward`.

Slot order must match `enum jodo_wslot_global` / `enum jodo_wslot_block` in include/jodo_hip.h
(tests/test_packing.py parses the header and checks it).
"""
import numpy as np

import py_packing as P
from jodo_amd.models.dims import ModelDims  # noqa: F401

GLOBAL_SLOTS = [
    'TIME_FREQ', 'TIME_W1', 'TIME_B1', 'TIME_W3', 'TIME_B3',
    'COND_W0', 'COND_B0', 'COND_W2', 'COND_B2', 'COND_LIN_W', 'COND_LIN_B',
    'MOD_W', 'MOD_B',
    'NODE_EMB_W', 'NODE_EMB_B', 'EDGE_EMB_W', 'EDGE_EMB_B', 'GBF_TOP',
    'NH1_W', 'NH1_B', 'NH2_W', 'NH2_B', 'NH3_W', 'NH3_B',
    'EH1_W', 'EH1_B', 'EH2_W', 'EH2_B', 'EH3_W', 'EH3_B',
]
BLOCK_SLOTS = [
    'WQ', 'BQ', 'WK', 'BK', 'WV', 'BV',
    'EE_W', 'EE_B', 'LE0_W', 'LE1_W', 'N2E_W', 'N2E_B',
    'FF1_W', 'FF1_B', 'FF2_W', 'FF2_B', 'FF3_W', 'FF3_B', 'FF4_W', 'FF4_B',
    'INE_W', 'ROW_W', 'COL_W', 'IN_B', 'C0_W', 'C0_B', 'C2_W', 'CSCALE',
    'NRO_W', 'NRO_B', 'ERO_W', 'ERO_B', 'GBF',
    'ROWQ_W', 'COLQ_W', 'INQ_B', 'LQ_W', 'INEC_W', 'QT_W',
]


def _np(t):
    return t.detach().cpu().numpy().astype(np.float32) if hasattr(t, 'detach') else np.asarray(t, np.float32)


def _gbf_table(sd, prefix, De):
    """[3][De]: mu, 1/sigma, 1/(sqrt(2*3.14159)*sigma); slot 0 is feature x' itself (unused entries)."""
    mu = _np(sd[prefix + '.means.weight']).reshape(-1).astype(np.float64)
    sg = np.abs(_np(sd[prefix + '.stds.weight']).reshape(-1).astype(np.float64)) + 1e-5
    a = (2 * 3.14159) ** 0.5
    tab = np.zeros((3, De), np.float64)
    tab[0, 1:] = mu
    tab[1, 1:] = 1.0 / sg
    tab[2, 1:] = 1.0 / (a * sg)
    tab[1, 0] = 1.0
    return tab.astype(np.float32).reshape(-1)


def rot_stats(Win, bin_, D, De):
    """Rotated LayerNorm statistics of equi_update (csrc/dgt_pack.cpp rot_stats): Householder QR of the centred [e ; G]
    columns of input_lin, Q (P W_eg) = [L ; 0]; returns (Q P W_row, Q P W_col, Q P b, L, P W_eg, Q^T) in float64."""
    KL = 2 * De
    Win = Win.astype(np.float64)
    A = Win[:, 2 * D:2 * D + KL].copy()
    A -= A.mean(axis=0, keepdims=True)
    wec = A.copy()
    Q = np.eye(D)
    for k in range(KL):
        x = A[k:, k]
        nrm = np.sqrt((x * x).sum())
        if nrm == 0.0:
            continue
        alpha = -nrm if x[0] > 0.0 else nrm
        v = x.copy()
        v[0] -= alpha
        vn = (v * v).sum()
        if vn == 0.0:
            continue
        A[k:, k:] -= np.outer(v, (2.0 / vn) * (v @ A[k:, k:]))
        Q[k:, :] -= np.outer((2.0 / vn) * v, v @ Q[k:, :])
    L = np.triu(A[:KL, :])
    QP = Q - Q.mean(axis=1, keepdims=True)
    return QP @ Win[:, :D], QP @ Win[:, D:2 * D], QP @ bin_.astype(np.float64), L, wec, Q.T.copy()


def _hid_in_map(base_w, n_layers, per_block_true, per_block_pad):
    """Input map of a head MLP's first layer: padded activation layout
    [base_w | L x per_block_pad] -> true column base_w + l*per_block_true + c (or -1)."""
    width = base_w + n_layers * per_block_pad
    col = np.full(width, -1, dtype=np.int64)
    col[:base_w] = np.arange(base_w)
    for l in range(n_layers):
        col[base_w + l * per_block_pad: base_w + l * per_block_pad + per_block_true] = \
            base_w + l * per_block_true + np.arange(per_block_true)
    nat = P.natural_in_map(width)
    return col[nat]


def pack_model(sd, dims):
    """sd: state_dict with the reference's key names (no 'module.' prefix).
    Returns (blob float32 [n], woff int64 [n_slots])."""
    d = dims
    D, De, T, L = d.D, d.De, d.T, d.L
    chunks, offs = [], {}
    cursor = [0]

    def put(name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        offs[name] = cursor[0]
        chunks.append(arr)
        pad = (-arr.size) % 64                     # keep every slot 256-byte aligned
        if pad:
            chunks.append(np.zeros(pad, np.float32))
        cursor[0] += arr.size + pad

    nat = P.natural_in_map
    nout = P.natural_out_map
    w = lambda k: _np(sd[k])

    # ---- time embedding ----
    put('TIME_FREQ', w('time_mlp.0.weights'))
    put('TIME_W1', w('time_mlp.1.weight'))                       # raw [T,17]
    put('TIME_B1', w('time_mlp.1.bias'))
    put('TIME_W3', P.pack_projection(w('time_mlp.3.weight'), nat(T), nout(T)))
    put('TIME_B3', w('time_mlp.3.bias'))
    if d.cond_ch > 0:
        put('COND_W0', w('cond_mlp.0.weight').reshape(-1))       # [D,1]
        put('COND_B0', w('cond_mlp.0.bias'))
        put('COND_W2', P.pack_projection(w('cond_mlp.2.weight'), nat(D), nout(D)))
        put('COND_B2', w('cond_mlp.2.bias'))
        kc = d.cond_ch * D
        assert kc % 64 == 0
        put('COND_LIN_W', P.pack_projection(w('cond_lin.weight'), nat(kc), nout(T)))
        put('COND_LIN_B', w('cond_lin.bias'))
    else:
        for s in ('COND_W0', 'COND_B0', 'COND_W2', 'COND_B2', 'COND_LIN_W', 'COND_LIN_B'):
            put(s, np.zeros(1, np.float32))

    # ---- all time-modulation linears fused into one [Mtot, T] projection ----
    Wm = np.zeros((d.Mtot, T), np.float32)
    bm = np.zeros(d.Mtot, np.float32)
    Wm[0:2] = w('dist_layer.time_mlp.1.weight')
    bm[0:2] = w('dist_layer.time_mlp.1.bias')
    for l in range(L):
        b = 'e_block_%d' % l
        o = 32 + l * d.MB
        Wm[o:o + 6 * D] = w(b + '.node_time_mlp.1.weight'); bm[o:o + 6 * D] = w(b + '.node_time_mlp.1.bias')
        o += 6 * D
        Wm[o:o + 6 * De] = w(b + '.edge_time_mlp.1.weight'); bm[o:o + 6 * De] = w(b + '.edge_time_mlp.1.bias')
        o += 6 * De
        Wm[o:o + 2 * D] = w(b + '.equi_update.time_mlp.1.weight'); bm[o:o + 2 * D] = w(b + '.equi_update.time_mlp.1.bias')
        o += 2 * D
        Wm[o:o + 2] = w(b + '.dist_layer.time_mlp.1.weight'); bm[o:o + 2] = w(b + '.dist_layer.time_mlp.1.bias')
        o += 32
        # y = coord_mlp.0(LN(pre) (1 + sc) + sh) = [W0 (pre (1 + sc)) - mean W0 (1 + sc)] rstd + (W0 sh + b0): the two
        # per-molecule vectors are affine in SiLU(time_emb) like every other modulation output -> two more row groups
        W0 = w(b + '.equi_update.coord_mlp.0.weight').astype(np.float64)
        Wt = w(b + '.equi_update.time_mlp.1.weight').astype(np.float64)          # rows: shift [D] | scale [D]
        bt = w(b + '.equi_update.time_mlp.1.bias').astype(np.float64)
        Wm[o:o + D] = (W0 @ Wt[D:2 * D]).astype(np.float32); bm[o:o + D] = (W0 @ (1.0 + bt[D:2 * D])).astype(np.float32)
        o += D
        Wm[o:o + D] = (W0 @ Wt[0:D]).astype(np.float32)
        bm[o:o + D] = (W0 @ bt[0:D] + w(b + '.equi_update.coord_mlp.0.bias').astype(np.float64)).astype(np.float32)
    put('MOD_W', P.pack_projection(Wm, nat(T), nout(d.Mtot)))
    put('MOD_B', bm)

    # ---- embeddings ----
    put('NODE_EMB_W', P.pack_projection(w('node_emb.weight'), P.small_in_map(2 * d.nd), nout(D)))
    put('NODE_EMB_B', w('node_emb.bias'))
    ee_in = P.concat_in_maps(nat(De) + 2 * d.ch, P.small_in_map(2 * d.ch))   # [G0 ; raw edge inputs]
    put('EDGE_EMB_W', P.pack_projection(w('edge_emb.weight'), ee_in, nout(De)))
    put('EDGE_EMB_B', w('edge_emb.bias'))
    put('GBF_TOP', _gbf_table(sd, 'dist_layer', De))

    # ---- heads ----
    put('NH1_W', P.pack_projection(w('node_pred_mlp.0.weight'), _hid_in_map(D, L, d.cn, d.cnp), nout(D)))
    put('NH1_B', w('node_pred_mlp.0.bias'))
    put('NH2_W', P.pack_projection(w('node_pred_mlp.2.weight'), nat(D), nout(D // 2)))
    put('NH2_B', w('node_pred_mlp.2.bias'))
    om = nout(32, n_valid=d.nd)
    put('NH3_W', P.pack_projection(w('node_pred_mlp.4.weight'), nat(D // 2), om))
    put('NH3_B', P.pack_vector(w('node_pred_mlp.4.bias'), om))
    eh_in = _hid_in_map(De, L, d.ce, d.cep)
    W1 = np.concatenate([w('edge_exist_mlp.0.weight'), w('edge_type_mlp.0.weight')], axis=0)     # [2De, cat]
    put('EH1_W', P.pack_projection(W1, eh_in, nout(2 * De)))
    put('EH1_B', np.concatenate([w('edge_exist_mlp.0.bias'), w('edge_type_mlp.0.bias')]))
    h2 = De // 2
    W2 = np.zeros((2 * h2, 2 * De), np.float32)
    W2[:h2, :De] = w('edge_exist_mlp.2.weight')
    W2[h2:, De:] = w('edge_type_mlp.2.weight')
    put('EH2_W', P.pack_projection(W2, nat(2 * De), nout(2 * h2)))
    put('EH2_B', np.concatenate([w('edge_exist_mlp.2.bias'), w('edge_type_mlp.2.bias')]))
    W3 = np.zeros((d.ch, 2 * h2), np.float32)
    W3[0, :h2] = w('edge_exist_mlp.4.weight')[0]
    W3[1:, h2:] = w('edge_type_mlp.4.weight')
    om3 = nout(32, n_valid=d.ch)
    put('EH3_W', P.pack_projection(W3, nat(2 * h2), om3))
    put('EH3_B', P.pack_vector(np.concatenate([w('edge_exist_mlp.4.bias'), w('edge_type_mlp.4.bias')]), om3))

    # ---- blocks ----
    qk = P.qk_out_map_wide(d.SH, d.SC) if d.wide else P.qk_out_map(d.SH, d.SC)
    assert qk.shape[0] * 32 == d.QKP
    for l in range(L):
        b = 'e_block_%d' % l
        a = b + '.attn_mpnn'
        pre = 'B%d_' % l
        put(pre + 'WQ', P.pack_projection(w(a + '.lin_query.weight'), nat(D), qk))
        put(pre + 'BQ', P.pack_vector(w(a + '.lin_query.bias'), qk))
        put(pre + 'WK', P.pack_projection(w(a + '.lin_key.weight'), nat(D), qk))
        put(pre + 'BK', P.pack_vector(w(a + '.lin_key.bias'), qk))
        put(pre + 'WV', P.pack_projection(w(a + '.lin_value.weight'), nat(D), nout(D)))
        put(pre + 'BV', w(a + '.lin_value.bias'))
        put(pre + 'EE_W', P.pack_projection(w(b + '.edge_emb.weight'), P.concat_in_maps(nat(De), nat(De) + De), nout(De)))
        put(pre + 'EE_B', w(b + '.edge_emb.bias'))
        put(pre + 'LE0_W', P.pack_projection(w(a + '.lin_edge0.weight'), nat(De), qk))
        put(pre + 'LE1_W', P.pack_projection(w(a + '.lin_edge1.weight'), nat(De), nout(D)))
        put(pre + 'N2E_W', P.pack_projection(w(b + '.node2edge_lin.weight'), nat(D), nout(De)))
        put(pre + 'N2E_B', w(b + '.node2edge_lin.bias'))
        put(pre + 'FF1_W', P.pack_projection(w(b + '.ff_linear1.weight'), nat(D), nout(d.r * D)))
        put(pre + 'FF1_B', w(b + '.ff_linear1.bias'))
        put(pre + 'FF2_W', P.pack_projection(w(b + '.ff_linear2.weight'), nat(d.r * D), nout(D)))
        put(pre + 'FF2_B', w(b + '.ff_linear2.bias'))
        put(pre + 'FF3_W', P.pack_projection(w(b + '.ff_linear3.weight'), nat(De), nout(d.r * De)))
        put(pre + 'FF3_B', w(b + '.ff_linear3.bias'))
        put(pre + 'FF4_W', P.pack_projection(w(b + '.ff_linear4.weight'), nat(d.r * De), nout(De)))
        put(pre + 'FF4_B', w(b + '.ff_linear4.bias'))
        Win = w(b + '.equi_update.input_lin.weight')                         # [D, 2D + De + De]: h_row|h_col|e|G
        ine = P.concat_in_maps(nat(De) + 2 * D, nat(De) + 2 * D + De)        # kernel feeds [e ; G]
        put(pre + 'INE_W', P.pack_projection(Win, ine, nout(D)))
        put(pre + 'ROW_W', P.pack_projection(Win[:, :D], nat(D), nout(D)))
        put(pre + 'COL_W', P.pack_projection(Win[:, D:2 * D], nat(D), nout(D)))
        put(pre + 'IN_B', w(b + '.equi_update.input_lin.bias'))
        put(pre + 'C0_W', P.pack_projection(w(b + '.equi_update.coord_mlp.0.weight'), nat(D), nout(D)))
        put(pre + 'C0_B', w(b + '.equi_update.coord_mlp.0.bias'))
        put(pre + 'C2_W', w(b + '.equi_update.coord_mlp.2.weight'))           # raw [3, D]
        put(pre + 'CSCALE', w(b + '.equi_update.coord_norm.scale'))
        omn = nout(d.cnp, n_valid=d.cn)
        put(pre + 'NRO_W', P.pack_projection(w('node_%d.weight' % l), nat(D), omn))
        put(pre + 'NRO_B', P.pack_vector(w('node_%d.bias' % l), omn))
        ome = nout(32, n_valid=d.ce)
        put(pre + 'ERO_W', P.pack_projection(w('edge_%d.weight' % l), nat(De), ome))
        put(pre + 'ERO_B', P.pack_vector(w('edge_%d.bias' % l), ome))
        put(pre + 'GBF', _gbf_table(sd, b + '.dist_layer', De))
        rowq, colq, bq, Lq, wec, qt = rot_stats(Win, w(b + '.equi_update.input_lin.bias'), D, De)
        eg = P.concat_in_maps(nat(De), nat(De) + De)
        put(pre + 'ROWQ_W', P.pack_projection(rowq.astype(np.float32), nat(D), nout(D)))
        put(pre + 'COLQ_W', P.pack_projection(colq.astype(np.float32), nat(D), nout(D)))
        put(pre + 'INQ_B', bq.astype(np.float32))
        put(pre + 'LQ_W', P.pack_projection(Lq.astype(np.float32), eg, nout(2 * De)))
        put(pre + 'INEC_W', P.pack_projection(wec.astype(np.float32), eg, nout(D)))
        put(pre + 'QT_W', P.pack_projection(qt.astype(np.float32), nat(D), nout(D)))

    woff = [offs[s] for s in GLOBAL_SLOTS]
    for l in range(L):
        woff += [offs['B%d_%s' % (l, s)] for s in BLOCK_SLOTS]
    return np.concatenate(chunks), np.asarray(woff, dtype=np.int64)
