"""Stage-by-stage comparison of the HIP forward against the dense oracle (GPU box debugging aid)."""
import ctypes
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jodo_amd import configs, capi                                    # noqa: E402
from jodo_amd.models import get_model_class, deterministic_init_      # noqa: E402
from jodo_amd.sampling import build_masks                             # noqa: E402
from oracle import dgt_oracle as O                                    # noqa: E402


def fetch(model, what, n):
    plan = model._last_plan
    L = capi.lib()
    dst = torch.empty(n, device='cuda')
    cnt = ctypes.c_int64()
    capi.check(L.jodo_debug_fetch(plan['handle'], capi.ptr(plan['ws']), what, capi.ptr(dst), ctypes.byref(cnt),
                                  capi.current_stream_ptr()), 'fetch')
    torch.cuda.synchronize()
    return dst[:cnt.value].cpu()


def main(cfg_name='vpsde_qm9_uncond_jodo', L=8, n_nodes=(3, 9, 17, 29, 12, 5), gain=1.0, second=False, max_chunk=0):
    cfg = configs.get(cfg_name)
    cfg.model.n_layers = L
    dev = torch.device('cuda:0')
    name = cfg.model.name
    model = deterministic_init_(get_model_class(name)(cfg), seed=7, gain=gain).to(dev).eval()
    model.max_chunk = max_chunk
    hp = O.Hyper.from_config(cfg)
    n_nodes = list(n_nodes)
    B, N = len(n_nodes), max(n_nodes)
    g = torch.Generator().manual_seed(0)
    node_mask, edge_mask = build_masks(n_nodes, N, 'cpu')
    xh = torch.randn(B, N, 3 + hp.in_node_dim, generator=g) * node_mask
    ex = torch.randn(B, N, N, hp.edge_ch, generator=g)
    ex = (ex + ex.transpose(1, 2)) * edge_mask.reshape(B, N, N, 1)
    nl = torch.full((B,), 0.7)
    ctx = torch.randn(B, 1, generator=g) if hp.cond_ch else None
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cx = cex = None
    with torch.no_grad():
        if second:
            cx, cex = O.forward_dense(sd, hp, xh, node_mask, edge_mask, ex, None, None, nl, ctx)
        rx, re, inter = O.forward_dense(sd, hp, xh, node_mask, edge_mask, ex, cx, cex, nl, ctx, return_intermediates=True)
    order = sorted(range(B), key=lambda b: -n_nodes[b])        # stable, descending n
    tod = lambda x: None if x is None else x.to(dev)
    Lh = capi.lib()
    for nb in (1, 2, L):
        with torch.no_grad():
            model._plans.clear()
            # run with a block limit: need the plan first, so do a dry call then set the limit
            out = model(tod(nl), tod(xh), tod(node_mask), tod(edge_mask), edge_x=tod(ex), cond_x=tod(cx), cond_edge_x=tod(cex),
                        noise_level=tod(nl), context=tod(ctx))
            Lh.jodo_debug_set_max_blocks(model._last_plan['handle'], nb)
            out = model(tod(nl), tod(xh), tod(node_mask), tod(edge_mask), edge_x=tod(ex), cond_x=tod(cx), cond_edge_x=tod(cex),
                        noise_level=tod(nl), context=tod(ctx))
            torch.cuda.synchronize()
        Nn = sum(n_nodes)
        rows = sum(n * n for n in n_nodes)
        h = fetch(model, 0, Nn * 256).reshape(Nn, 256)
        e = fetch(model, 1, rows * 64).reshape(rows, 64)
        pos = fetch(model, 2, Nn * 4).reshape(Nn, 4)[:, :3]
        href = torch.cat([inter[b][nb - 1]['h'] for b in order])
        eref = torch.cat([inter[b][nb - 1]['e'].reshape(-1, 64) for b in order])
        pref = torch.cat([inter[b][nb - 1]['pos'] for b in order])
        # positions: ours are not centred per block -> centre per molecule before comparing
        o = 0
        pc = []
        for b in order:
            n = n_nodes[b]
            pc.append(pos[o:o + n] - pos[o:o + n].mean(0, keepdim=True))
            o += n
        pc = torch.cat(pc)
        # off-diagonal rows only for e
        mask = torch.cat([(~torch.eye(n_nodes[b], dtype=torch.bool)).reshape(-1) for b in order])
        print("blocks=%d  h %.3e  e %.3e  pos %.3e   (ref mag h %.2f e %.2f pos %.2f)" % (
            nb, (h - href).abs().max(), (e - eref)[mask].abs().max(), (pc - pref).abs().max(),
            href.abs().max(), eref.abs().max(), pref.abs().max()))
        if nb == 1:
            S = fetch(model, 4, rows * 16).reshape(rows, 16)
            # ours: slot half*8+b = head 2b+half
            perm = [2 * b + hf for hf in (0, 1) for b in range(8)]
            Sref = torch.cat([inter[b][0]['S'].reshape(-1, 16) for b in order])[:, perm]
            print("   S %.3e (mag %.2f)" % ((S - Sref)[mask].abs().max(), Sref[mask][:, 2:].abs().max()))
    Lh.jodo_debug_set_max_blocks(model._last_plan['handle'], -1)
    with torch.no_grad():
        out = model(tod(nl), tod(xh), tod(node_mask), tod(edge_mask), edge_x=tod(ex), cond_x=tod(cx), cond_edge_x=tod(cex),
                    noise_level=tod(nl), context=tod(ctx))
        torch.cuda.synchronize()
    print("final: xh %.3e  edge %.3e  flags %s" % ((out[0].cpu() - rx).abs().max(), (out[1].cpu() - re).abs().max(),
                                                   model.last_flags.cpu().tolist()))


if __name__ == '__main__':
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default='vpsde_qm9_uncond_jodo')
    ap.add_argument('--layers', type=int, default=8)
    ap.add_argument('--gain', type=float, default=1.0)
    ap.add_argument('--second', action='store_true')
    ap.add_argument('--max-chunk', type=int, default=0)
    ap.add_argument('--n', type=str, default='3,9,17,29,12,5')
    a = ap.parse_args()
    main(a.cfg, a.layers, tuple(int(x) for x in a.n.split(',')), a.gain, a.second, a.max_chunk)
