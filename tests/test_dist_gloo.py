"""CPU, world_size 2 over gloo: batch sharding and the single end-of-round gather (jodo_amd/dist.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jodo_amd.dist import gather_molecules, shard_range, unpack_molecules


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_all, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = shard_range(len(n_all), rank, world)
    n = torch.tensor(n_all[lo:hi])
    B, N = len(n), int(n.max())
    g = torch.Generator().manual_seed(100 + rank)
    pos = torch.randn(B, N, 3, generator=g)
    at = torch.randint(0, 5, (B, N), generator=g)
    ch = torch.randint(-1, 2, (B, N), generator=g)
    bd = torch.randint(0, 4, (B, N, N), generator=g)
    out = gather_molecules(pos, at, ch, bd, n)
    mols = unpack_molecules(out)
    # every rank can check its own slice landed at the right place, bit-exact
    ok = len(mols) == len(n_all)
    for k in range(B):
        m = mols[lo + k]
        nk = int(n[k])
        ok &= torch.equal(m[0], pos[k, :nk]) and torch.equal(m[1], at[k, :nk])
        ok &= torch.equal(m[2], bd[k, :nk, :nk].float()) and torch.equal(m[3], ch[k, :nk])
    ok &= [int(m[0].shape[0]) for m in mols] == list(n_all)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    n_all = [5, 9, 3, 17, 12, 29, 4]          # ragged: 4 + 3 molecules, different N per rank
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_all, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_shard_range_covers_everything():
    for n, w in ((10, 4), (7, 2), (2500, 8), (3, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
