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


# ---- sharded sampling through get_sampling_fn: RNG contract, parity mode, LPT ------------------
def _sampling_setup(method, steps):
    from helpers import OracleModel, make_config, make_model, state_dict_cpu
    from jodo_amd.diffusion import NoiseScheduleVP
    from jodo_amd.models import load_dataset_info, get_node_dist
    from jodo_amd.utils import get_data_inverse_scaler
    from oracle import dgt_oracle as O
    cfg = make_config('vpsde_qm9_uncond_jodo')
    cfg.device = 'cpu'
    cfg.sampling.steps = steps
    cfg.sampling.method = method
    cfg.sampling['dpm_solver_method'] = 'singlestep_fixed'
    cfg.sampling['dpm_solver_order'] = 2
    model = OracleModel(state_dict_cpu(make_model(cfg, 5, head_gain=30.0)), O.Hyper.from_config(cfg))
    ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
    return cfg, model, ns, get_node_dist(load_dataset_info('qm9_with_h')), get_data_inverse_scaler(cfg)


def _sample_worker(rank, world, port, method, steps, mode, assign, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from jodo_amd.dist import gather_sampled
    from jodo_amd.sampling import get_sampling_fn
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg, model, ns, nodes_dist, inv = _sampling_setup(method, steps)
    torch.manual_seed(1234 + rank)                       # whatever the process did before must not matter
    fn = get_sampling_fn(cfg, ns, nodes_dist, 5, 9, inv, shard=(rank, world), shard_mode=mode, shard_assign=assign, seed=77)
    mols = fn(model)
    full = gather_sampled(fn.last_decoded, fn.last_indices)
    assert len(mols) == len(fn.last_indices)
    q.put((rank, fn.last_indices, [tuple(t.numpy().copy() for t in m) for m in full]))      # by value (no shared-memory handles)
    dist.barrier()
    dist.destroy_process_group()


def _run_world2(method, steps, mode, assign='contiguous'):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sample_worker, args=(r, 2, port, method, steps, mode, assign, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    return [(r, idx, [tuple(torch.from_numpy(a) for a in m) for m in mols]) for r, idx, mols in res]


def _same(a, b):
    return len(a) == len(b) and all(all(torch.equal(x, y) for x, y in zip(m, w)) for m, w in zip(a, b))


def test_sharded_sampling_parity_mode_reproduces_the_unsharded_run():
    """world 2, shard_mode='parity': the gathered molecules equal the world-size-1 run bit for bit — ancestral
    (BASELINE configs 1-4 path) and hybrid DPM-solver (config 5 path); 9 samples in rounds of 5, so the second
    round is ragged and the ranks hold 3 + 2 molecules of it."""
    from jodo_amd.sampling import get_sampling_fn
    threads = torch.get_num_threads()
    torch.set_num_threads(2)                  # as the workers: CPU GEMM blocking (hence low bits) depends on the thread count
    try:
        _parity_body(get_sampling_fn)
    finally:
        torch.set_num_threads(threads)


def _parity_body(get_sampling_fn):
    for method, steps in (('ancestral', 3), ('fast', 4)):
        cfg, model, ns, nodes_dist, inv = _sampling_setup(method, steps)
        torch.manual_seed(77)
        want = get_sampling_fn(cfg, ns, nodes_dist, 5, 9, inv, return_raw=True)(model)       # the reference's procedure
        one = get_sampling_fn(cfg, ns, nodes_dist, 5, 9, inv, shard=(0, 1), shard_mode='parity', seed=77)
        got1 = one(model)
        assert one.last_indices == list(range(10)) and _same(got1, want)
        res = _run_world2(method, steps, 'parity')
        assert sorted(res[0][1] + res[1][1]) == list(range(10)) and not set(res[0][1]) & set(res[1][1])
        assert res[0][1] == [0, 1, 2, 5, 6, 7]                  # contiguous slice of every round
        for _, _, full in res:
            assert _same(full, want)
        assert len({int(m[0].shape[0]) for m in want}) > 1      # not a degenerate batch


def test_sharded_sampling_perf_mode_rng_contract_and_lpt():
    """shard_mode='perf': molecules are dealt to the ranks before rounds are cut, atom counts come from the shared
    seed (identical on all ranks whatever their prior RNG state), noise from a hash of (seed, 1 + rank) (no two (seed, rank) pairs share a
    stream), LPT balances the n^2 work."""
    from jodo_amd.dist import assign_lpt
    res = _run_world2('ancestral', 2, 'perf')
    assert res[0][1] == list(range(0, 5)) and res[1][1] == list(range(5, 10))
    assert _same(res[0][2], res[1][2])                          # both ranks hold the same gathered list
    full = res[0][2]
    cfg, model, ns, nodes_dist, inv = _sampling_setup('ancestral', 2)
    torch.manual_seed(77)
    n_all = nodes_dist.sample(10).tolist()
    assert [int(m[0].shape[0]) for m in full] == n_all          # shared-seed atom counts, global order restored
    # same-size molecules on different ranks must not be duplicates of each other (independent noise streams)
    pos = {}
    for i, m in enumerate(full):
        for j, w in pos.get(int(m[0].shape[0]), []):
            assert not torch.allclose(m[0], w, atol=1e-3)
        pos.setdefault(int(m[0].shape[0]), []).append((i, m[0]))
    lpt = _run_world2('ancestral', 2, 'perf', 'lpt')
    assert sorted(lpt[0][1] + lpt[1][1]) == list(range(10))
    assert [int(m[0].shape[0]) for m in lpt[0][2]] == n_all
    a = assign_lpt(n_all, 2)
    assert a == [lpt[0][1], lpt[1][1]]
    load = [sum(n_all[i] ** 2 for i in part) for part in a]
    assert abs(load[0] - load[1]) <= max(n_all) ** 2            # LPT bound: loads differ by at most one job


def test_shard_range_covers_everything():
    for n, w in ((10, 4), (7, 2), (2500, 8), (3, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _train_worker(rank, world, port, q):
    """One rank of a data-parallel training step on the CPU: the training kernels' host-emulation build (tests/emul) computes this
    rank's gradients, jodo_amd/dist.py allreduce_gradients averages them over the flat buffer."""
    import ctypes
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from helpers import make_config, make_model, masks, random_inputs
    from jodo_amd.dist import allreduce_gradients
    from jodo_amd.train import TrainEngine
    from oracle import dgt_oracle as O
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    emul = ctypes.CDLL(os.path.join(here, 'emul', 'libjodo_train_emul.so'))
    cfg = make_config('vpsde_qm9_uncond_jodo')
    model = make_model(cfg, 3)
    hp = O.Hyper.from_config(cfg)
    n_all = [4, 7, 3, 6, 5]                                   # ragged shards: 3 + 2 molecules
    named = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    params = [v.detach().float().contiguous() for v in model.state_dict().values()]

    def grads_of(n_nodes, lo, hi, scale):
        """gradient of scale * sum_b <d_b, out_b> over molecules lo .. hi - 1 of the full batch's random inputs"""
        xh, ex, nl, ctx, nm, em = random_inputs(hp, n_all, seed=9)
        g = torch.Generator().manual_seed(17)
        d_x, d_e = torch.randn(xh.shape, generator=g), torch.randn(ex.shape, generator=g)
        N = max(n_nodes)
        cut = lambda t, dims: t[(slice(lo, hi),) + (slice(0, N),) * dims].contiguous()
        eng = TrainEngine(model._cfg(), n_nodes, N, named, 'cpu', lib=emul, stream_ptr=lambda: ctypes.c_void_p(0))
        nm_, em_ = masks(n_nodes)
        xs, es = cut(xh, 1) * nm_, cut(ex, 2) * em_.reshape(len(n_nodes), N, N, 1)
        eng.forward(params, xs, es, None, None, nl[lo:hi].contiguous(), None, 0.0, 0)
        return eng.backward(params, nl[lo:hi].contiguous(), (cut(d_x, 1) * nm_ * scale).contiguous(),
                            (cut(d_e, 2) * em_.reshape(len(n_nodes), N, N, 1) * scale).contiguous(), 0.0, 0)

    lo, hi = shard_range(len(n_all), rank, world)
    mine = grads_of(n_all[lo:hi], lo, hi, 1.0 / (hi - lo))    # this rank's MEAN over its molecules
    ps = [torch.nn.Parameter(p.clone()) for p in params]
    for p, g in zip(ps, mine):
        p.grad = g
    flat = allreduce_gradients(ps, weight=hi - lo)
    ok = flat is not None and flat.numel() >= sum(p.numel() for p in ps)
    want = grads_of(n_all, 0, len(n_all), 1.0 / len(n_all))   # the whole batch's mean on one process
    worst = 0.0
    for p, w in zip(ps, want):
        scale = float(w.abs().max())
        worst = max(worst, float((p.grad - w).abs().max()) / max(scale, 1e-12) if scale > 0 else float(p.grad.abs().max()))
    q.put((rank, bool(ok), worst))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_training_gradients_world2():
    """Two ranks with 3 and 2 molecules: after allreduce_gradients every rank holds the gradient of the mean over all 5 — what one
    process computes on the whole batch (to float32 reorder noise: the ranks' batches have different row orders and split-K slices)."""
    import subprocess
    subprocess.run(['make', '-C', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul')], check=True, capture_output=True)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[:2] for r in res] == [(0, True), (1, True)]
    assert max(r[2] for r in res) < 2e-4, res
