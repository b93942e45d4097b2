"""Multi-GPU sampling: one process per GPU, batch sharded, one gather of the generated molecules.

Replaces the reference's `torch.nn.DataParallel` (models/utils.py:27: single process, per-forward
parameter broadcast + input scatter + output gather).  Each molecule's trajectory is independent, so
there is no collective on the data path: every rank keeps the weights resident, samples its own
slice of the batch, and the decoded results are gathered once per round over RCCL/xGMI
(`torch.distributed` backend "nccl" on MI355X; "gloo" in the CPU tests).

Wire format per rank (padded to the global maxima so a single all_gather per tensor suffices):
  n_nodes [Bmax] i32 (0 = padding molecule), pos [Bmax,Nmax,3] f32, atom_type [Bmax,Nmax] u8,
  charge [Bmax,Nmax] i8, bond [Bmax,Nmax,Nmax] u8.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous slice [lo, hi) of n_items for `rank` (same split rule as sampling.get_sampling_fn)."""
    per = (n_items + world - 1) // world
    return min(rank * per, n_items), min((rank + 1) * per, n_items)


def assign_lpt(n_nodes, world):
    """Longest-processing-time assignment of molecules to ranks by their edge work n^2: molecules in
    descending n^2 go to the currently least-loaded rank.  Returns `world` ascending index lists (deterministic: ties
    broken by index and rank)."""
    order = sorted(range(len(n_nodes)), key=lambda i: (-int(n_nodes[i]) ** 2, i))
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], len(out[k]), k))
        out[r].append(i)
        load[r] += int(n_nodes[i]) ** 2
    return [sorted(o) for o in out]


def _pad_to(t, shape):
    out = torch.zeros(shape, dtype=t.dtype, device=t.device)
    out[tuple(slice(0, s) for s in t.shape)] = t
    return out


def gather_molecules(pos, atom_type, charge, bond, n_nodes, group=None):
    """All ranks call with their local decoded batch; returns on every rank a dict of concatenated
    tensors (rank order, padding molecules removed):
      n_nodes [Btot], pos [Btot,Nmax,3], atom_type [Btot,Nmax], charge [Btot,Nmax], bond [Btot,Nmax,Nmax]."""
    world = dist.get_world_size(group)
    dev = pos.device
    dims = torch.tensor([pos.shape[0], pos.shape[1]], device=dev, dtype=torch.int64)
    dist.all_reduce(dims, op=dist.ReduceOp.MAX, group=group)
    Bm, Nm = int(dims[0]), int(dims[1])
    send = dict(n_nodes=_pad_to(n_nodes.to(torch.int32), (Bm,)),
                pos=_pad_to(pos.float(), (Bm, Nm, 3)),
                atom_type=_pad_to(atom_type.to(torch.uint8), (Bm, Nm)),
                charge=_pad_to(charge.to(torch.int8), (Bm, Nm)),
                bond=_pad_to(bond.to(torch.uint8), (Bm, Nm, Nm)))
    out = {}
    for k, t in send.items():
        buf = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(buf, t.contiguous(), group=group)
        out[k] = torch.cat(buf, dim=0)
    keep = out['n_nodes'] > 0
    return {k: v[keep] for k, v in out.items()}


def unpack_molecules(g):
    """dict from gather_molecules -> list of (pos[n,3], atom_type[n], edge_type[n,n], fc[n]) CPU tuples,
    the format evaluation code of the reference consumes (sampling.py:12-32)."""
    n = g['n_nodes'].cpu().tolist()
    pos, at, ch, bd = g['pos'].cpu(), g['atom_type'].cpu().long(), g['charge'].cpu().long(), g['bond'].cpu().float()
    return [(pos[i, :k], at[i, :k], bd[i, :k, :k], ch[i, :k]) for i, k in enumerate(n)]


def _cat_rounds(decoded):
    """Rounds of padded decoded tensors (sampling_fn.last_decoded) -> one padded batch, on the device they live on."""
    if not decoded:
        raise ValueError("gather_sampled: no decoded rounds (a rank with no molecules passes decoded=[] and device=...)")
    N = max(int(r[0].shape[1]) for r in decoded)
    cat = lambda k, shape: torch.cat([_pad_to(r[k], (r[k].shape[0],) + shape) for r in decoded], dim=0)
    return cat(0, (N, 3)), cat(1, (N,)), cat(2, (N,)), cat(3, (N, N)), torch.cat([r[4] for r in decoded], dim=0)


def gather_sampled(decoded, indices, device=None, group=None):
    """End-of-sampling collective for `get_sampling_fn(shard=...)`: every rank passes the decoded tensors of its rounds
    (`sampling_fn.last_decoded`: per round pos [B,N,3] f32, atom_type [B,N] u8, charge [B,N] i8, bond [B,N,N] u8, n_nodes [B] i32 —
    on the GPU box the outputs of jodo_decode, still on the device: nothing is re-packed on the host) and their global indices
    (`sampling_fn.last_indices`); returns, on every rank, the full list of molecule tuples (pos[n,3], atom_type[n],
    edge_type[n,n], fc[n]) in global order — the list an unsharded run would have produced.  One all_gather per tensor
    (RCCL over xGMI with device tensors and the "nccl" backend; gloo with CPU tensors)."""
    if decoded:
        pos, at, ch, bd, n = _cat_rounds(decoded)
        device = pos.device if device is None else torch.device(device)
        if pos.device != device:
            pos, at, ch, bd, n = (t.to(device) for t in (pos, at, ch, bd, n))
    else:                                                      # a rank whose share is empty still takes part in the collectives
        device = torch.device(device or 'cpu')
        pos = torch.zeros(0, 1, 3, device=device)
        at, ch = torch.zeros(0, 1, dtype=torch.uint8, device=device), torch.zeros(0, 1, dtype=torch.int8, device=device)
        bd, n = torch.zeros(0, 1, 1, dtype=torch.uint8, device=device), torch.zeros(0, dtype=torch.int32, device=device)
    B = int(pos.shape[0])
    if len(indices) != B:
        raise ValueError("gather_sampled: %d molecules but %d global indices" % (B, len(indices)))
    g = gather_molecules(pos, at, ch, bd, n, group=group)
    world = dist.get_world_size(group)
    cnt = torch.tensor([B], dtype=torch.int64, device=device)
    cnts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    Bm = max(int(c) for c in cnts)
    idx = torch.full((max(Bm, 1),), -1, dtype=torch.int64, device=device)
    idx[:B] = torch.as_tensor(list(indices), dtype=torch.int64, device=device)
    idxs = [torch.empty_like(idx) for _ in range(world)]
    dist.all_gather(idxs, idx, group=group)
    order = torch.cat([i[:int(c)] for i, c in zip(idxs, cnts)]).cpu()
    out = unpack_molecules(g)
    assert len(out) == order.numel()
    return [out[k] for k in torch.argsort(order).tolist()]


# ---- training (SURVEY.md §8f row 4 on more than one GPU) ------------------------------------------------------------------------
def allreduce_gradients(params, weight=None, group=None):
    """Data-parallel training step, one process per GPU: every rank has run the step's forward / backward on ITS molecules, and the
    gradients become those of the mean loss over all ranks' molecules with ONE all-reduce — over the flat buffer every p.grad is a slice
    of (jodo_amd/train.py hands all 351 gradients out of one allocation; 110 MB at QM9), not 351 small ones.  What the reference gets
    from `torch.nn.DataParallel` (models/utils.py:27: scatter the batch, gather the outputs, one loss on GPU 0) without its per-forward
    parameter broadcast: weights stay resident and in step because every rank applies the same averaged gradient.

    weight: this rank's number of molecules when shards differ in size (the loss is a mean over molecules, losses.py:385: the global
    gradient is sum_r B_r g_r / sum_r B_r); None: equal shards, plain mean.  Returns the flat gradient (or None: nothing to reduce)."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    world = dist.get_world_size(group)
    if world == 1:
        return None
    from .optim import flat_view, carve, slice_offsets
    params = [p for p in params if p.grad is not None]
    if not params:
        return None
    flat = flat_view([p.grad for p in params])
    gathered = flat is None
    if gathered:                                             # ordinary separate gradients: through one buffer and back
        offs, total = slice_offsets([p.numel() for p in params])
        flat = torch.zeros(total, dtype=params[0].grad.dtype, device=params[0].grad.device)
        pieces = carve(flat, [tuple(p.shape) for p in params], offs)
        torch._foreach_copy_(pieces, [p.grad for p in params])
    if weight is None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
    else:
        w = torch.tensor([float(weight)], dtype=torch.float32, device=flat.device)
        flat.mul_(float(weight))
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(w, op=dist.ReduceOp.SUM, group=group)
        flat.div_(w)
    if gathered:
        torch._foreach_copy_([p.grad for p in params], pieces)
    return flat
