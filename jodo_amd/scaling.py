"""Host-side prediction of the multi-GPU balance of a sharded sampling run — no GPU needed.

The path shards by molecule with no exchange step, so the N-GPU time of a round is the time of the slowest rank, and a rank's
time follows the work of its share: edge work ~ n^2, node work ~ n, both quantised by the kernels' work items.  The library's
own executed-work model (`jodo_plan_work`: MFMA flops per launch class from the plan's work-item lists, checked against
SQ_INSTS_MFMA in profiles/*_pmc_work_*.txt) prices a share exactly as the kernels would run it, so

    predicted efficiency = mean over ranks (work) / max over ranks (work)

is the scaling loss that comes from the n^2 variance of the shares alone (what `shard_assign='lpt'` is there to remove).  It says
nothing about the gather (0.4 - 44 MB per rank once per round) or about host jitter; both are small against
minutes of sampling.  `class_fracs` (the per-class fraction of the fp32-MFMA peak measured at N = 1) turns work into seconds
when given: classes run at different fractions, and a share with more node work per edge shifts the mix.

Reference being replaced: torch.nn.DataParallel (models/utils.py:27) splits a batch into equal COUNTS of molecules per GPU and
waits for the slowest replica every forward; `contiguous` below is that split applied once per round instead of once per step."""
import ctypes

import numpy as np

from . import capi
from .dist import assign_lpt, shard_range

CLASS_NAMES = ['prologue', 'node_pre', 'edge_attn', 'reserved3', 'reserved4', 'node_post', 'edge_update', 'epilogue']


def plan_work(cfg_struct, n_nodes, uniform_t, symmetric=1, plan_options=None):
    """Executed fp32 MFMA flops per launch class (8 doubles) of ONE score-network evaluation of the molecules `n_nodes`."""
    L = capi.lib()
    n_host = np.ascontiguousarray(np.asarray(n_nodes, dtype=np.int32))
    if n_host.size == 0:
        return [0.0] * 8
    handle = ctypes.c_void_p()
    capi.check(L.jodo_plan_create(ctypes.byref(cfg_struct), int(n_host.size), int(n_host.max()), n_host.ctypes.data_as(ctypes.c_void_p), 0,
                                  ctypes.byref(handle)), 'jodo_plan_create')
    try:
        for opt, val in (plan_options or {}).items():
            capi.check(L.jodo_plan_set_option(handle, int(opt), int(val)), 'jodo_plan_set_option')
        w = (ctypes.c_double * 8)()
        capi.check(L.jodo_plan_work(handle, int(uniform_t), int(symmetric), w), 'jodo_plan_work')
        return list(w)
    finally:
        L.jodo_plan_destroy(handle)


def _seconds(work, class_fracs, peak):
    if not class_fracs:
        return None
    t = 0.0
    for c, name in enumerate(CLASS_NAMES):
        f = class_fracs.get(name)
        if work[c] > 0 and f:
            t += work[c] / (f * peak)
    return t


def predict_shares(cfg_struct, shares, batch_size, uniform_t, class_fracs=None, peak=157.3e12):
    """shares: per rank, the atom counts of its molecules in the order it samples them; each rank cuts its share into rounds of up to
    `batch_size` (get_sampling_fn, shard_mode='perf').  Returns per-rank work and the implied efficiency."""
    per_rank = []
    for mine in shares:
        tot = [0.0] * 8
        for r0 in range(0, len(mine), batch_size):
            w = plan_work(cfg_struct, mine[r0:r0 + batch_size], uniform_t)
            tot = [a + b for a, b in zip(tot, w)]
        per_rank.append(tot)
    sums = [sum(w) for w in per_rank]
    out = {'ranks': len(shares), 'molecules_per_rank': [len(s) for s in shares],
           'rounds_per_rank': [(len(s) + batch_size - 1) // batch_size for s in shares],
           'sum_n2_per_rank': [int(sum(int(n) * int(n) for n in s)) for s in shares],
           'mfma_flops_per_rank': sums,
           'max_over_mean_work': (max(sums) / (sum(sums) / len(sums))) if sum(sums) > 0 else None,
           'predicted_efficiency': ((sum(sums) / len(sums)) / max(sums)) if max(sums) > 0 else None}
    if class_fracs:
        secs = [_seconds(w, class_fracs, peak) for w in per_rank]
        out['model_seconds_per_evaluation_per_rank'] = secs
        out['predicted_efficiency_time_model'] = (sum(secs) / len(secs)) / max(secs) if max(secs) > 0 else None
    return out


def predict_dealt(cfg_struct, n_nodes_all, world, batch_size, uniform_t, class_fracs=None):
    """A global list of molecules (BASELINE configs[3] / [4]: 10 000 molecules) dealt to `world` ranks the two ways
    get_sampling_fn(shard=..., shard_mode='perf') deals them: contiguous slices and LPT by n^2."""
    n_all = [int(n) for n in n_nodes_all]
    res = {}
    cont = []
    for r in range(world):
        lo, hi = shard_range(len(n_all), r, world)
        cont.append(n_all[lo:hi])
    res['contiguous'] = predict_shares(cfg_struct, cont, batch_size, uniform_t, class_fracs)
    lpt = [[n_all[i] for i in idx] for idx in assign_lpt(n_all, world)]
    res['lpt'] = predict_shares(cfg_struct, lpt, batch_size, uniform_t, class_fracs)
    return res


def predict_weak(cfg_struct, draws, uniform_t, class_fracs=None):
    """bench.py --gpus N: rank r samples its own B molecules (`draws[r]`, drawn from seed + r) — weak scaling, the step time of the
    job is the slowest rank's."""
    return predict_shares(cfg_struct, [[int(n) for n in d] for d in draws], max(len(d) for d in draws), uniform_t, class_fracs)
