"""Headline benchmark: QM9 unconditional JODO, 1000-step ancestral sampling, batch 2500 per GPU
(BASELINE.json configs[1]) — molecules/s and ms per denoise step on MI355X.

A "step" is one pass of the hot path over one batch: one score-network evaluation (HIP kernels via
libjodo_hip.so) plus the ancestral update of the reference's sampler, inputs resident in HBM.
`value` = molecules/s of a full 1000-step sampling round = (B * n_gpus) / (1000 * step_time);
per-step cost is independent of the step index, so K timed steps measure it directly.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload qm9|geom|geom384|cond]

N > 1: one process per GPU.  `python bench.py --gpus N` launches the ranks itself (it re-executes under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`); started by a launcher
(RANK / WORLD_SIZE in the environment) it runs as that rank.  Every rank samples its own B molecules (weak
scaling), no collective on the data path; RCCL carries the timing reduction and the final gather of the
generated molecules (`sharded_round`: one complete round through get_sampling_fn(shard=(rank, world)) +
dist.gather_sampled, wall clock, with the number of ranks whose molecules arrived).

The JSON line also carries
  roofline      fp32-MFMA roofline of the dominant kernel (k_edge_update_sym), timed live with HIP events
                on the launch stream (jodo_profile_* in the C ABI).  `achieved` / `frac` price the work the
                kernel EXECUTES: the MFMA flops of one launch from the plan's work-item lists (jodo_plan_work,
                = SQ_INSTS_MFMA x 4096 of the committed PMC pass) plus the vector flops of its per-direction
                tails, over the launch time, against the 157.3 TFLOP/s fp32 matrix peak — always <= 1.  The
                SURVEY.md 8d figure (reference formulation, directed, no symmetry / linearity savings) over the
                same time is reported as `reference_formulation_ratio` (> 1 means the algorithm does less work
                than the reference's, not that the hardware exceeds its peak).  `traffic` = HBM bytes per
                launch of that kernel from the committed PMC passes of this same command
                (profiles/*_pmc_traffic.json, written by tools/pmc_traffic.py from separate FETCH_SIZE /
                WRITE_SIZE runs; FETCH_SIZE doubled per MI355X_MICROARCH.md), `hbm` = the same for the
                three edge kernels with GB/s against the 8 TB/s peak; null when no committed pass matches
                this workload
  steady_state  >= 100 timed steps (when --steps is smaller) — the per-step figure SURVEY.md §8d asks for
  split_bf16_opt_in   the same step with the OPT-IN split-bf16 pair update (JODO_OPT_SPLIT_BF16; its own dtype label); never the
                headline: `value`, `dtype` and `roofline` above are the exact-fp32 default path
  full_round    one complete 1000-step round through the public entry points (get_sampling_fn -> sampler
                -> device decode -> host tuples), wall clock: the end-to-end molecules/s, not extrapolated
  cpu_baseline  BASELINE config 1 in full (QM9, batch 64, 50 ancestral steps) with the port of the
                reference's CPU path (our host sampler + oracle.forward_faithful) on the box's host cores
                (rank 0, N = 1 only); calibration of the port against the real reference: BASELINE.md §3
"""
import argparse
import contextlib
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    'qm9': dict(cfg='vpsde_qm9_uncond_jodo', info='qm9_with_h', batch=2500,
                name='QM9 uncond JODO (DGT_concat nf=256 L=8), 1000-step ancestral, batch 2500 per GPU'),
    'geom': dict(cfg='vpsde_geom_uncond_jodo', info='geom_with_h_1', batch=512,
                 name='GEOM-Drugs uncond JODO medium (nf=256 L=10), 1000-step ancestral, batch 512 per GPU'),
    'geom384': dict(cfg='vpsde_geom_uncond_jodo', info='geom_with_h_1', batch=1250, nf=384,
                    name='GEOM-Drugs uncond JODO large (nf=384 L=10), 1000-step ancestral, batch 1250 per GPU'),
    # BASELINE configs[4]: 10 000 molecules over 8 GPUs.  get_sampling_fn(shard=...) deals the molecules to the ranks before
    # cutting rounds, so a GPU runs ONE round of 1250 (round 1 ran four rounds of 313: --batch 313 reproduces that)
    'cond': dict(cfg='vpsde_qm9_cond_jodo', info='qm9_second_half', batch=1250,
                 name='QM9 cond JODO (cond_DGT_concat nf=256 L=8), ancestral steps, batch 1250 per GPU'),
}
PEAK_FP32_MFMA = 157.3e12      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
SAMPLING_STEPS = 1000


def reference_formulation_flops(d, n_nodes, shared_time):
    """FLOPs (multiply-add = 2) of one forward in the reference's formulation — per DIRECTED edge, no pair
    symmetry, no LayerNorm hoist / fold (per-edge time MLPs excluded, as 8d does).  d: jodo_amd.models.dims.ModelDims.
    Returns (total, edge-update share per directed edge)."""
    D, De, T, L, r, XH = d.D, d.De, d.T, d.L, d.r, d.XH
    QK = d.SH * d.SC
    upd = (2 * 2 * De * r * De + 2 * (2 * De) * D + 2 * D * D + 2 * D * (1 + XH) + 2 * De * ((2 * De) // L) + 12 * De + 8 * D)
    f_edge = (2 * (2 * De) * De + 2 * De * QK + 2 * De * D + 2 * 2 * De * r * De + 2 * (2 * De) * D + 2 * D * D + 2 * D * (1 + XH)
              + 2 * De * ((2 * De) // L) + (3 * QK + 3 * D) + 8 * (De - 1) + (24 * De + 8 * D))
    f_node = 2 * D * (2 * QK + D) + 2 * 2 * D * r * D + 2 * D * ((2 * D) // L) + 2 * D * De + 2 * 2 * D * D
    f_mol = 2 * T * (6 * D + 6 * De + 2 * D + 2)
    E = float(sum(n * (n - 1) for n in n_nodes))
    Nn = float(sum(n_nodes))
    Bm = 1.0 if shared_time else float(len(n_nodes))
    catn, cate = ((2 * D) // L) * L + D, ((2 * De) // L) * L + De
    f_pro_edge = 2 * (2 * d.ch + De) * De + 8 * (De - 1)
    f_head_node = 2 * catn * D + 2 * D * (D // 2) + 2 * (D // 2) * d.nd
    f_head_edge = 2 * (2 * cate * De + 2 * De * (De // 2)) + 2 * (De // 2) * d.ch
    total = (L * (E * f_edge + Nn * f_node + Bm * f_mol) + E * (f_pro_edge + f_head_edge) + Nn * (2 * (2 * d.nd) * D + f_head_node)
             + Bm * (2 * 17 * T + 2 * T * T))
    return total, upd


def edge_update_vector_flops_per_directed_edge(d, rot=False):
    """fp32 flops the pair update issues on the VALU per DIRECTED edge besides its MFMAs: LayerNorm statistics of
    pre = S + R_a + C_c (add, subtract, fma per feature: 4 D), the assembled coord_mlp.0 output + SiLU (7 D), coord_mlp.2's
    three dot products (6 D) and the shared edge LN2 / modulate / gate work (12 De per pair = 6 De per direction).  With the
    rotated statistics (shared modulation row, JODO_OPT_ROT_STATS): the statistics touch 2 De = D / 2 features (2 D flops) and
    the assembled output loses its mean term (5 D)."""
    return (13 if rot else 17) * d.D + 6 * d.De


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(seed=42, steps=50, batch=64, config2_batch=2500, config2_steps=2):
    """BASELINE config 1 in full: vpsde_qm9_uncond_jodo, batch 64, 50 ancestral steps on the host CPU with the
    port of the reference's path (jodo_amd's host sampler driving oracle.forward_faithful, the op-for-op mirror of
    the reference's sparse formulation; the reference itself cannot travel to the GPU box).  Same procedure as
    oracle/calibrate_cpu.py, which times it against the real reference in the build container.

    Threads: the best of {16, 32, 64} (capped at os.cpu_count()) on a short sweep, reported as `cores`.  torch's CPU
    scatter / index code does not scale to a big box's core count — measured on the GPU box of round 2 (EPYC 9575F, 256
    hardware threads): 0.96 s per step with 32 threads, 162.9 s per step with 256 (oversubscribed OpenMP barriers) — so the
    baseline runs where the reference's own torch code runs best, and says so.  `config2`: the GPU's own batch (B = 2500)
    for up to 3 steps, extrapolated."""
    from jodo_amd import configs
    from jodo_amd.diffusion import NoiseScheduleVP
    from jodo_amd.models import get_model_class, deterministic_init_, load_dataset_info, get_node_dist
    from jodo_amd.models.utils import sample_combined_position_feature_noise, sample_symmetric_edge_feature_noise
    from jodo_amd.sampling import AncestralSampler, build_masks
    from jodo_amd.utils import get_self_cond_fn
    from oracle import dgt_oracle as O
    cfg = configs.get('vpsde_qm9_uncond_jodo')
    model = deterministic_init_(get_model_class('DGT_concat')(cfg), seed=seed)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    hp = O.Hyper.from_config(cfg)
    torch.manual_seed(seed)
    n_nodes = get_node_dist(load_dataset_info('qm9_with_h')).sample(batch).tolist()
    N = max(n_nodes)
    nm, em = build_masks(n_nodes, N, 'cpu')
    z = sample_combined_position_feature_noise(batch, N, 6, nm)
    ez = sample_symmetric_edge_feature_noise(batch, N, 2, em)

    class Port:
        def __call__(self, t, xh, node_mask, edge_mask, context=None, **kw):
            return O.forward_faithful(sd, hp, xh, node_mask, edge_mask, kw['edge_x'], kw.get('cond_x'), kw.get('cond_edge_x'),
                                      kw['noise_level'], context)

    ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0, continuous_beta_1=cfg.sde.continuous_beta_1)
    smp = AncestralSampler(ns, torch.linspace(ns.T, 1e-3, steps), True, True, True, get_self_cond_fn(cfg))
    ncpu = os.cpu_count() or 1
    # thread sweep (BASELINE.md 2.2 asks for the host's cores; torch's CPU scatter / index code stops scaling long before a big
    # box's core count): three denoise steps of config 1 at {16, 32, 64} threads (one untimed), the full run at the best
    sweep = {}
    with torch.no_grad():
        for th in sorted({min(c_, ncpu) for c_ in (16, 32, 64)}):
            torch.set_num_threads(th)
            stt = smp.init_state(z, ez)
            stt = smp.step(Port(), 0, stt, nm, em, None)
            t0 = time.perf_counter()
            for i in (1, 2):
                stt = smp.step(Port(), i, stt, nm, em, None)
            sweep[th] = (time.perf_counter() - t0) / 2
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    with torch.no_grad():
        t0 = time.perf_counter()
        x_mean, e_mean = smp.sampling(Port(), z, nm, em, ez, None)
        wall = time.perf_counter() - t0
    assert bool(torch.isfinite(x_mean).all())
    # config 2's own batch on the host as well — B = 2500 molecules, up to 3 denoise steps of the same port
    # (first-step + self-conditioned evaluations; 2 by default, --cpu-config2-steps 3 for SURVEY.md 8d's three); a 1000-step round is extrapolated from the per-step time
    config2 = None
    try:
        torch.manual_seed(seed)
        b2 = config2_batch
        n2 = get_node_dist(load_dataset_info('qm9_with_h')).sample(b2).tolist()
        N2 = max(n2)
        nm2, em2 = build_masks(n2, N2, 'cpu')
        z2 = sample_combined_position_feature_noise(b2, N2, 6, nm2)
        ez2 = sample_symmetric_edge_feature_noise(b2, N2, 2, em2)
        smp2 = AncestralSampler(ns, torch.linspace(ns.T, 1e-3, SAMPLING_STEPS), True, True, True, get_self_cond_fn(cfg))
        per_step = []
        with torch.no_grad():
            st2 = smp2.init_state(z2, ez2)
            for i in range(config2_steps):
                t0 = time.perf_counter()
                st2 = smp2.step(Port(), i, st2, nm2, em2, None)
                per_step.append(time.perf_counter() - t0)
                if sum(per_step) + per_step[-1] > 75.0 * config2_steps:      # keeps the default run within a few minutes on a slow host
                    break
        s2 = sum(per_step) / len(per_step)
        config2 = dict(batch=b2, steps_timed=len(per_step), s_per_step=s2, seconds_each=per_step, value=b2 / (s2 * SAMPLING_STEPS),
                       unit='molecules/s', extrapolated=True,
                       note='BASELINE config 2 on the host: %d denoise steps of the port at B = %d (%d directed edges), value = B / '
                            '(mean step time x 1000): extrapolated from %d steps, not a full round'
                            % (len(per_step), b2, sum(n_ * (n_ - 1) for n_ in n2), len(per_step)))
    except Exception as exc:                          # e.g. a host without the memory for [E, 1024] temporaries at B = 2500
        config2 = dict(error=repr(exc))
    return dict(value=batch / (wall * SAMPLING_STEPS / steps), unit='molecules/s', cores=threads, kind='port',
                sample='BASELINE config 1 in full: QM9 uncond, batch %d, %d ancestral steps, %.1f s wall (%.3f s/step); value = '
                       'molecules/s of a 1000-step round = batch / (wall x %d) (per-step cost is step-independent)'
                       % (batch, steps, wall, wall / steps, SAMPLING_STEPS // steps),
                config1_wall_s=wall, config1_molecules_per_s=batch / wall, ms_per_step=wall / steps * 1e3,
                cpu=cpu_model(), cpu_count=ncpu, thread_sweep_s_per_step={str(k): v for k, v in sweep.items()},
                config2=config2,
                threads_note='best of {16, 32, 64} threads on 2 timed steps of config 1 (thread_sweep_s_per_step); with all 256 hardware '
                             'threads of the round-2 box the same port ran 170x slower (profiles/r02_bench_qm9_baseline.json)',
                calibration='port vs the real reference on this config: see BASELINE.md §3 (build container, 8 threads)')


class _SyntheticContext:
    """Stand-in for cond_gen's DistributionProperty (needs the dataset): context ~ N(0,1) [B, cond_ch]."""

    def __init__(self, cond_ch):
        self.cond_ch = cond_ch

    def sample_batch(self, n_nodes):
        return torch.randn(len(n_nodes), self.cond_ch)


def load_pmc_traffic(workload, batch, upd_ms):
    """HBM bytes per launch from the newest committed PMC summary of this workload (profiles/*_pmc_traffic.json,
    tools/pmc_traffic.py: separate --pmc FETCH_SIZE and WRITE_SIZE passes of `python bench.py`, FETCH_SIZE x2 per
    the gfx950 note in MI355X_MICROARCH.md, per dispatch).  Returns (traffic_bytes_of_the_dominant_launch, table) or
    (None, None).  Counters cannot be collected from inside the process being profiled, hence the committed pass."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic*.json'))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get('workload') == workload and int(d.get('batch', -1)) == int(batch):
            best = (f, d)
    if best is None:
        return None, None
    f, d = best
    table = {}
    dominant = 0.0
    for name, k in d['kernels'].items():
        bytes_launch = k['fetch_bytes_x2_per_block'] + k['write_bytes_per_block']
        row = {'hbm_bytes_per_launch': bytes_launch, 'fetch_bytes_x2': k['fetch_bytes_x2_per_block'],
               'write_bytes': k['write_bytes_per_block'], 'algorithmic_bytes': k.get('algorithmic_bytes_per_block'),
               'dispatches_per_launch': k.get('dispatches_per_block')}
        ms_ = upd_ms if (name == 'edge_update' and upd_ms > 0) else k.get('avg_ms_per_block')
        if ms_:
            row['GBps'] = bytes_launch / (ms_ * 1e-3) / 1e9
            row['frac_of_8TBps'] = row['GBps'] / 8000.0
            row['ms_per_launch'] = ms_
        table[name] = row
        if name == 'edge_update':
            dominant = bytes_launch
    table['source'] = os.path.relpath(f, ROOT)
    table['command'] = d.get('command')
    return (dominant or None), table


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='qm9', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=0)
    ap.add_argument('--seed', type=int, default=-1, help='seed of the atom-count draw and the noise (default: config.seed = 42); e.g. --workload geom '
                                                          '--seed 80 gives a batch with two molecules above an attention group (n = 144, 156)')
    ap.add_argument('--max-chunk', type=int, default=0)
    ap.add_argument('--pair-chunk', type=int, default=0)
    ap.add_argument('--spair-chunk', type=int, default=0)
    ap.add_argument('--layout', default='auto', choices=['auto', 'wide'], help="'wide': width-generic kernels at nf=256")
    ap.add_argument('--graph', action='store_true', help='replay one captured HIP graph per step (jodo_amd/graphed.py)')
    ap.add_argument('--torch-noise', action='store_true',
                    help='per-step noise from three torch.randn launches (the reference RNG stream) instead of in-kernel Philox draws')
    ap.add_argument('--streams', type=int, default=1, help='evaluate the batch as that many concurrent sub-batches on separate HIP streams')
    ap.add_argument('--no-pin', action='store_true', help='keep launching every kernel variant (device flags pick; experiments)')
    ap.add_argument('--plan-opt', action='append', default=[], help='jodo_plan_option=value (experiments), e.g. 3=0')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-config2-steps', type=int, default=2,
                    help='denoise steps of the host port at B = 2500 (about a minute each; profiles/r04_bench_qm9.json was taken with 3)')
    ap.add_argument('--no-full-round', action='store_true', help='skip the end-to-end 1000-step round')
    ap.add_argument('--no-split-leg', action='store_true', help='skip the opt-in split-bf16 leg (reported under split_bf16_opt_in)')
    ap.add_argument('--full-round', action='store_true', help='run the end-to-end round for workloads other than qm9 too')
    ap.add_argument('--breakdown', action='store_true',
                    help='time every kernel class with HIP events (costs ~0.5 ms/step of event packets; default: only the '
                         'dominant pair-update class, which the roofline object needs) and print the table to stderr')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: become the launcher — one process per GPU over RCCL (same command line per rank)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr',
               '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execvpe(sys.executable, cmd, env)
    # stdout carries exactly ONE line, the JSON record: whatever a library writes to file descriptor 1 during the run (RCCL's version banner
    # on boxes that export NCCL_DEBUG=VERSION, HIP runtime notices) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    import torch.distributed as dist
    # under a launcher (torch.distributed.run sets WORLD_SIZE) the RCCL leg runs whatever the world size: a 1-rank launch takes
    # the same init / barrier / all_reduce / all_gather path as N > 1, so the N = 1 point of a scaling sweep is like the others
    use_dist = 'WORLD_SIZE' in os.environ
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if torch.cuda.device_count() < world:
            raise SystemExit("--gpus %d but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl')
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    from jodo_amd import configs, capi, fused
    from jodo_amd.diffusion import NoiseScheduleVP
    from jodo_amd.models import get_model_class, deterministic_init_, load_dataset_info, get_node_dist
    from jodo_amd.sampling import AncestralSampler, build_masks, get_sampling_fn
    from jodo_amd.models.utils import sample_combined_position_feature_noise, sample_symmetric_edge_feature_noise
    from jodo_amd.utils import get_self_cond_fn, get_data_inverse_scaler

    wl = WORKLOADS[args.workload]
    cfg = configs.get(wl['cfg'])
    cfg.device = dev
    if 'nf' in wl:
        cfg.model.nf = wl['nf']                 # README.md:168 `--config.model.nf 384`
    if args.layout != 'auto':
        cfg.model['kernel_layout'] = args.layout
    B = args.batch or wl['batch']
    model = deterministic_init_(get_model_class(cfg.model.name)(cfg), seed=cfg.seed).to(dev).eval()
    dims = model.dims                            # product-side derived sizes (jodo_amd/models/dims.py)
    model.max_chunk = args.max_chunk
    model.pair_chunk = args.pair_chunk
    model.spair_chunk = args.spair_chunk
    model.plan_options = {int(o.split('=')[0]): int(o.split('=')[1]) for o in args.plan_opt}
    model.n_streams = args.streams
    if args.no_pin:
        model.pin_paths = lambda: None
    model.force_directed = bool(int(os.environ.get("JODO_FORCE_DIRECTED", "0")))   # debug: skip the symmetric pair kernels

    # synthetic inputs: atom counts from the training histogram (seed 42 + rank), reference noise shapes
    if args.seed >= 0:
        cfg.seed = args.seed
    torch.manual_seed(cfg.seed + rank)
    nodes_dist = get_node_dist(load_dataset_info(wl['info']))
    n_nodes = nodes_dist.sample(B).tolist()
    N = max(n_nodes)
    node_mask, edge_mask = build_masks(n_nodes, N, dev)
    node_nf = cfg.data.atom_types + int(cfg.model.include_fc_charge)
    z = sample_combined_position_feature_noise(B, N, node_nf, node_mask)
    edge_z = sample_symmetric_edge_feature_noise(B, N, cfg.model.edge_ch, edge_mask)
    context = torch.randn(B, dims.cond_ch, device=dev) if dims.cond_ch else None
    ns = NoiseScheduleVP(cfg.sde.schedule, continuous_beta_0=cfg.sde.continuous_beta_0,
                         continuous_beta_1=cfg.sde.continuous_beta_1)
    time_steps = torch.linspace(ns.T, 1e-3, SAMPLING_STEPS)
    # per-step noise: drawn inside the fused update kernel (Philox keyed by (seed, rank)) unless --torch-noise
    new_noise = lambda: None if args.torch_noise else fused.DeviceNoise.for_rank(cfg.seed, rank)
    sampler = AncestralSampler(ns, time_steps, True, True, True, get_self_cond_fn(cfg), device_noise=new_noise())

    L = capi.lib()
    if args.graph:
        # eager step 0 + warm-up step, capture, then untimed / timed replays (per-class HIP-event profiling is not
        # available inside a captured graph: the roofline leg needs the default eager mode)
        from jodo_amd.graphed import GraphedAncestralRound
        with torch.no_grad():
            rnd = GraphedAncestralRound(sampler, model, node_mask, edge_mask, context)
            rnd.prepare(z, edge_z)
            for _ in range(max(args.warmup - 2, 0)):
                rnd.replay()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                rnd.replay()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
        st = dict(x_mean=rnd.x_mean, edge_x_mean=rnd.e_mean)
    else:
        with torch.no_grad():
            st = sampler.init_state(z, edge_z)
            for i in range(args.warmup):
                st = sampler.step(model, i, st, node_mask, edge_mask, context)
                if i == 0 and args.warmup > 1:
                    model.profile_enable(1)            # class timers on every launch class for the rest of the warm-up
            # roofline leg: the timed region brackets the dominant KERNEL — the single dispatch that took the most time in the
            # warm-up.  With pinned paths the attention class (2) and the pair-update class (6) are one dispatch per block each;
            # the node class (5) is three dispatches per block, the largest of which (k_node_post, profiles/*_rocprofv3_kernel_stats_*.csv)
            # is shorter than either edge kernel on every workload, so it can lead the CLASS table (`roofline.classes`,
            # `roofline.dominant_class`) without being the kernel a profiler ranks first.
            dom, warm_classes = 6, None
            if args.warmup > 1:
                wms, wcnt = model.profile_read()
                if sum(wcnt) > 0:
                    dom = max((2, 6), key=lambda c_: wms[c_])
                    warm_classes = (wms, wcnt, args.warmup - 1)
            model.profile_enable(1 if args.breakdown else 16 + dom)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.warmup, args.warmup + args.steps):
                st = sampler.step(model, i, st, node_mask, edge_mask, context)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
    # HIP-event timings of exactly the timed region above
    ms, cnt = model.profile_read()
    model.profile_enable(0)
    flags_now = model.last_flags.cpu().tolist()
    work = model.work_model()
    n_sub = len(model._last_plans)                  # sub-batches evaluated concurrently (--streams)
    graph_info = None
    if not args.graph and world == 1:
        # extra, reported beside the headline: the same step replayed as ONE captured HIP graph (no per-step
        # Python / launch overhead).  The headline stays the eager loop, whose kernels carry the HIP-event brackets
        # the roofline leg needs.
        from jodo_amd.graphed import GraphedAncestralRound
        try:
            with torch.no_grad():
                sampler.device_noise = new_noise()
                rnd = GraphedAncestralRound(sampler, model, node_mask, edge_mask, context)
                rnd.prepare(z, edge_z)
                for _ in range(3):
                    rnd.replay()
                torch.cuda.synchronize()
                tg = time.perf_counter()
                for _ in range(args.steps):
                    rnd.replay()
                torch.cuda.synchronize()
                tg = (time.perf_counter() - tg) / args.steps
            graph_info = {'ms_per_step': tg * 1e3, 'value': B / (SAMPLING_STEPS * tg), 'unit': 'molecules/s'}
        except Exception as exc:                      # the extra must never take the headline down with it
            graph_info = {'error': repr(exc)}
    # ---- >= 100 steady-state steps when the timed region above was shorter -----------------------
    steady = None
    if not args.graph and world == 1 and args.steps < 100:
        with torch.no_grad():
            torch.cuda.synchronize()
            ts0 = time.perf_counter()
            for i in range(args.warmup + args.steps, args.warmup + args.steps + 100):
                st = sampler.step(model, i, st, node_mask, edge_mask, context)
            torch.cuda.synchronize()
            steady = {'steps': 100, 'ms_per_step': (time.perf_counter() - ts0) * 10.0}
        steady['value'] = B / (SAMPLING_STEPS * steady['ms_per_step'] * 1e-3)
        steady['unit'] = 'molecules/s'
    # ---- OPT-IN split-bf16 pair update (JODO_OPT_SPLIT_BF16), reported under its own key with its own dtype label: the headline above
    # and every roofline figure are the exact-fp32 default.  nf 256 unconditional workloads only; 50 steps with the pair-update class
    # bracketed.  Parity of this path: tests/test_split_gate.py (same stated tolerance / K64 as the default).
    split_info = None
    if not args.graph and world == 1 and dims.D in (256, 384) and not dims.cond_ch and not args.no_pin and not args.no_split_leg:
        try:
            with torch.no_grad():
                model.unpin_paths()
                model.split_bf16 = True
                model.pin_paths()                          # (re-pins the last call's plan; hands the weight tape over)
                i0 = args.warmup + args.steps + (100 if steady is not None else 0)
                for i in range(i0, i0 + 3):
                    st = sampler.step(model, i, st, node_mask, edge_mask, context)
                model.profile_enable(16 + 6)
                torch.cuda.synchronize()
                tsp = time.perf_counter()
                for i in range(i0 + 3, i0 + 53):
                    st = sampler.step(model, i, st, node_mask, edge_mask, context)
                torch.cuda.synchronize()
                tsp = (time.perf_counter() - tsp) / 50
                sms, scnt = model.profile_read()
                model.profile_enable(0)
            split_info = {'dtype': 'bf16x3 (three-term split operands, fp32 accumulate: fp32-equivalent, not bit-identical)',
                          'scope': 'pair update (k_edge_update_sym_split) and, at nf 256, the node kernels (k_node_post_split; k_node_ab_split where it is a launch of its own: >= 1024 node strips); attention, Gram tiles, embeddings, heads exact fp32',
                          'ms_per_step': tsp * 1e3, 'value': B / (SAMPLING_STEPS * tsp), 'unit': 'molecules/s',
                          'pair_update_avg_launch_ms': (sms[6] / scnt[6]) if scnt[6] else None, 'launches': scnt[6],
                          'nan_guard': bool(model.nan_guard_fired())}
        except Exception as exc:                           # the extra must never take the headline down with it
            split_info = {'error': repr(exc)}
        finally:
            model.split_bf16 = False
            model.unpin_paths()
    if use_dist:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    step_s = elapsed / args.steps

    # not timed: finish the round's host side once so the whole path is exercised (device decode)
    with torch.no_grad():
        pos, at, fc, et = fused.decode(cfg, st['x_mean'], st['edge_x_mean'], fused.n_nodes_from_mask(node_mask))
    n_total = len(fused.mols_from_decoded(pos, at, fc, et, n_nodes))
    nan_fired = model.nan_guard_fired()

    # ---- one complete round through the public entry points, wall clock ---------------------------------------------
    prop = _SyntheticContext(dims.cond_ch) if dims.cond_ch else None
    full_round = None
    if world == 1 and not args.no_full_round and (args.workload == 'qm9' or args.full_round):
        full_round = {}
        # (under a launcher the sharded round below is the third complete round: leave the torch.randn variant out there)
        for mode, hg, dn in (('eager', False, not args.torch_noise), ('eager_torch_randn', False, False))[:1 if use_dist else 2]:
            try:
                torch.manual_seed(cfg.seed)
                fn = get_sampling_fn(cfg, ns, nodes_dist, B, B, get_data_inverse_scaler(cfg), prop_dist=prop, return_raw=True,
                                     hip_graph=hg, device_noise=dn)
                torch.cuda.synchronize()
                tr = time.perf_counter()
                with contextlib.redirect_stdout(sys.stderr):      # the sampler prints its progress like the reference does;
                    mols = fn(model)                             # stdout carries the ONE JSON line only
                torch.cuda.synchronize()
                tr = time.perf_counter() - tr
                full_round[mode] = {'round_seconds': tr, 'molecules': len(mols), 'value': len(mols) / tr, 'unit': 'molecules/s',
                                    'ms_per_step_incl_everything': tr / int(cfg.sampling.steps) * 1e3,
                                    'steps': int(cfg.sampling.steps)}
            except Exception as exc:                  # an extra leg must never take the headline down with it
                full_round[mode] = {'error': repr(exc)}
        full_round['note'] = ('get_sampling_fn -> sampler -> device decode -> host tuples, one call: atom-count draw, masks, '
                              'initial noise, plan creation, %d denoise steps, decode and device->host copies all inside the '
                              'clock (weights already packed); eager draws the per-step noise inside the update kernel '
                              '(Philox), eager_torch_randn with three torch.randn launches per step (the reference RNG stream)'
                              % int(cfg.sampling.steps))
    # ---- BASELINE configs[4]'s own metric: one complete round of the hybrid DPM-solver at 50 NFE (conditional workload only) --------------
    dpm_round = None
    if world == 1 and dims.cond_ch and not args.no_full_round:
        try:
            import copy
            cfg2 = copy.deepcopy(cfg)
            cfg2.sampling.method = 'fast'                       # mix_dpm_solver.py; the keys the reference's cond config lacks take the
            cfg2.sampling.steps = 50                            # values of its uncond config (configs/vpsde_qm9_uncond_jodo.py:102-103)
            cfg2.sampling['dpm_solver_method'] = 'singlestep_fixed'
            cfg2.sampling['dpm_solver_order'] = 2
            fn = get_sampling_fn(cfg2, ns, nodes_dist, B, B, get_data_inverse_scaler(cfg2), prop_dist=prop, return_raw=True)
            with contextlib.redirect_stdout(sys.stderr):
                torch.manual_seed(cfg.seed)
                fn(model)                                       # warm round (plans, first-touch)
                torch.cuda.synchronize()
                tr = time.perf_counter()
                mols = fn(model)
                torch.cuda.synchronize()
                tr = time.perf_counter() - tr
            dpm_round = {'nfe': 50, 'round_seconds': tr, 'molecules': len(mols), 'value': len(mols) / tr, 'unit': 'molecules/s',
                         'ms_per_evaluation_incl_everything': tr / 50 * 1e3,
                         'note': 'get_sampling_fn(method=fast: DPM_Solver_hybrid, singlestep_fixed order 2, 50 NFE) -> device decode -> host tuples, '
                                 'one call, wall clock (BASELINE configs[4]: QM9 conditional + mix_dpm_solver 50 steps, per-GPU share)'}
        except Exception as exc:
            dpm_round = {'error': repr(exc)}
    # ---- under a launcher (any N): one complete SHARDED round + the RCCL gather of the generated molecules --------------------------------
    sharded = None
    if use_dist and not args.no_full_round:
        from jodo_amd.dist import gather_sampled
        try:
            fn = get_sampling_fn(cfg, ns, nodes_dist, B, B * world, get_data_inverse_scaler(cfg), prop_dist=prop, return_raw=True,
                                 shard=(rank, world), shard_mode='perf', seed=cfg.seed)
            torch.cuda.synchronize()
            dist.barrier()
            tr = time.perf_counter()
            mine, local_err = [], None
            try:                                       # the sampling itself has no collective: a rank that fails here must still
                with contextlib.redirect_stdout(sys.stderr):       # take part in the ones below, or the others wait for it forever
                    mine = fn(model)
            except Exception as exc:
                local_err = repr(exc)
            okf = torch.tensor([0 if local_err else 1], device=dev, dtype=torch.int64)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if int(okf) == 0:
                raise RuntimeError("sharded round failed on a rank" + (": " + local_err if local_err else ""))
            with contextlib.redirect_stdout(sys.stderr):
                # RCCL all_gather of the DEVICE tensors jodo_decode left (nothing re-packed on the host)
                everyone = gather_sampled(fn.last_decoded, fn.last_indices, device=dev)
            torch.cuda.synchronize()
            dist.barrier()
            tr = time.perf_counter() - tr
            tt = torch.tensor([tr], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            # which ranks' molecules arrived: every rank reports the size of its share, the gathered list must hold them all
            share = torch.tensor([len(mine)], device=dev, dtype=torch.int64)
            shares = [torch.empty_like(share) for _ in range(world)]
            dist.all_gather(shares, share)
            sharded = {'round_seconds': tt.item(), 'molecules': len(everyone), 'value': len(everyone) / tt.item(), 'unit': 'molecules/s',
                       'ranks_seen': sum(1 for s_ in shares if int(s_) > 0), 'molecules_per_rank': [int(s_) for s_ in shares],
                       'steps': int(cfg.sampling.steps),
                       'note': 'get_sampling_fn(shard=(rank, world), shard_mode=perf) on every rank + dist.gather_sampled (RCCL '
                               'all_gather over xGMI), wall clock between two barriers, max over ranks'}
        except Exception as exc:
            sharded = {'error': repr(exc)}
    if rank == 0:
        E = sum(n * (n - 1) for n in n_nodes)
        shared_row = bool(flags_now[2]) and not dims.cond_ch
        ref_total, ref_upd_edge = reference_formulation_flops(dims, n_nodes, shared_row)
        names = ['prologue', 'node_pre', 'edge_attn', 'reserved3', 'reserved4', 'node_post', 'edge_update', 'epilogue']
        per_class = {names[c]: (ms[c] / max(cnt[c], 1), cnt[c]) for c in range(8)}
        if args.graph:
            dom, warm_classes = 6, None
        nblk = dims.L
        # the dominant launch class of THIS run (largest class time in the warm-up, bracketed with HIP events over the timed region)
        dom_name = names[dom]
        dom_ms, dom_n = per_class[dom_name]
        brackets_per_fwd = (dom_n / args.steps) if dom_n else nblk      # one bracket per block (per forward for prologue / epilogue)
        mfma_launch = work[dom] / max(brackets_per_fwd, 1)           # executed MFMA flops inside ONE bracket of that class
        achieved = mfma_launch / (dom_ms * 1e-3) if dom_ms > 0 else 0.0
        # the pair update beside it (round 1-3's roofline kernel): from the timed region when it is the dominant class, else from the
        # class timers of the warm-up steps of this run
        if dom == 6:
            upd_ms, upd_n, upd_src = dom_ms, dom_n, 'timed region'
        elif warm_classes is not None and warm_classes[1][6] > 0:
            upd_ms, upd_n, upd_src = warm_classes[0][6] / warm_classes[1][6], warm_classes[1][6], 'warm-up steps of this run (class timers)'
        else:
            upd_ms, upd_n, upd_src = per_class['edge_update'][0], per_class['edge_update'][1], 'timed region'
        upd_mfma = work[6] / (nblk * n_sub)
        rot_stats = shared_row and not flags_now[4] and int(getattr(model, 'plan_options', {}).get(6, 1)) == 1
        upd_valu = E * edge_update_vector_flops_per_directed_edge(dims, rot_stats) / n_sub
        exec_step = sum(work)                                    # executed MFMA flops of one forward, all kernels
        traffic, hbm = load_pmc_traffic(args.workload, B, upd_ms)
        if hbm is not None and dom_name in hbm and isinstance(hbm[dom_name], dict):
            traffic = hbm[dom_name].get('hbm_bytes_per_launch', traffic)
        classes = None
        if warm_classes is not None:
            wms, wcnt, wsteps = warm_classes
            classes = {names[c]: {'ms_per_step': wms[c] / wsteps, 'brackets_per_step': wcnt[c] / wsteps,
                                  'mfma_frac': (work[c] / (wms[c] / wsteps * 1e-3) / PEAK_FP32_MFMA) if wms[c] > 0 else None}
                       for c in range(8) if wcnt[c] > 0}
        dom_class = None
        if classes:
            dc = max(classes, key=lambda k_: classes[k_]['ms_per_step'])
            dom_class = {'launch_class': dc, 'ms_per_step': classes[dc]['ms_per_step'], 'mfma_frac': classes[dc]['mfma_frac'],
                         'brackets_per_step': classes[dc]['brackets_per_step'],
                         'note': 'the launch CLASS with the most time per step (warm-up class timers); a class can be several dispatches '
                                 '(node_post: k_node_post + the merged k_node_ab / Gram / next-block q k v launches)'}
        # multi-GPU balance predicted on the host (jodo_amd/scaling.py): rank r of `bench.py --gpus N` draws its own B
        # molecules from seed + r — executed-work model per rank, mean / max = the efficiency the n^2 variance of the draws allows
        scaling_pred = None
        if world == 1:
            try:
                from jodo_amd import scaling
                fr = {k_: v_['mfma_frac'] for k_, v_ in (classes or {}).items() if v_.get('mfma_frac')}
                draws = []
                for r_ in range(8):
                    torch.manual_seed(cfg.seed + r_)
                    draws.append(nodes_dist.sample(B).tolist())
                scaling_pred = {'kind': 'model',          # NOT a measurement and not a SCALE line: a host-side balance prediction
                                'weak_scaling_as_bench_runs_it': {str(n_): scaling.predict_weak(model._cfg(), draws[:n_], int(shared_row), fr)
                                                                   for n_ in (2, 4, 8)},
                                'note': 'host-side prediction, not a measurement: executed MFMA flops of every rank\'s own draw (seed + rank) from '
                                        'jodo_plan_work; predicted_efficiency = mean / max over ranks; *_time_model weights the classes by the '
                                        'fractions measured in this run.  BASELINE configs[3] / [4] dealt to 8 ranks (contiguous / lpt): '
                                        'profiles/r05_scaling_prediction.json (tools/scaling_predict.py)'}
            except Exception as exc:
                scaling_pred = {'error': repr(exc)}
        kernel_names = {'edge_update': 'k_edge_update_sym' if not flags_now[4] else 'k_edge_update',
                        'node_post': 'node class: k_node_post + the two k_node_mix launches (k_node_ab items, Gram tiles)',
                        'edge_attn': 'k_edge_attn', 'node_pre': 'k_node_pre', 'prologue': 'prologue (time / fold / embeddings)',
                        'epilogue': 'epilogue (k_node_head, k_edge_head, outputs)'}
        out = {
            'metric': 'molecules/sec (1000-step ancestral)',
            'value': B * world / (SAMPLING_STEPS * step_s),
            'unit': 'molecules/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': step_s * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': wl['name'], 'batch_per_gpu': B, 'sampling_steps': SAMPLING_STEPS,
                       'directed_edges_per_step': E, 'nodes_per_step': sum(n_nodes), 'max_n': N, 'seed': int(cfg.seed),
                       'weights': 'deterministic random init (trained checkpoints are external downloads)',
                       'step_noise': 'torch.randn x3 per step' if args.torch_noise else 'in-kernel Philox4x32-10 (jodo_sampler_step_rng)',
                       'streams_per_gpu': n_sub,
                       'parallelism': 'batch shard x%d, no data-path collective' % world},
            'roofline': {'bound': 'mfma', 'kernel': kernel_names.get(dom_name, dom_name), 'launch_class': dom_name,
                         'achieved': achieved / 1e12, 'peak': PEAK_FP32_MFMA / 1e12,
                         'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP32_MFMA, 'traffic': traffic, 'hbm': hbm,
                         'note': 'kernel = the single dispatch that took the most time in this run (class timers over the warm-up steps; the '
                                 'attention and pair-update classes are one dispatch per block under pinned paths — what rocprofv3 --stats ranks '
                                 'first); achieved = fp32 MFMA flops it EXECUTES per launch '
                                 '(plan work model jodo_plan_work = SQ_INSTS_MFMA x 4096 of the committed PMC pass) / the mean '
                                 'bracket time measured over the timed region; vector flops are NOT in the numerator (reported '
                                 'separately for the pair update).  traffic: HBM bytes per bracket from the committed PMC passes of the '
                                 'same command (`hbm.source`), not from this run.  reference_formulation_ratio = the SURVEY.md 8d count '
                                 '(reference formulation: per directed edge, no pair symmetry, coord_mlp.0 per edge) / time / peak: an '
                                 'algorithmic speed-up figure, > 1 is not a hardware fraction',
                         'selection_rule': 'the larger of the attention (2) and pair-update (6) classes over the warm-up steps: under pinned paths '
                                           'each is ONE dispatch per block, i.e. the single kernel a profiler ranks first; the node class (three '
                                           'dispatches per block) can lead the class table and is reported as dominant_class (rounds 1-4 put the '
                                           'leading CLASS here: not comparable across that change)',
                         'avg_launch_ms': dom_ms, 'launches': dom_n,
                         'executed_mfma_flops_per_launch': mfma_launch,
                         'mfma_frac': achieved / PEAK_FP32_MFMA,
                         'classes': classes, 'dominant_class': dom_class,
                         'pair_update': {'kernel': kernel_names['edge_update'], 'avg_launch_ms': upd_ms, 'launches': upd_n, 'measured_over': upd_src,
                                         'executed_mfma_flops_per_launch': upd_mfma, 'executed_vector_flops_per_launch': upd_valu,
                                         'mfma_frac': (upd_mfma / (upd_ms * 1e-3) / PEAK_FP32_MFMA) if upd_ms > 0 else 0.0,
                                         'reference_formulation_flops_per_launch': E * ref_upd_edge / n_sub,
                                         'reference_formulation_ratio': (E * ref_upd_edge / n_sub / (upd_ms * 1e-3) / PEAK_FP32_MFMA) if upd_ms > 0 else 0.0},
                         'whole_step_executed_mfma_TFLOP': exec_step / 1e12,
                         'whole_step_TFLOPs': exec_step / step_s / 1e12,
                         'whole_step_frac': exec_step / step_s / PEAK_FP32_MFMA,
                         'whole_step_reference_formulation_ratio': ref_total / step_s / PEAK_FP32_MFMA,
                         'executed_mfma_flops_per_class': {names[c]: work[c] for c in range(8) if work[c] > 0}},
            # per-class totals per step; classes other than edge_update are only timed with --breakdown
            'kernel_ms': {k: round(v[0] * (v[1] / args.steps), 4) for k, v in per_class.items() if v[1] > 0},
            'hip_graph_replay': graph_info,
            'steady_state': steady,
            'split_bf16_opt_in': split_info,
            'full_round': full_round,
            'sharded_round': sharded,
            'dpm_round': dpm_round,
            'scaling_prediction': scaling_pred,
            'molecules_decoded': n_total, 'nan_guard': bool(nan_fired),
            'device_flags': dict(zip(('nan', 'first_step', 'uniform_t', 'cond_nonzero', 'asymmetric_edges'), flags_now[:5])),
        }
        if args.breakdown:
            print(json.dumps(per_class), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg.seed, config2_steps=args.cpu_config2_steps)
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
