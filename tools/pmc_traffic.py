"""Turn the rocprofv3 PMC passes of `python bench.py` into the per-launch HBM-traffic summary bench.py reports
(`roofline.traffic`, `roofline.hbm`).

Inputs (all from ONE command line, run once per counter set as MI355X_MICROARCH.md prescribes — FETCH_SIZE and
WRITE_SIZE do not fit one pass, and PMC runs carry --kernel-trace only):
    --fetch  <counter_collection.csv of the FETCH_SIZE pass>
    --write  <counter_collection.csv of the WRITE_SIZE pass>
    --stats  <kernel_stats.csv of the plain --kernel-trace --stats run>      (durations, not perturbed by counters)
    --bench  <the JSON line bench.py printed in one of those runs>           (workload, batch, edges, blocks)
Corrections: FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B; on gfx950 FETCH_SIZE counts 64 B per
128-B request of a wide coalesced read, so it is DOUBLED before being compared with byte counts (the guide's gfx950
note); WRITE_SIZE is used as reported (uncalibrated, the guide says so too).

"Per block" = per (forward, DGT block): the sum over the dispatches a kernel class issues for one block (the pair
update, for instance, is two dispatches: full rounds + direction-split remainder).

    python tools/pmc_traffic.py --fetch F.csv --write W.csv --stats S.csv --bench B.json --out profiles/r02_pmc_traffic.json
"""
import argparse
import csv
import json
import re
from collections import defaultdict

CLASSES = [                      # (class name, regex on the kernel name)
    ('edge_update', r'k_edge_update'),
    ('edge_attn', r'k_edge_attn'),
    ('edge_scores', r'k_edge_scores'),
    ('edge_msgs', r'k_edge_msgs'),
    ('softmax', r'k_softmax'),
    ('node_post', r'k_node_post|k_node_ab|k_node_gram|k_node_mix|k_node_ab_pre'),
    ('node_pre', r'k_node_pre|k_pre_embed'),
]


def classify(name):
    for cls, rx in CLASSES:
        if re.search(rx, name):
            return cls
    return None


def per_kernel_counter(path, counter):
    tot, disp = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name']
        tot[k] += float(r['Counter_Value'])
        disp[k].add(r['Dispatch_Id'])
    return tot, {k: len(v) for k, v in disp.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--fetch', required=True)
    ap.add_argument('--write', required=True)
    ap.add_argument('--stats')
    ap.add_argument('--bench', required=True)
    ap.add_argument('--out', required=True)
    ap.add_argument('--command', default='python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-round')
    a = ap.parse_args()
    bench = json.loads([l for l in open(a.bench) if l.startswith('{')][-1])
    cfg = bench['config']
    E, Nn = cfg['directed_edges_per_step'], cfg['nodes_per_step']
    ftot, fdisp = per_kernel_counter(a.fetch, 'FETCH_SIZE')
    wtot, wdisp = per_kernel_counter(a.write, 'WRITE_SIZE')
    n_fwd_f = max(v for k, v in fdisp.items() if 'k_flags_init' in k)
    n_fwd_w = max(v for k, v in wdisp.items() if 'k_flags_init' in k)
    L = int(re.search(r'L=(\d+)', cfg['workload']).group(1))
    dur = {}
    if a.stats:
        for r in csv.DictReader(open(a.stats)):
            dur[r['Name']] = (float(r['TotalDurationNs']), int(r['Calls']))
        n_fwd_s = max(c for k, (t, c) in dur.items() if 'k_flags_init' in k)
    De = int(re.search(r'nf=(\d+)', cfg['workload']).group(1)) // 4
    cep = (De // 4 + 15) // 16 * 16
    # algorithmic HBM bytes per block of each class (edge state once in / once out; per-node arrays are small)
    alg = {
        'edge_update': E * (2 * De * 4 + cep * 4 + 16 + 4),            # e in, e + readout + position term out, flags
        'edge_attn': E * (De * 4 + 4) + Nn * (De * 16) * 4,             # e in (+ flags), message sums out
        'edge_scores': E * (De * 4 + 4 + De * 4 + 64),                  # e in, et + 16 scores out
        'edge_msgs': E * (De * 4 + 64) + Nn * (De * 16) * 4,            # et + scores in, message partial sums out
        # node class (k_node_post + k_node_ab + Gram tiles): SURVEY.md 8d's per-node term of the fused-per-block floor, h in + h out +
        # positions = n (2 D 4 + 24) bytes — 93.4 MB at QM9 B = 2500.  What the class actually moves on top of it (q / k / v, the two
        # input_lin halves and their coord_mlp.0 images, node2edge, the readout: the per-node arrays that feed the edge kernels) is
        # implementation traffic of the strip model, 4 x this figure in writes alone (round-5 review)
        'node_post': Nn * (2 * 4 * De * 4 + 24),
    }
    kernels = {}
    for cls, _ in CLASSES:
        f = sum(v for k, v in ftot.items() if classify(k) == cls)
        w = sum(v for k, v in wtot.items() if classify(k) == cls)
        nd = sum(v for k, v in fdisp.items() if classify(k) == cls)
        if nd == 0:
            continue
        row = {'fetch_bytes_x2_per_block': 2.0 * f * 1024.0 / (n_fwd_f * L),
               'write_bytes_per_block': w * 1024.0 / (n_fwd_w * L),
               'dispatches_per_block': nd / (n_fwd_f * L),
               'kernel_names': sorted(k[:80] for k in fdisp if classify(k) == cls)}
        if cls in alg:
            row['algorithmic_bytes_per_block'] = alg[cls]
        if dur:
            t = sum(v[0] for k, v in dur.items() if classify(k) == cls)
            row['avg_ms_per_block'] = t / (n_fwd_s * L) * 1e-6
        kernels[cls] = row
    out = {'workload': 'qm9' if 'QM9 uncond' in cfg['workload'] else ('geom384' if 'nf=384' in cfg['workload'] else
                                                                         ('cond' if 'QM9 cond' in cfg['workload'] else 'geom')),
           'batch': cfg['batch_per_gpu'], 'command': 'rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- ' + a.command,
           'corrections': 'FETCH_SIZE x 1024 B x 2 (gfx950: 64 B tallied per 128-B request); WRITE_SIZE x 1024 B as reported',
           'forwards_profiled': n_fwd_f, 'blocks_per_forward': L, 'directed_edges': E, 'nodes': Nn, 'kernels': kernels}
    json.dump(out, open(a.out, 'w'), indent=1)
    for k, v in kernels.items():
        print(k, {x: (round(y / 1e6, 1) if 'bytes' in x else y) for x, y in v.items() if x != 'kernel_names'})


if __name__ == '__main__':
    main()
