"""Check the executed-work model of bench.py (jodo_plan_work: MFMA flops per launch class from the plan) against the hardware:
SQ_INSTS_MFMA x 4096 flop of a rocprofv3 --pmc pass of the same command.

    python tools/pmc_work.py --pmc <counter_collection.csv with SQ_INSTS_MFMA> --bench <bench JSON line> [--out summary.json]

Prints, per launch class, model vs counter per forward and their ratio; exits 1 if the whole-step figures differ by more than 5 %."""
import argparse
import csv
import json
import re
import sys
from collections import defaultdict

CLASSES = [('edge_update', r'k_edge_update'), ('edge_attn', r'k_edge_attn'), ('node_post', r'k_node_post|k_node_ab|k_node_gram|k_node_mix|k_node_ab_pre'),
           ('node_pre', r'k_node_pre|k_pre_embed'), ('epilogue', r'k_node_head|k_edge_head|k_heads_sym'),
           ('prologue', r'k_rowgemm|k_embed_nodes|k_embed_edges|k_time1|k_cond1|k_fold_coord')]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pmc', required=True)
    ap.add_argument('--bench', required=True)
    ap.add_argument('--out')
    a = ap.parse_args()
    bench = json.loads([l for l in open(a.bench) if l.startswith('{')][-1])
    model = bench['roofline']['executed_mfma_flops_per_class']
    tot, disp = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(a.pmc)):
        if r['Counter_Name'] != 'SQ_INSTS_MFMA':
            continue
        tot[r['Kernel_Name']] += float(r['Counter_Value'])
        disp[r['Kernel_Name']].add(r['Dispatch_Id'])
    n_fwd = max(len(v) for k, v in disp.items() if 'k_flags_init' in k)
    rows, hw_total = {}, 0.0
    for cls, rx in CLASSES:
        hw = sum(v for k, v in tot.items() if re.search(rx, k)) * 4096.0 / n_fwd
        hw_total += hw
        m = float(model.get(cls, 0.0))
        rows[cls] = {'model_flops_per_forward': m, 'counter_flops_per_forward': hw, 'ratio': (m / hw) if hw else None}
        print('%-12s model %.4e  SQ_INSTS_MFMA x 4096 %.4e  ratio %s' % (cls, m, hw, ('%.4f' % (m / hw)) if hw else '-'))
    m_total = sum(float(v) for v in model.values())
    ratio = m_total / hw_total
    print('%-12s model %.4e  SQ_INSTS_MFMA x 4096 %.4e  ratio %.4f  (forwards profiled: %d)' % ('whole step', m_total, hw_total, ratio, n_fwd))
    if a.out:
        json.dump({'workload': bench['config']['workload'], 'forwards_profiled': n_fwd, 'classes': rows,
                   'whole_step': {'model': m_total, 'counter': hw_total, 'ratio': ratio}}, open(a.out, 'w'), indent=1)
    sys.exit(0 if abs(ratio - 1.0) <= 0.05 else 1)


if __name__ == '__main__':
    main()
