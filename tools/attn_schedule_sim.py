"""Host-side simulation of the pair-mode attention launch: groups of whole molecules -> items of pair offsets ->
longest-first greedy onto 256 one-workgroup CUs; fits the two-parameter item cost (a + b x offsets) to the measured
offsets-per-item sweep at QM9 B = 2500 and ranks decompositions.  No GPU needed:  python tools/attn_schedule_sim.py"""
import sys, heapq
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from jodo_amd.models import load_dataset_info, get_node_dist
torch.manual_seed(42)
n_nodes = get_node_dist(load_dataset_info('qm9_with_h')).sample(2500).tolist()
G = 128
def groups(n_nodes):
    mol = sorted(n_nodes, reverse=True)
    used = [False]*len(mol); out = []; head = 0
    while head < len(mol):
        if used[head]: head += 1; continue
        fill = 0; nmax = mol[head]
        for m in range(head, len(mol)):
            if fill >= G: break
            if used[m] or mol[m] > G - fill: continue
            fill += mol[m]; used[m] = True
        out.append(nmax)
    return out
gn = groups(n_nodes)
print('groups', len(gn), 'sum dmax', sum(n//2 for n in gn))
def items_for(achunk, rule='ceil'):
    its = []
    for nmax in gn:
        dmax = nmax // 2
        if rule == 'ceil': parts = max(1, -(-dmax // achunk))
        else: parts = max(1, int(round(dmax / achunk)))
        cp = -(-dmax // parts) if dmax else 0
        for q in range(parts):
            l = min(dmax, (q+1)*cp) - min(dmax, q*cp)
            its.append(l)
    return sorted(its, reverse=True)
def makespan(its, a, b, P=256):
    h = [0.0]*P; heapq.heapify(h)
    for l in its:
        t = heapq.heappop(h); heapq.heappush(h, t + a + b*l)
    return max(h)
meas = {4: 515, 5: 540, 6: 489, 7: 508, 8: 510}
best = None
for a in np.arange(2, 30, 1.0):
    for b in np.arange(20, 45, 0.5):
        err = sum((makespan(items_for(c), a, b) - v)**2 for c, v in meas.items())
        if best is None or err < best[0]: best = (err, a, b)
print('fit', best)
_, a, b = best
for c in (3,4,5,6,7,8,9,10):
    for rule in ('ceil','round'):
        its = items_for(c, rule)
        print(c, rule, 'items', len(its), 'iters', sum(its), 'makespan %.0f' % makespan(its, a, b), 'ideal %.0f' % ((a*len(its)+b*sum(its))/256), 'meas', meas.get(c) if rule=='ceil' else '')

print('--- greedy refinement')
def items_parts(parts_list):
    its = []
    for nmax, parts in zip(gn, parts_list):
        dmax = nmax // 2
        cp = -(-dmax // parts) if dmax else 0
        for q in range(parts):
            l = min(dmax, (q+1)*cp) - min(dmax, q*cp)
            if l > 0 or q == 0: its.append(l)
    return sorted(its, reverse=True)
for c0 in (6, 9, 12, 15):
    parts = [max(1, -(-(n//2) // c0)) for n in gn]
    cur = makespan(items_parts(parts), a, b)
    improved = True
    while improved:
        improved = False
        # candidates: groups with the longest items
        order = sorted(range(len(gn)), key=lambda g: -(-(-(gn[g]//2) // parts[g])))
        for g in order[:40]:
            if parts[g] >= max(1, gn[g]//2): continue
            parts[g] += 1
            m = makespan(items_parts(parts), a, b)
            if m < cur - 0.5:
                cur = m; improved = True; break
            parts[g] -= 1
    its = items_parts(parts)
    print('start chunk', c0, '-> makespan %.0f' % cur, 'items', len(its), 'ideal %.0f' % ((a*len(its)+b*sum(its))/256))
print('--- lower per-item overhead')
for aa in (24.0, 18.0, 15.0, 12.0, 8.0):
    res = []
    for c in range(2, 13):
        its = items_for(c)
        res.append((makespan(its, aa, b), c, len(its)))
    print('a=%.0f' % aa, 'best', min(res), ' chunk6 %.0f chunk9 %.0f' % (res[4][0], res[7][0]))
