"""Static issue-cycle budget per PHASE of the two edge kernels, to be read beside the measured phase cycles of the same build
(tools/phase_timing.py on a -DJODO_PHASE_TIMING[_ATTN] library) — round-5 review, next-round item 6.

    python tools/cycle_budget.py [update|attn] [--measured gpurun_out/phase_<kernel>.txt]

Cross-compiles csrc/dgt_edge.hip for gfx950 with the phase timers compiled in (no GPU needed), takes the kernel's instruction stream
apart at the timers' s_memtime reads, multiplies loop bodies by their trip counts and prices every instruction class with the issue
costs measured on MI355X:
    v_mfma_f32_32x32x2_f32  64 cycles (MI355X_MICROARCH.md)       plain / packed VALU  4.6 cycles beside fp32 MFMAs (jodo_debug_mfma_valu:
    a dependent MFMA chain goes from 64 to 82.6 cycles per MFMA with 4 v_fma behind each)       transcendentals (v_exp / v_rcp / v_rsq /
    v_sqrt) 16        v_accvgpr moves 4.6        VMEM / LDS / SMEM / SALU 4 (issue only: what they WAIT for is the measured side)
    s_nop N  N + 1
The static model is an issue-cycle LOWER bound of a phase; measured - static = waiting (memory, barriers, dependency stalls)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COST = dict(mfma=64.0, valu=4.6, valu_pk=4.6, trans=16.0, agpr=4.6, vmem=4.0, lds=4.0, smem=4.0, salu=4.0, branch=4.0, barrier=4.0, wait=0.0)

# per kernel: mangled-name fragment, the phase of every timer read in STATIC order (the first read, PT_INIT, opens the stream), the trip
# count of every loop in static order of their back edges, phase names
SPECS = {
    'update': dict(
        frag='k_edge_update_symILi256ELi2ELb1ELb1ELi1E',
        phases=[0, 1, 2, 3, 4, 3, 4, 3, 4, 3, 4, 4, 5, 7, 7, 6],
        names={0: 'top: gathers, GBF, LN2 + modulate', 1: 'edge FFN (256 MFMA)', 2: 'readout (32 MFMA)', 3: 'L blocks: MFMA (160)',
               4: 'rotated statistics riding on L', 5: "folded coord_mlp.0 Z': MFMA (512)", 6: 'item end: tanh, position terms',
               7: 'SiLU / coord_mlp.2 tails riding on Z'},
        unit='one pair offset (= one item)'),
    'attn': dict(
        frag='k_edge_attnILi256ELb0ELb1ELi0E',
        # the kernel holds the item body twice (attn_item<X, PAIR = true> and the directed form for molecules above a group), each with
        # its own timer start (-1): 'attn' reports the pair form (the body with the hand-over barrier), 'attn_dir' the other
        phases=[0, 1, 2, 3, 4, 5, 6, -1, 10, 11, 12, 13, 14, 15, 16],
        names={0: 'item prologue (weights to LDS, own rows)', 1: 'edge input: GBF, edge_emb, LN (128 MFMA)', 2: 'scores: lin_edge0, tanh, q.k (256 MFMA)',
               3: 'hand-over barrier + read', 4: 'softmax update', 5: 'messages: lin_edge1, tanh, v, hand-over (256 MFMA)', 6: 'item epilogue (partials out)'},
        unit='one pair offset (loop body) / one item (prologue, epilogue)'),
}


def classify(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('v_accvgpr'):
        return 'agpr'
    if op.startswith('v_pk_'):
        return 'valu_pk'
    if op.startswith(('v_exp', 'v_rcp', 'v_rsq', 'v_log', 'v_sqrt', 'v_sin', 'v_cos')):
        return 'trans'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('s_load', 's_buffer_load')):
        return 'smem'
    if op == 's_barrier':
        return 'barrier'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op == 's_nop':
        return 'nop'
    return 'salu'


def kernel_stream(asm, frag):
    lines, on = [], False
    for line in open(asm):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            on = frag in m.group(1)
            continue
        if on:
            t = line.split(';')[0].strip()
            if t.startswith('.Lfunc_end') or t.startswith('.section'):
                break
            if not t or t.startswith('//'):
                continue
            if t.startswith('.') and not t.endswith(':'):
                continue
            lines.append(t)
    return lines


def budget(which, trips):
    spec = SPECS[which]
    out = tempfile.mktemp(suffix='.s')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', '-mllvm', '-amdgpu-mfma-vgpr-form',
                    '-DJODO_PHASE_TIMING', '-DJODO_PHASE_TIMING_ATTN', os.path.join(ROOT, 'jodo_amd', 'csrc', 'dgt_edge.hip'), '-o', out], check=True)
    ins = kernel_stream(out, spec['frag'])
    if not ins:
        raise SystemExit('kernel %s not found' % spec['frag'])
    label_at = {t[:-1]: i for i, t in enumerate(ins) if t.endswith(':')}
    # loops = backward branches; multiplicity of position p = product of the trip counts of the loops that contain it
    loops = []
    for i, t in enumerate(ins):
        m = re.match(r'^s_cbranch_\w+\s+(\S+)', t)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            loops.append((label_at[m.group(1)], i))
    timers = [i for i, t in enumerate(ins) if t.startswith('s_memtime')]
    mult = [1.0] * len(ins)
    for lo, hi in loops:
        # trips: {number of timer reads inside the loop body: trip count} — the pair update's Z loop is the one loop with exactly two
        # (PT(5), PT(7)); item / offset loops count once (the tables are per pair offset)
        tc = trips.get(sum(1 for q in timers if lo <= q <= hi), 1.0)
        for p in range(lo, hi + 1):
            mult[p] *= tc
    if len(timers) != len(spec['phases']) + 1:
        raise SystemExit('%d timer reads in the ISA, the phase list of %s expects %d — update SPECS' % (len(timers), which, len(spec['phases']) + 1))
    # an instruction belongs to the phase of the NEXT timer read it runs into: the next one in static order, or — at the bottom of a
    # loop body — the first one of that body
    phase_of_timer = dict(zip(timers[1:], spec['phases']))
    counts = collections.defaultdict(collections.Counter)
    for p in range(timers[0] + 1, timers[-1]):
        t = ins[p]
        if t.endswith(':') or t.startswith('s_memtime'):
            continue
        nxt = next(q for q in timers[1:] if q > p)
        if phase_of_timer[nxt] < 0:                             # between two bodies: belongs to neither
            continue
        for lo, hi in loops:
            if lo <= p <= hi and nxt > hi:                      # runs into the back edge first
                inner = [q for q in timers[1:] if lo <= q <= hi]
                if inner:
                    nxt = inner[0]
        op = t.split()[0]
        cl = classify(op)
        ph = phase_of_timer[nxt]
        if cl == 'nop':
            n = int(t.split()[1]) + 1 if len(t.split()) > 1 else 1
            counts[ph]['nop_cycles'] += n * mult[p]
        else:
            counts[ph][cl] += mult[p]
    return spec, counts, loops


def main():
    which = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in SPECS else 'update'
    measured = {}
    if '--measured' in sys.argv:
        for line in open(sys.argv[sys.argv.index('--measured') + 1]):
            m = re.match(r'^(\d) .*?\s+(\d+) cycles', line)
            if m:
                measured[int(m.group(1))] = float(m.group(2))
    trips = {2: 8.0} if which == 'update' else {}              # the Z loop of the pair update: D / 32 = 8 output blocks; attention: per offset
    spec, counts, loops = budget(which, trips)
    print('# %s: static issue budget per phase, %s' % (spec['frag'], spec['unit']))
    print('# loops (static positions): %s, trip counts applied: %s' % (loops, trips))
    hdr = '%-52s %6s %6s %6s %6s %6s %5s %5s %6s | %9s %9s %8s' % ('phase', 'MFMA', 'VALU', 'packed', 'trans', 'AGPR', 'VMEM', 'LDS', 'nop', 'static', 'measured', 'waiting')
    print(hdr)
    tot_s = tot_m = 0.0
    if which == 'attn':                                        # which of the two bodies is the pair form: the one whose phase 3 holds the barrier
        if counts[13]['barrier'] > counts[3]['barrier']:
            for ph in range(7):
                counts[ph] = counts[10 + ph]
    for ph in sorted(spec['names']):
        c = counts[ph]
        static = sum(c[k] * COST[k] for k in COST if k in c) + c['nop_cycles']
        meas = measured.get(ph)
        tot_s += static
        tot_m += meas or 0.0
        print('%-52s %6.0f %6.0f %6.0f %6.0f %6.0f %5.0f %5.0f %6.0f | %9.0f %9s %8s' % (
            '%d %s' % (ph, spec['names'][ph]), c['mfma'], c['valu'], c['valu_pk'], c['trans'], c['agpr'], c['vmem'], c['lds'], c['nop_cycles'],
            static, '%.0f' % meas if meas is not None else '-', '%.0f' % (meas - static) if meas is not None else '-'))
    print('%-52s %54s | %9.0f %9s' % ('total', '', tot_s, '%.0f' % tot_m if measured else '-'))


if __name__ == '__main__':
    main()
