"""ISA census of the device code of a .hip file: per kernel, instruction counts by class + register / scratch use.

    python tools/isa_census.py jodo_amd/csrc/dgt_forward.hip [filter-substring] [-- extra hipcc flags]

Cross-compiles for gfx950 (no GPU needed) with `--cuda-device-only -S` and parses the assembly.  Used to A/B compiler
flags and source changes by what they do to the instruction mix (MFMA, plain / packed VALU, AGPR moves, transcendentals,
memory, waits) before spending GPU time."""
import collections
import os
import re
import subprocess
import sys
import tempfile


def census(src, extra):
    out = tempfile.mktemp(suffix='.s')
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', src, '-o', out] + extra
    subprocess.run(cmd, check=True)
    kern, cur = collections.OrderedDict(), None
    meta = {}
    for line in open(out):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1)
            kern[cur] = collections.Counter()
            continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith('.end_amdhsa_kernel') or t.startswith('.Lfunc_end'):
            cur = None
            continue
        if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
            mm = re.match(r'\.amdhsa_(next_free_vgpr|accum_offset|private_segment_fixed_size|next_free_sgpr)\s+(\d+)', t)
            if mm:
                kern[cur]['meta_' + mm.group(1)] = int(mm.group(2))
            continue
        op = t.split()[0]
        c = kern[cur]
        c['total'] += 1
        if op.startswith('v_mfma'):
            c['mfma'] += 1
        elif op.startswith('v_accvgpr'):
            c['agpr_mov'] += 1
        elif op.startswith('v_pk_'):
            c['valu_pk'] += 1
        elif op.startswith(('v_exp', 'v_rcp', 'v_rsq', 'v_log', 'v_sqrt', 'v_sin', 'v_cos')):
            c['trans'] += 1
        elif op.startswith('v_'):
            c['valu'] += 1
        elif op.startswith(('buffer_', 'global_', 'flat_')):
            c['vmem'] += 1
        elif op.startswith('scratch_'):
            c['scratch'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith('s_waitcnt'):
            c['waitcnt'] += 1
        elif op.startswith('s_barrier'):
            c['barrier'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
    os.unlink(out)
    return kern


def demangle(n):
    try:
        return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip()[:90]
    except OSError:
        return n[:90]


if __name__ == '__main__':
    args = sys.argv[1:]
    extra = []
    if '--' in args:
        i = args.index('--')
        args, extra = args[:i], args[i + 1:]
    src = args[0]
    flt = args[1] if len(args) > 1 else ''
    cols = ['total', 'mfma', 'valu', 'valu_pk', 'agpr_mov', 'trans', 'vmem', 'scratch', 'lds', 'waitcnt', 'barrier', 'salu',
            'meta_next_free_vgpr', 'meta_accum_offset', 'meta_private_segment_fixed_size']
    print('%-92s' % 'kernel' + ' '.join('%8s' % c.replace('meta_', '')[:8] for c in cols))
    for k, c in census(src, extra).items():
        name = demangle(k)
        if flt and flt not in name:
            continue
        if c['total'] < 50:
            continue
        print('%-92s' % name + ' '.join('%8d' % c[x] for x in cols))
