"""Gate experiments for the opt-in split-bf16 (three-term) MFMA form (round-5 review, next-round item 1, step A).
  (i)   v_mfma_f32_32x32x16_bf16 chains with v_fma fillers: does vector work hide under the bf16 matrix pipe?  (the fp32 MFMA's
        answer, jodo_debug_mfma_valu, is printed beside it)
  (ii)  one K = 256 -> 256 projection in the strip model: error against float64 of the exact-fp32 MFMA chain and of the split form
  (iii) the K = 128 -> 256 shape of the pair update's largest projection with streamed weights, timed: fp32 form vs split form with
        one and two item tiles per wave
Writes gpurun_out/split_gate.txt (copied to profiles/ by hand)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jodo_amd import capi

L = capi.lib()
out_lines = []


def say(s=''):
    print(s, flush=True)
    out_lines.append(s)


def pack(W):
    n_out, n_in = W.shape
    f = np.zeros(n_out * n_in, dtype=np.float32)
    s = np.zeros(n_out * n_in * 3, dtype=np.uint16)
    Wc = np.ascontiguousarray(W, dtype=np.float32)
    capi.check(L.jodo_debug_pack_split(Wc.ctypes.data_as(ctypes.c_void_p), n_out, n_in, f.ctypes.data_as(ctypes.c_void_p),
                                       s.ctypes.data_as(ctypes.c_void_p)), 'pack_split')
    return torch.from_numpy(f).cuda(), torch.from_numpy(s.view(np.int16)).cuda()


def chain(mode, K, tiles, x, wf, ws, iters=1, timed=False):
    rows = x.shape[0]
    y = torch.zeros(rows, 256, device='cuda')
    ms = ctypes.c_float()
    capi.check(L.jodo_debug_chain(mode, K, tiles, capi.ptr(x), rows, capi.ptr(wf), capi.ptr(ws), capi.ptr(y), iters,
                                  ctypes.byref(ms) if timed else None, capi.current_stream_ptr()), 'debug_chain')
    torch.cuda.synchronize()
    return y, ms.value


def main():
    sink = torch.zeros(4, device='cuda')
    say('# split-bf16 gate, %s' % torch.cuda.get_device_name(0))
    say()
    say('## (i) vector work beside dependent MFMA chains (one wave per SIMD unless noted; matrix TFLOP/s)')
    say('fp32 v_mfma_f32_32x32x2_f32 (4 096 flop, 64 cycles / SIMD):')
    for wps in (1, 2):
        row = []
        for nv in (0, 4, 8, 16):
            o = ctypes.c_float()
            capi.check(L.jodo_debug_mfma_valu(20000, nv, 0, wps, capi.ptr(sink), ctypes.byref(o)), 'mfma_valu')
            row.append('%2d v_fma: %6.1f' % (nv, o.value))
        say('  %d wave(s)/SIMD  ' % wps + ' | '.join(row))
    say('bf16 v_mfma_f32_32x32x16_bf16 (32 768 flop, 32 cycles / SIMD):')
    for chains, nvs in ((1, (0, 2, 4, 6, 8, 16)), (2, (0, 4, 8, 16))):
        for wps in (1, 2):
            row = []
            for nv in nvs:
                o = ctypes.c_float()
                capi.check(L.jodo_debug_mfma_bf16_valu(20000, nv, chains, wps, capi.ptr(sink), ctypes.byref(o)), 'mfma_bf16_valu')
                row.append('%2d v_fma: %7.1f' % (nv, o.value))
            say('  %d chain(s), %d wave(s)/SIMD  ' % (chains, wps) + ' | '.join(row))
    say()
    say('## (ii) K = 256 -> 256 projection, error against float64 (4 096 rows)')
    g = torch.Generator().manual_seed(1)
    for name, xgen, wgen in (
            ('LayerNorm-like x ~ N(0,1), W ~ U(-1/16, 1/16) (torch Linear init)',
             lambda: torch.randn(4096, 256, generator=g), lambda: (torch.rand(256, 256, generator=g) * 2 - 1) / 16),
            ('x ~ N(0,1) * 10^U(-3,3) per row, W ~ N(0, 1/16)',
             lambda: torch.randn(4096, 256, generator=g) * 10 ** (torch.rand(4096, 1, generator=g) * 6 - 3), lambda: torch.randn(256, 256, generator=g) / 16),
            ('positive x ~ |N(0,1)| (worst case for accumulation: no cancellation), W ~ |N(0,1/16)|',
             lambda: torch.randn(4096, 256, generator=g).abs(), lambda: torch.randn(256, 256, generator=g).abs() / 16)):
        x, W = xgen(), wgen()
        y64 = x.double() @ W.double().t()
        ycpu = (x @ W.t()).double()
        wf, ws = pack(W.numpy())
        xd = x.cuda()
        say(name)
        scale = y64.abs().max().item()
        errs = {}
        for mode, label in ((0, 'exact-fp32 MFMA chain (product form)'), (1, 'split bf16x3, one accumulator'), (2, 'split bf16x3, corrections apart')):
            y, _ = chain(mode, 256, 1, xd, wf, ws)
            e = (y.cpu().double() - y64)
            rel = (e.abs() / (y64.abs() + 1e-30))
            errs[mode] = e.abs().max().item()
            say('  %-40s max |err| %.3e  rms %.3e  max |err| / max |y| %.3e  median rel %.3e' %
                (label, e.abs().max().item(), e.pow(2).mean().sqrt().item(), e.abs().max().item() / scale, rel.median().item()))
        e = ycpu - y64
        say('  %-40s max |err| %.3e  rms %.3e' % ('torch CPU fp32 matmul', e.abs().max().item(), e.pow(2).mean().sqrt().item()))
        say('  ratio split / fp32 chain (max |err|): one accumulator %.2f, corrections apart %.2f' % (errs[1] / errs[0], errs[2] / errs[0]))
    say()
    say('## (iii) K = 128 -> 256 projection, weights streamed from L2, repeated in registers (timing)')
    W = (torch.rand(256, 128, generator=g) * 2 - 1) / 11.3
    wf, ws = pack(W.numpy())
    iters = 40
    for waves in (1024, 4096):
        say('%d waves of 32-item tiles:' % waves)
        base = None
        for mode, tiles, label in ((0, 1, 'exact fp32 MFMA'), (1, 1, 'split, 1 tile / wave'), (2, 1, 'split (2 accumulators), 1 tile / wave'),
                                   (1, 2, 'split, 2 tiles / wave'), (2, 2, 'split (2 accumulators), 2 tiles / wave')):
            rows = waves * 32
            x = torch.randn(rows, 128, device='cuda')
            best = 1e9
            for _ in range(3):
                _, ms = chain(mode, 128, tiles, x, wf, ws, iters=iters, timed=True)
                best = min(best, ms)
            flops = rows * 2.0 * 128 * 256 * iters
            if base is None:
                base = best
            say('  %-40s %8.3f ms   %7.1f fp32-equivalent TFLOP/s   x%.2f' % (label, best, flops / best / 1e9, base / best))
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/split_gate.txt', 'w') as f:
        f.write('\n'.join(out_lines) + '\n')


if __name__ == '__main__':
    main()
