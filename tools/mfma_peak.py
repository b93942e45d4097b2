"""Measured fp32-MFMA ceiling of the box (jodo_debug_mfma_peak): 8 independent chains and 1 dependent chain per wave."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jodo_amd import capi
L = capi.lib()
sink = torch.zeros(4, device='cuda')
for chains in (8, 1):
    for wps in (4, 1):
        out = ctypes.c_float()
        capi.check(L.jodo_debug_mfma_peak(20000, chains, wps, capi.ptr(sink), ctypes.byref(out)), 'mfma_peak')
        print('chains', chains, 'waves/SIMD', wps, 'TFLOP/s %.1f' % out.value)
for wps in (1, 2):
    for nv, nt in ((0, 0), (4, 0), (8, 0), (16, 0), (4, 1), (4, 2)):
        out = ctypes.c_float()
        capi.check(L.jodo_debug_mfma_valu(20000, nv, nt, wps, capi.ptr(sink), ctypes.byref(out)), 'mfma_valu')
        print('%d wave(s)/SIMD, dependent chain + %2d v_fma + %d v_exp per MFMA: %.1f TFLOP/s (matrix flops only)' % (wps, nv, nt, out.value))
