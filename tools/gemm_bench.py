"""Throughput of the training path's fp32 MFMA GEMM (jodo_train_gemm) on the shapes a QM9 training step at batch 128 issues."""
import ctypes, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jodo_amd import capi

R, Nn = int(os.environ.get('GEMM_R', 43000)), int(os.environ.get('GEMM_NN', 2300))
SHAPES = [('fwd  c0      ', 0, 1, R, 256, 256), ('fwd  input_e ', 0, 1, R, 256, 64), ('fwd  lin_e1  ', 0, 1, R, 256, 64), ('fwd  ff3     ', 0, 1, R, 128, 64),
          ('fwd  ff4     ', 0, 1, R, 64, 128), ('fwd  ee      ', 0, 1, R, 64, 64), ('fwd  node ff1', 0, 1, Nn, 512, 256), ('fwd  mods    ', 0, 1, 128, 1536, 1024),
          ('dX   c0      ', 0, 0, R, 256, 256), ('dX   lin_e1  ', 0, 0, R, 64, 256), ('dX   ff4     ', 0, 0, R, 128, 64),
          ('dW   c0      ', 1, 0, 256, 256, R), ('dW   lin_e1  ', 1, 0, 256, 64, R), ('dW   ff3     ', 1, 0, 128, 64, R), ('dW   node ff1', 1, 0, 512, 256, Nn)]
L = capi.lib()
dev = 'cuda:0'
ws = torch.empty(32 << 20, device=dev)
for name, tA, tB, M, N, K in SHAPES:
    A = torch.randn((K, M) if tA else (M, K), device=dev)
    B = torch.randn((N, K) if tB else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    call = lambda: capi.check(L.jodo_train_gemm(tA, tB, M, N, K, capi.ptr(A), A.shape[1], capi.ptr(B), B.shape[1], capi.ptr(C), N, None, 0, capi.ptr(ws),
                                                 ctypes.c_size_t(ws.numel()), capi.current_stream_ptr()), 'gemm')
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    byt = 4.0 * (M * K + K * N + M * N)
    print('%s M %6d N %5d K %6d  %7.1f us  %6.1f TFLOP/s  %6.2f TB/s' % (name, M, N, K, dt * 1e6, 2.0 * M * N * K / dt / 1e12, byt / dt / 1e12))
