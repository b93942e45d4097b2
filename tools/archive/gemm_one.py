"""One product of the training GEMM, repeated (for rocprofv3 counter passes):  python tools/gemm_one.py tA tB M N K [iters]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jodo_amd import capi
tA, tB, M, N, K = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
L = capi.lib()
dev = 'cuda:0'
ws = torch.empty(32 << 20, device=dev)
A = torch.randn((K, M) if tA else (M, K), device=dev)
B = torch.randn((N, K) if tB else (K, N), device=dev)
C = torch.empty(M, N, device=dev)
for _ in range(iters):
    capi.check(L.jodo_train_gemm(tA, tB, M, N, K, capi.ptr(A), A.shape[1], capi.ptr(B), B.shape[1], capi.ptr(C), N, None, 0, capi.ptr(ws),
                                 ctypes.c_size_t(ws.numel()), capi.current_stream_ptr()), 'gemm')
torch.cuda.synchronize()
