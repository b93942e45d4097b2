"""Ablation of the strip-form product (tools/experiments/strip_gemm.hip, NOT part of the library): which of weight stream, matrix
instructions and output stores bounds [M x N x K] = [41096 x 256 x 256]?   python tools/experiments/strip_ablate.py"""
import ctypes, os, sys, time
import torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, 'libxstrip.so'))
dev = 'cuda:0'
M, N, K = 41096, 256, 256
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev); ws = torch.empty(8 << 20, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(nob, no_w, no_store, no_mfma, cap, copies=1, iters=30):
    call = lambda: L.x_strip(1, M, N, K, p(A), K, p(B), K, p(C), N, None, p(ws), nob, no_w, no_store, no_mfma, cap, copies, st)
    for _ in range(3): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6
ref = (A.double() @ B.double().t())
L.x_strip(1, M, N, K, p(A), K, p(B), K, p(C), N, None, p(ws), 2, 0, 0, 0, 512, 3, st); torch.cuda.synchronize()
print('max err', float((C.double() - ref).abs().max()))
for copies in (1, 2, 4, 8, 16, 32):
    print('copies %2d: full %7.1f us   no_mfma %7.1f us' % (copies, run(4, 0, 0, 0, 512, copies), run(4, 0, 0, 1, 512, copies)))
