// TEST INFRASTRUCTURE ONLY: plain-loop stand-in for jt::gemm (jodo_amd/csrc/train_gemm.hip) in the host emulation build
// (tests/emul/hip/hip_runtime.h explains the build).  Same signature, same semantics; accumulates in double.
#include "train_gemm.h"
thread_local emu_idx threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
namespace jt {
void gemm(hipStream_t, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
          const float* bias, int acc, float*, size_t) {
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) {
                const float a = tA ? A[(long)k * lda + m] : A[(long)m * lda + k];
                const float b = tB ? B[(long)n * ldb + k] : B[(long)k * ldb + n];
                s += (double)a * (double)b;
            }
            if (bias) s += bias[n];
            float* o = C + (long)m * ldc + n;
            *o = acc ? (float)(*o + s) : (float)s;
        }
}
}
