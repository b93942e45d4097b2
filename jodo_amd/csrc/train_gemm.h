// Launcher of the training path's fp32 GEMM (train_gemm.hip):  C[M, N] (+)= op(A) op(B) (+ bias[N]).
//   tA = 0: A[m * lda + k];  tA = 1: A[k * lda + m]        tB = 0: B[k * ldb + n];  tB = 1: B[n * ldb + k]
//   acc = 1 adds into C; ws / ws_floats: scratch for split-K partial tiles (may be NULL: no split)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
namespace jt {
void gemm(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
          const float* bias, int acc, float* ws, size_t ws_floats);
}
