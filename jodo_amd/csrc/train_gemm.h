// Launcher of the training path's fp32 GEMM (train_gemm.hip):  C[M, N] (+)= op(A) op(B) (+ bias[N]).
//   tA = 0: A[m * lda + k];  tA = 1: A[k * lda + m]        tB = 0: B[k * ldb + n];  tB = 1: B[n * ldb + k]
//   acc = 1 adds into C; ws / ws_floats: scratch for split-K partial tiles (may be NULL: no split)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
namespace jt {
void gemm(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
          const float* bias, int acc, float* ws, size_t ws_floats);

// Tiling decision, shared by the device launcher and by the host emulation of the CPU suite (tests/emul/emul_gemm.cpp mirrors the
// kernel's rounding structure from the same numbers).  A workgroup of 4 waves (2 x 2) computes a (64 rm) x (64 rn) tile, every wave
// rm x rn accumulator blocks of 32 x 32; K runs in tiles of 32.  Split-K over grid.z: weight gradients (tA: K = every row of the
// batch) in partial sums over 512 rows; products with too few output tiles to fill the chip (the per-molecule modulation
// projections: 128 rows) over 128-wide slices of K.
struct GemmPlan { int rm, rn, nsplit, kchunk; };
inline GemmPlan gemm_plan(int tA, int M, int N, int K, bool have_ws, size_t ws_floats) {
    GemmPlan p;
    p.rm = M > 64 ? 2 : 1;
    p.rn = N > 64 ? 2 : 1;
    const long tiles = (long)((M + 64 * p.rm - 1) / (64 * p.rm)) * ((N + 64 * p.rn - 1) / (64 * p.rn));
    int nsplit = 1;
    if (have_ws && K >= 512 && (tA || tiles < 128)) {
        nsplit = tA ? (K + 511) / 512 : (K + 127) / 128;
        if (!tA && nsplit > 256 / tiles) nsplit = (int)(256 / tiles);
        const long cap = (long)(ws_floats / ((size_t)M * N));
        if (nsplit > cap) nsplit = (int)cap;
        if (nsplit > 256) nsplit = 256;
        if (nsplit < 1) nsplit = 1;
    }
    int kchunk = (K + nsplit - 1) / nsplit;
    kchunk = (kchunk + 31) / 32 * 32;
    if (kchunk < 32) kchunk = 32;
    p.nsplit = K > 0 ? (K + kchunk - 1) / kchunk : 1;
    p.kchunk = kchunk;
    return p;
}
}
