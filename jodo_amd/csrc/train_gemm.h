// Launcher of the training path's fp32 GEMM (train_gemm.hip):  C[M, N] (+)= op(A) op(B) (+ bias[N]).
//   tA = 0: A[m * lda + k];  tA = 1: A[k * lda + m]        tB = 0: B[k * ldb + n];  tB = 1: B[n * ldb + k]
//   acc = 1 adds into C; ws / ws_floats: scratch for split-K partial tiles (may be NULL: no split)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "train_common.h"
namespace jt {
// Activation fused behind the product (v = sum + bias):  act 1: C = tanh(v);  act 2: C = v (the pre-activation the backward needs)
// and out2[m, n] = SiLU(v) * dropout(element m N + n) with out2 laid out like C.  Not combined with acc.
// dbias (weight-gradient products, tA = 1, act = 0): dbias[m] += sum_k A(k, m) — the bias gradient is the column sum of the very
// dY tiles the product stages in LDS, so it rides along instead of a reduction pass of its own (accumulated in double per slice).
struct GemmEpi { int act; float* out2; Drop drop; float* dbias; };
void gemm(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
          const float* bias, int acc, float* ws, size_t ws_floats, const GemmEpi* epi = nullptr);
// Weight-gradient products queued by the backward and launched together (train_gemm.hip gemm_dw_group):
//     C[M, N] (ldc) += A[K, M]^T (lda) B[K, N] (ldb),   dbias[M] += column sums of A   (dbias may be NULL)
// — what gemm(s, 1, 0, M, N, K, A, lda, B, ldb, C, ldc, nullptr, 1, ws, plan_floats, {dbias}) computes, bit for bit.
struct GemmJob { int M, N, K; const float* A; int lda; const float* B; int ldb; float* C; int ldc; float* dbias; };
void gemm_dw_group(hipStream_t s, const GemmJob* jobs, int n, float* ws, size_t ws_floats, size_t plan_floats);
// the table a grouped launch carries in its kernel arguments (device side of gemm_dw_group)
#define GEMM_GROUP_MAX 24
struct GemmGroupJob { const float *A, *B; float *C, *part, *dbias; int M, N, K, kchunk, lda, ldb, ldc, nx, ny, nz, vecA, vecB, wg0, sb0; };
struct GemmGroup { int n; GemmGroupJob j[GEMM_GROUP_MAX]; };
__host__ __device__ __forceinline__ void gemm_epilogue(const GemmEpi& e, float v, float* C, long cidx, long didx) {
    if (e.act == 1) C[cidx] = tanhf(v);
    else { C[cidx] = v; e.out2[cidx] = silu_f(v) * drop_mul(e.drop, (unsigned long long)didx); }
}

// Tiling decision, shared by the device launcher and by the host emulation of the CPU suite (tests/emul/emul_gemm.cpp mirrors the
// kernel's rounding structure from the same numbers).  A workgroup of 4 waves (2 x 2) computes a (64 rm) x (64 rn) tile, every wave
// rm x rn accumulator blocks of 32 x 32 (rm = rn = 1 is what runs, see below); K runs in tiles of 32.  Split-K over grid.z: weight gradients (tA: K = every row of the
// batch) in partial sums over 512 rows; products with too few output tiles to fill the chip (the per-molecule modulation
// projections: 128 rows) over 128-wide slices of K.
struct GemmPlan { int rm, rn, nsplit, kchunk; };
inline GemmPlan gemm_plan(int tA, int M, int N, int K, bool have_ws, size_t ws_floats) {
    GemmPlan p;
    // 64 x 64 tiles everywhere: measured on MI355X (tools/gemm_bench.py, QM9 batch-128 shapes) the 128-wide tiles lose — 271
    // registers leave one wave per SIMD, and with K <= 256 a tile has too few K steps to hide its own loads: c0 [43 000 x 256 x 256]
    // 95.7 us at 64 x 64 (five to six workgroups per CU overlap each other's loads and barriers) against 139.3 us at 128 x 128,
    // ff4 16.9 against 24.0 (128 x 64).  The kernel keeps the general form.
    p.rm = 1;
    p.rn = 1;
    (void)M; (void)N;
    const long tiles = (long)((M + 64 * p.rm - 1) / (64 * p.rm)) * ((N + 64 * p.rn - 1) / (64 * p.rn));
    int nsplit = 1;
    if (have_ws && K >= 512 && (tA || tiles < 128)) {
        if (tA) {
            // slices of 512 rows
            nsplit = (K + 511) / 512;
            // ... unless that leaves most of the chip idle (outputs of one to four tiles: [64 x 64], [128 x 64], [3 x 256] x 43 000 rows
            // ran 81 - 170 workgroups on 256 CUs, 24 - 33 us each): slices down to 64 rows until about 768 workgroups exist (k_splitk_sum
            // spreads the slices of an output over eight lanes, so hundreds of slices cost it little)
            if (tiles * nsplit < 512) {
                const long want = 768 / tiles, finest = (K + 63) / 64;
                const long fine = want < finest ? want : finest;
                if (fine > nsplit) nsplit = (int)fine;
            }
        } else {
            nsplit = (K + 127) / 128;
            if (nsplit > 256 / tiles) nsplit = (int)(256 / tiles);
        }
        const long cap = (long)(ws_floats / ((size_t)M * N + (size_t)M));      // partial tiles + partial bias sums
        if (nsplit > cap) nsplit = (int)cap;
        if (nsplit > 512) nsplit = 512;
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
