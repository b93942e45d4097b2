// fp32 GEMM of the training path on v_mfma_f32_32x32x2_f32 (exact fp32, gfx950):
//     C[M, N] (+)= op(A)[M, K] op(B)[K, N] (+ bias[N])
// The three products a linear layer needs are one kernel with two layout switches:
//     forward      Y  = X W^T + b        A = X  [M, K] row-major,            B = W stored [N, K]   (TB)
//     input grad   dX = dY W             A = dY [M, N'] row-major,           B = W stored [N', K'] as [K, N]
//     weight grad  dW = dY^T X           A = dY stored [rows, N'] = [K, M] (TA), B = X [rows, K'] = [K, N]; K = rows is the long
//                                        dimension: split over grid.z, partial tiles summed by k_splitk_sum in a fixed order
// Tile: 64 x 64 outputs per 256-thread workgroup (4 waves, one 32 x 32 accumulator block each), K in steps of 32 through LDS
// ([k][m] / [k][n], consecutive lanes read consecutive m / n: conflict-free operand reads); global loads are 16 bytes per thread
// along whichever index is contiguous in memory, and the next K tile is in flight (registers) while the current one is multiplied.
// Bounds are checked on every edge (N = 3, K = 17 and ragged row counts all occur; unaligned operands take an element-wise path).
// Not yet: XCD-aware tile order, larger tiles for the square node-level products.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "train_gemm.h"

namespace jt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define TM 64
#define TN 64
#define TK 32
#define LDSW (TM + 4)

// one operand tile (64 x TK) from global memory into registers: two 16-byte loads per thread along whichever index is contiguous
// in memory (KC: the k index is contiguous — X / dY rows, W rows; otherwise the m / n index is: transposed reads of dY, plain B),
// element-wise with bounds checks where a quad is not whole or not 16-byte aligned.  r[e * 4 + j] <-> (kk, i) as store_tile says.
template <bool KC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int i0, int imax, int k0, int kend, bool vec, int tid, float (&r)[8]) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        int i, kk;                                           // first element of this thread's quad
        if (KC) { i = (tid >> 3) + 32 * e; kk = (tid & 7) * 4; } else { kk = (tid >> 4) + 16 * e; i = (tid & 15) * 4; }
        const int gi = i0 + i, gk = k0 + kk;
        const bool whole = KC ? (gi < imax && gk + 3 < kend) : (gk < kend && gi + 3 < imax);
        if (vec && whole) {
            const float4 v = *reinterpret_cast<const float4*>(KC ? P + (long)gi * ld + gk : P + (long)gk * ld + gi);
            r[e * 4 + 0] = v.x; r[e * 4 + 1] = v.y; r[e * 4 + 2] = v.z; r[e * 4 + 3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ii = KC ? gi : gi + j, kq = KC ? gk + j : gk;
                r[e * 4 + j] = (ii < imax && kq < kend) ? (KC ? P[(long)ii * ld + kq] : P[(long)kq * ld + ii]) : 0.f;
            }
        }
    }
}
template <bool KC>
__device__ __forceinline__ void store_tile(float (*S)[LDSW], int tid, const float (&r)[8]) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if (KC) {
            const int i = (tid >> 3) + 32 * e, kk = (tid & 7) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) S[kk + j][i] = r[e * 4 + j];
        } else {
            const int kk = (tid >> 4) + 16 * e, i = (tid & 15) * 4;
            *reinterpret_cast<float4*>(&S[kk][i]) = make_float4(r[e * 4 + 0], r[e * 4 + 1], r[e * 4 + 2], r[e * 4 + 3]);
        }
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm(int M, int N, int K, int kchunk, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                              float* __restrict__ C, int ldc, const float* __restrict__ bias, int acc, float* __restrict__ part, int vecA, int vecB) {
    __shared__ __attribute__((aligned(16))) float As[TK][LDSW];
    __shared__ __attribute__((aligned(16))) float Bs[TK][LDSW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    // four accumulator blocks taken in turn by the k-pairs: each rounding chain is a quarter of the K range (weight gradients
    // contract over every row of the batch; a single fp32 chain of that length costs a digit against blocked CPU summation)
    f32x16 c4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) c4[j][i] = 0.f;
    float ra[8], rb[8];
    // A is k-contiguous unless transposed; B (stored [N, K] when TB) is k-contiguous when TB
    if (kbeg < kend) {
        load_tile<!TA>(A, lda, m0, M, kbeg, kend, vecA != 0, tid, ra);
        load_tile<TB>(B, ldb, n0, N, kbeg, kend, vecB != 0, tid, rb);
    }
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        store_tile<!TA>(As, tid, ra);
        store_tile<TB>(Bs, tid, rb);
        __syncthreads();
        if (k0 + TK < kend) {                                // the next tile travels while this one is multiplied
            load_tile<!TA>(A, lda, m0, M, k0 + TK, kend, vecA != 0, tid, ra);
            load_tile<TB>(B, ldb, n0, N, k0 + TK, kend, vecB != 0, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            const float a = As[kk + (lane >> 5)][wm + (lane & 31)];
            const float b = Bs[kk + (lane >> 5)][wn + (lane & 31)];
            c4[(kk >> 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c4[(kk >> 1) & 3], 0, 0, 0);
        }
        __syncthreads();
    }
    f32x16 c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = (c4[0][i] + c4[1][i]) + (c4[2][i] + c4[3][i]);
    // C/D map of the 32x32 forms: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int col = n0 + wn + (lane & 31);
    if (col >= N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= M) continue;
        if (part) part[((long)blockIdx.z * M + row) * N + col] = c[r];
        else {
            float v = c[r] + (bias ? bias[col] : 0.f);
            float* o = C + (long)row * ldc + col;
            *o = acc ? *o + v : v;
        }
    }
}

__global__ void k_splitk_sum(int M, int N, int nsplit, const float* __restrict__ part, float* __restrict__ C, int ldc, const float* __restrict__ bias, int acc) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)M * N) return;
    const int row = (int)(i / N), col = (int)(i % N);
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += part[(long)z * M * N + i];
    s += bias ? bias[col] : 0.f;
    float* o = C + (long)row * ldc + col;
    *o = acc ? *o + s : s;
}

void gemm(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
          const float* bias, int acc, float* ws, size_t ws_floats) {
    if (M <= 0 || N <= 0) return;
    const int gx = (N + TN - 1) / TN, gy = (M + TM - 1) / TM;
    int nsplit = 1;
    if (ws && K >= 512 && (tA || (K >= 2048 && (long)gx * gy < 512))) {
        // the weight-gradient shape (few output tiles, K = every row of the batch): partial sums over 512 rows each — fills the chip
        // and keeps every fp32 rounding chain short; the partial tiles are added in a fixed order by k_splitk_sum
        nsplit = (K + 511) / 512;
        const long cap = (long)(ws_floats / ((size_t)M * N));
        if (nsplit > cap) nsplit = (int)cap;
        if (nsplit > 256) nsplit = 256;
        if (nsplit < 1) nsplit = 1;
    }
    int kchunk = (K + nsplit - 1) / nsplit;
    kchunk = (kchunk + TK - 1) / TK * TK;
    if (kchunk < TK) kchunk = TK;
    nsplit = K > 0 ? (K + kchunk - 1) / kchunk : 1;
    float* part = nsplit > 1 ? ws : nullptr;
    const dim3 grid(gx, gy, nsplit), block(256);
    // 16-byte loads need an aligned base and a row stride that keeps every quad aligned (column offsets of sliced weights included
    // in the base pointer); otherwise the element-wise path
    const int vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
    const int vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0 && (ldb & 3) == 0) ? 1 : 0;
    if (tA && tB) hipLaunchKernelGGL((k_gemm<true, true>), grid, block, 0, s, M, N, K, kchunk, A, lda, B, ldb, C, ldc, bias, acc, part, vecA, vecB);
    else if (tA) hipLaunchKernelGGL((k_gemm<true, false>), grid, block, 0, s, M, N, K, kchunk, A, lda, B, ldb, C, ldc, bias, acc, part, vecA, vecB);
    else if (tB) hipLaunchKernelGGL((k_gemm<false, true>), grid, block, 0, s, M, N, K, kchunk, A, lda, B, ldb, C, ldc, bias, acc, part, vecA, vecB);
    else hipLaunchKernelGGL((k_gemm<false, false>), grid, block, 0, s, M, N, K, kchunk, A, lda, B, ldb, C, ldc, bias, acc, part, vecA, vecB);
    if (nsplit > 1)
        hipLaunchKernelGGL(k_splitk_sum, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0, s, M, N, nsplit, part, C, ldc, bias, acc);
}

}  // namespace jt
