// fp32 GEMM of the training path on v_mfma_f32_32x32x2_f32 (exact fp32, gfx950):
//     C[M, N] (+)= op(A)[M, K] op(B)[K, N] (+ bias[N])
// The three products a linear layer needs are one kernel with two layout switches:
//     forward      Y  = X W^T + b        A = X  [M, K] row-major,            B = W stored [N, K]   (TB)
//     input grad   dX = dY W             A = dY [M, N'] row-major,           B = W stored [N', K'] as [K, N]
//     weight grad  dW = dY^T X           A = dY stored [rows, N'] = [K, M] (TA), B = X [rows, K'] = [K, N]; K = rows is the long
//                                        dimension: split over grid.z, partial tiles summed by k_splitk_sum in a fixed order
// Tile: 64 x 64 outputs per 256-thread workgroup (4 waves, one 32 x 32 accumulator block each), K in steps of 16 through LDS
// ([k][m] / [k][n], consecutive lanes read consecutive m / n: conflict-free operand reads; global loads run along whichever
// index is contiguous in memory).  Bounds are checked on every edge (N = 3, K = 17 and ragged row counts all occur).
// First-correct kernel of the training row: no double buffering, no XCD-aware tile order yet.
#include <hip/hip_runtime.h>
#include "train_gemm.h"

namespace jt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define TM 64
#define TN 64
#define TK 16
#define LDSW (TM + 4)

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm(int M, int N, int K, int kchunk, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                              float* __restrict__ C, int ldc, const float* __restrict__ bias, int acc, float* __restrict__ part) {
    __shared__ float As[TK][LDSW];
    __shared__ float Bs[TK][LDSW];
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
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e;
            {   // A tile: As[kk][i]
                int i, kk;
                if (TA) { i = idx % TM; kk = idx / TM; } else { kk = idx % TK; i = idx / TK; }
                const int gm = m0 + i, gk = k0 + kk;
                float v = 0.f;
                if (gm < M && gk < kend) v = TA ? A[(long)gk * lda + gm] : A[(long)gm * lda + gk];
                As[kk][i] = v;
            }
            {   // B tile: Bs[kk][j]
                int j, kk;
                if (TB) { kk = idx % TK; j = idx / TK; } else { j = idx % TN; kk = idx / TN; }
                const int gn = n0 + j, gk = k0 + kk;
                float v = 0.f;
                if (gn < N && gk < kend) v = TB ? B[(long)gn * ldb + gk] : B[(long)gk * ldb + gn];
                Bs[kk][j] = v;
            }
        }
        __syncthreads();
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
        // the weight-gradient shape (few output tiles, K = every row of the batch): partial sums over 256 rows each — fills the chip
        // and keeps every fp32 rounding chain short; the partial tiles are added in a fixed order by k_splitk_sum
        nsplit = (K + 255) / 256;
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
    if (tA && tB) hipLaunchKernelGGL((k_gemm<true, true>), grid, block, 0, s, M, N, K, kchunk, A, lda, B, ldb, C, ldc, bias, acc, part);
    else if (tA) hipLaunchKernelGGL((k_gemm<true, false>), grid, block, 0, s, M, N, K, kchunk, A, lda, B, ldb, C, ldc, bias, acc, part);
    else if (tB) hipLaunchKernelGGL((k_gemm<false, true>), grid, block, 0, s, M, N, K, kchunk, A, lda, B, ldb, C, ldc, bias, acc, part);
    else hipLaunchKernelGGL((k_gemm<false, false>), grid, block, 0, s, M, N, K, kchunk, A, lda, B, ldb, C, ldc, bias, acc, part);
    if (nsplit > 1)
        hipLaunchKernelGGL(k_splitk_sum, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0, s, M, N, nsplit, part, C, ldc, bias, acc);
}

}  // namespace jt
