// fp32 GEMM of the training path on v_mfma_f32_32x32x2_f32 (exact fp32, gfx950):
//     C[M, N] (+)= op(A)[M, K] op(B)[K, N] (+ bias[N])
// The three products a linear layer needs are one kernel with two layout switches:
//     forward      Y  = X W^T + b        A = X  [M, K] row-major,            B = W stored [N, K]   (TB)
//     input grad   dX = dY W             A = dY [M, N'] row-major,           B = W stored [N', K'] as [K, N]
//     weight grad  dW = dY^T X           A = dY stored [rows, N'] = [K, M] (TA), B = X [rows, K'] = [K, N]; K = rows is the long
//                                        dimension: split over grid.z, partial tiles summed by k_splitk_sum in a fixed order
// Tile (train_gemm.h gemm_plan): (64 rm) x (64 rn) outputs per 256-thread workgroup, 2 x 2 waves of rm x rn accumulator blocks each
// (a wave reads rm + rn operand registers from LDS per rm rn MFMAs: 128 x 128 tiles need one read per instruction, 64 x 64 two), K in
// steps of 32 through LDS ([k][m] / [k][n], consecutive lanes read consecutive m / n: conflict-free); global loads are 16 bytes per
// thread along whichever index is contiguous in memory, and the next K tile is in flight (registers) while the current one is
// multiplied.  Two accumulator chains per block (even / odd k-pairs) and short split-K partials keep the fp32 rounding chains short.
// Bounds are checked on every edge (N = 3, K = 17 and ragged row counts all occur; unaligned operands take an element-wise path);
// products whose tiles are all whole in K take the branch-free loader (FAST), small launches with four tiles in flight.  The tile order
// is XCD-aware (k_gemm).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include "train_gemm.h"

namespace jt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define TK 32

// one operand tile (64 R rows x TK) from global memory into registers: 2 R 16-byte loads per thread along whichever index is
// contiguous in memory (KC: the k index is — X / dY rows, W rows; otherwise the m / n index is: transposed reads of dY, plain B),
// element-wise with bounds checks where a quad is not whole or not 16-byte aligned.
template <bool KC, int R>
__device__ __forceinline__ void tile_pos(int tid, int e, int& i, int& kk) {
    if (KC) { i = (tid >> 3) + 32 * e; kk = (tid & 7) * 4; }
    else { const int q = tid + 256 * e; kk = q / (16 * R); i = (q % (16 * R)) * 4; }
}
template <bool KC, int R>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int i0, int imax, int k0, int kend, bool vec, int tid, float (&r)[8 * R]) {
#pragma unroll
    for (int e = 0; e < 2 * R; ++e) {
        int i, kk;
        tile_pos<KC, R>(tid, e, i, kk);
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
// The same without a single branch, for operands that are 16-byte addressable (FAST): rows / columns past the edge are clamped to the
// last valid ones — they only feed outputs that are never stored — and quads past the END OF K (the last tile of a product or of a
// split-K slice: weight gradients contract over the batch's rows, 41 096 of them, not a multiple of 32) are clamped to the last valid
// quad and zeroed by a select on their way into LDS (store_tile), so every load is unconditional and the compiler can count them: several tiles then really stay in flight
// (around the bounds branches it waits for every load issued).  k-contiguous operands need K % 4 == 0 for whole quads (gemm()).
template <bool KC, int R>
__device__ __forceinline__ void load_tile_fast(const float* __restrict__ P, int ld, int i0, int imax, int k0, int kend, int tid, float (&r)[8 * R]) {
#pragma unroll
    for (int e = 0; e < 2 * R; ++e) {
        int i, kk;
        tile_pos<KC, R>(tid, e, i, kk);
        int gi = i0 + i;
        int gk = k0 + kk;
        const bool in = gk < kend;
        if (KC) { gi = gi < imax ? gi : imax - 1; gk = in ? gk : kend - 4; }
        else { gi = gi + 3 < imax ? gi : imax - 4; gk = in ? gk : kend - 1; }
        const float4 v = *reinterpret_cast<const float4*>(KC ? P + (long)gi * ld + gk : P + (long)gk * ld + gi);
        r[e * 4 + 0] = v.x; r[e * 4 + 1] = v.y; r[e * 4 + 2] = v.z; r[e * 4 + 3] = v.w;     // (zeroed in store_tile: a select here would wait for the load)
    }
}
// LDS row length: k-contiguous operands are transposed on the way in (four 4-byte stores per quad; a wave covers 8 quads of k x 8 rows):
// with 64 R + 2 words per k row the 64 lanes fall on every bank exactly twice (the minimum; 64 R + 4 put them on 8 banks, eight deep:
// SQ_LDS_BANK_CONFLICT was 1.3 cycles per LDS instruction).  The other layout stores whole quads and needs 16-byte rows.
template <bool KC, int R> struct TileLd { static constexpr int value = 64 * R + (KC ? 2 : 4); };
template <bool KC, int R, int LD>
__device__ __forceinline__ void store_tile(float (*S)[LD], int tid, const float (&r)[8 * R], int k0, int kend) {
#pragma unroll
    for (int e = 0; e < 2 * R; ++e) {
        int i, kk;
        tile_pos<KC, R>(tid, e, i, kk);
        const bool in = k0 + kk < kend;                      // quads past the end of K count as zeros (whole quads: see load_tile_fast)
        if (KC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) S[kk + j][i] = in ? r[e * 4 + j] : 0.f;
        } else {
            *reinterpret_cast<float4*>(&S[kk][i]) = in ? make_float4(r[e * 4 + 0], r[e * 4 + 1], r[e * 4 + 2], r[e * 4 + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// PF (FAST only): operand tiles in flight.  1: the next tile travels while the current one is multiplied — enough where five to seven
// workgroups per CU cover each other's waits; 4: launches of fewer workgroups than the chip has SIMDs (node- and molecule-level products:
// [2 260 x 256 x 256] is 144 workgroups of eight dependent K steps, each an exposed load latency).  Same arithmetic, same order.
// One workgroup's tile (bx, by) of slice bz out of nz: the body of k_gemm and of k_gemm_group.
template <bool TA, bool TB, int RM, int RN, bool FAST, int PF>
__device__ __forceinline__ void gemm_tile(int M, int N, int K, int kchunk, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                          float* __restrict__ C, int ldc, const float* __restrict__ bias, int acc, float* __restrict__ part, int vecA, int vecB,
                                          const GemmEpi& epi, int bx, int by, int bz, int nz, float (*As)[TileLd<!TA, RM>::value], float (*Bs)[TileLd<TB, RN>::value]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = by * 64 * RM, n0 = bx * 64 * RN;
    const int kbeg = bz * kchunk, kend = min(K, kbeg + kchunk);
    const int wm = (wave >> 1) * 32 * RM, wn = (wave & 1) * 32 * RN;
    // two accumulator chains per block, taken in turn by the k-pairs (weight gradients contract over every row of the batch; a single
    // fp32 chain of that length costs a digit against blocked CPU summation)
    f32x16 c[RM][RN][2];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int q = 0; q < RN; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) c[r][q][j][i] = 0.f;
    float ra[PF][8 * RM], rb[PF][8 * RN];
    double bsum = 0.0;                                       // epi.dbias: column sum of this workgroup's dY tiles (threads 0 .. 64 RM - 1)
    const bool do_bias = TA && epi.dbias != nullptr && bx == 0 && tid < 64 * RM;
    // A is k-contiguous unless transposed; B (stored [N, K] when TB) is k-contiguous when TB
    auto fetch = [&](int u, int k0) {
        if (FAST) {
            load_tile_fast<!TA, RM>(A, lda, m0, M, k0, kend, tid, ra[u]);
            load_tile_fast<TB, RN>(B, ldb, n0, N, k0, kend, tid, rb[u]);
        } else {
            load_tile<!TA, RM>(A, lda, m0, M, k0, kend, vecA != 0, tid, ra[u]);
            load_tile<TB, RN>(B, ldb, n0, N, k0, kend, vecB != 0, tid, rb[u]);
        }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (kbeg + u * TK < kend) fetch(u, kbeg + u * TK);
    for (int kb = kbeg; kb < kend; kb += PF * TK) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int k0 = kb + u * TK;
            if (k0 < kend) {                                     // (uniform over the workgroup: the barriers below are safe)
                store_tile<!TA, RM>(As, tid, ra[u], k0, kend);
                store_tile<TB, RN>(Bs, tid, rb[u], k0, kend);
                __syncthreads();
                if (do_bias) {
#pragma unroll
                    for (int kk = 0; kk < TK; ++kk) bsum += (double)As[kk][tid];
                }
                if (k0 + PF * TK < kend) fetch(u, k0 + PF * TK);  // refill the slot just consumed: PF tiles stay in flight
#pragma unroll
                for (int kk = 0; kk < TK; kk += 2) {
                    float a[RM], b[RN];
#pragma unroll
                    for (int r = 0; r < RM; ++r) a[r] = As[kk + (lane >> 5)][wm + 32 * r + (lane & 31)];
#pragma unroll
                    for (int q = 0; q < RN; ++q) b[q] = Bs[kk + (lane >> 5)][wn + 32 * q + (lane & 31)];
#pragma unroll
                    for (int r = 0; r < RM; ++r)
#pragma unroll
                        for (int q = 0; q < RN; ++q)
                            c[r][q][(kk >> 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[q], c[r][q][(kk >> 1) & 1], 0, 0, 0);
                }
                __syncthreads();
            }
        }
    }
    if (do_bias && m0 + tid < M) {
        if (part) part[(long)nz * M * N + (long)bz * M + m0 + tid] = (float)bsum;     // behind the partial tiles
        else epi.dbias[m0 + tid] += (float)bsum;
    }
    // C/D map of the 32x32 forms: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int q = 0; q < RN; ++q) {
        const int col = n0 + wn + 32 * q + (lane & 31);
        if (col >= N) continue;
        const float bv = (bias && !part) ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int row = m0 + wm + 32 * r + (s & 3) + 8 * (s >> 2) + 4 * (lane >> 5);
                if (row >= M) continue;
                const float v = c[r][q][0][s] + c[r][q][1][s];
                if (part) part[((long)bz * M + row) * N + col] = v;
                else if (epi.act) gemm_epilogue(epi, v + bv, C, (long)row * ldc + col, (long)row * N + col);
                else {
                    float* o = C + (long)row * ldc + col;
                    *o = acc ? *o + (v + bv) : v + bv;
                }
            }
    }
}

template <bool TA, bool TB, int RM, int RN, bool FAST, int PF>
__global__ __launch_bounds__(256) void k_gemm(int M, int N, int K, int kchunk, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                              float* __restrict__ C, int ldc, const float* __restrict__ bias, int acc, float* __restrict__ part, int vecA, int vecB, GemmEpi epi) {
    __shared__ __attribute__((aligned(16))) float As[TK][TileLd<!TA, RM>::value];
    __shared__ __attribute__((aligned(16))) float Bs[TK][TileLd<TB, RN>::value];
    // Workgroups go to the 8 XCDs (one L2 each) round-robin in dispatch order, x fastest: the column tiles of one row tile — which read
    // the same 64 rows of A — would land on different XCDs and fetch those rows once each (N = 256: four times).  Remapped so that a
    // row tile's column tiles are the ids xcd, xcd + 8, xcd + 16, ...: same XCD, dispatched together.  (Whole groups of 8 row tiles
    // only; the remainder and split-K launches keep the plain order.)
    int bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.x > 1 && gridDim.z == 1) {
        const int NT = gridDim.x, lin = blockIdx.y * NT + blockIdx.x;
        if (lin < NT * 8 * ((int)gridDim.y / 8)) {
            const int xcd = lin & 7, slot = lin >> 3;
            by = (slot / NT) * 8 + xcd;
            bx = slot % NT;
        }
    }
    gemm_tile<TA, TB, RM, RN, FAST, PF>(M, N, K, kchunk, A, lda, B, ldb, C, ldc, bias, acc, part, vecA, vecB, epi, bx, by, (int)blockIdx.z, (int)gridDim.z, As, Bs);
}

// Several weight-gradient products (tA = 1, tB = 0, acc = 1) as ONE launch: a workgroup looks its job up in the table that travels in
// the kernel arguments (jobs lo .. hi - 1, their workgroups numbered consecutively from wg0) and runs that job's tile exactly as k_gemm
// would — same plan, same slices, same arithmetic, bit-identical results.  The QM9 training batch's backward has 157 such products of
// 6 - 76 us, most of them a few hundred workgroups of a few K steps: on their own each is a launch that cannot fill the chip.
template <bool FAST>
__global__ __launch_bounds__(256) void k_gemm_group(GemmGroup G, int lo, int hi) {
    __shared__ __attribute__((aligned(16))) float As[TK][TileLd<false, 1>::value];
    __shared__ __attribute__((aligned(16))) float Bs[TK][TileLd<false, 1>::value];
    int w = (int)blockIdx.x, ji = lo;
    while (ji + 1 < hi && w >= G.j[ji + 1].wg0) ++ji;
    const GemmGroupJob& J = G.j[ji];
    w -= J.wg0;
    const int bx = w % J.nx, by = (w / J.nx) % J.ny, bz = w / (J.nx * J.ny);
    GemmEpi e; e.act = 0; e.out2 = nullptr; e.drop.p = 0.f; e.drop.seed = 0; e.drop.site = 0; e.dbias = J.dbias;
    gemm_tile<true, false, 1, 1, FAST, 1>(J.M, J.N, J.K, J.kchunk, J.A, J.lda, J.B, J.ldb, J.C, J.ldc, nullptr, 1, J.part, J.vecA, J.vecB, e, bx, by, bz, J.nz, As, Bs);
}

// Sum of the split-K partial tiles.  A block owns 32 consecutive outputs; its eight rows of 32 lanes take the slices z = g, g + 8, ...
// (eight loads in flight each, coalesced 128-byte rows), the eight partial sums meet in LDS and are added in the order g = 0 .. 7:
//     total = (...((s_0 + s_1) + s_2) ... + s_7),   s_g = sum over z = g (mod 8), ascending
// — a fixed order (tests/emul/emul_gemm.cpp mirrors it).  One thread per output walking every slice took 7 us at 81 slices and forced a
// second level above 64; this form needs one launch up to the plan's 512 slices.  Behind the output blocks, blocks of 32 rows do the same
// for the partial bias sums (epi.dbias).
__device__ __forceinline__ void splitk_sum_block(long blk, int M, int N, int nsplit, const float* __restrict__ part, float* __restrict__ C, int ldc,
                                                 const float* __restrict__ bias, int acc, const GemmEpi& epi, float (*sh)[32]) {
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long MN = (long)M * N, nmain = (MN + 31) / 32;
    const bool is_bias = blk >= nmain;
    const long i = (is_bias ? blk - nmain : blk) * 32 + e;
    const long count = is_bias ? (long)M : MN, stride = count;
    const float* src = is_bias ? part + (long)nsplit * MN : part;
    float s = 0.f;
    if (i < count) {
        int z = g;
        for (; z + 56 < nsplit; z += 64) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(long)(z + 8 * j) * stride + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; z < nsplit; z += 8) s += src[(long)z * stride + i];
    }
    sh[g][e] = s;
    __syncthreads();
    if (g != 0 || i >= count) return;
    float t = sh[0][e];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += sh[q][e];
    if (is_bias) { epi.dbias[i] += t; return; }
    const int row = (int)(i / N), col = (int)(i % N);
    t += bias ? bias[col] : 0.f;
    if (epi.act) { gemm_epilogue(epi, t, C, (long)row * ldc + col, i); return; }
    float* o = C + (long)row * ldc + col;
    *o = acc ? *o + t : t;
}
__global__ __launch_bounds__(256) void k_splitk_sum(int M, int N, int nsplit, const float* __restrict__ part, float* __restrict__ C, int ldc,
                                                    const float* __restrict__ bias, int acc, GemmEpi epi) {
    __shared__ float sh[8][32];
    splitk_sum_block((long)blockIdx.x, M, N, nsplit, part, C, ldc, bias, acc, epi, sh);
}
// the sums of every split job of a group in one launch (blocks numbered consecutively from sb0; jobs without slices own no blocks)
__global__ __launch_bounds__(256) void k_splitk_sum_group(GemmGroup G) {
    __shared__ float sh[8][32];
    int ji = 0;
    while (ji + 1 < G.n && (int)blockIdx.x >= G.j[ji + 1].sb0) ++ji;
    const GemmGroupJob& J = G.j[ji];
    GemmEpi e; e.act = 0; e.out2 = nullptr; e.drop.p = 0.f; e.drop.seed = 0; e.drop.site = 0; e.dbias = J.dbias;
    splitk_sum_block((long)((int)blockIdx.x - J.sb0), J.M, J.N, J.nz, J.part, J.C, J.ldc, nullptr, 1, e, sh);
}

template <bool FAST, int PF>
static void launch(hipStream_t s, int tA, int tB, dim3 grid, int M, int N, int K, int kchunk, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   const float* bias, int acc, float* part, int vecA, int vecB, GemmEpi epi) {
    const dim3 block(256);
    if (tA && tB) hipLaunchKernelGGL((k_gemm<true, true, 1, 1, FAST, PF>), grid, block, 0, s, M, N, K, kchunk, A, lda, B, ldb, C, ldc, bias, acc, part, vecA, vecB, epi);
    else if (tA) hipLaunchKernelGGL((k_gemm<true, false, 1, 1, FAST, PF>), grid, block, 0, s, M, N, K, kchunk, A, lda, B, ldb, C, ldc, bias, acc, part, vecA, vecB, epi);
    else if (tB) hipLaunchKernelGGL((k_gemm<false, true, 1, 1, FAST, PF>), grid, block, 0, s, M, N, K, kchunk, A, lda, B, ldb, C, ldc, bias, acc, part, vecA, vecB, epi);
    else hipLaunchKernelGGL((k_gemm<false, false, 1, 1, FAST, PF>), grid, block, 0, s, M, N, K, kchunk, A, lda, B, ldb, C, ldc, bias, acc, part, vecA, vecB, epi);
}

static bool gemm_fast(int tA, int tB, int M, int N, int K, int vecA, int vecB) {
    const bool kc = !tA || tB;                               // some operand is k-contiguous
#ifdef JODO_X_GEMM_NO_RAGGED                                  // experiment builds: the round-4 rule (whole K tiles only)
    if (K % TK) return false;
#endif
    return vecA && vecB && K >= 4 && (!kc || (K % 4) == 0) && (tA ? (M % 4) == 0 && M >= 4 : M >= 1) && (tB ? N >= 1 : (N % 4) == 0 && N >= 4);
}
static FILE* shape_log() {
    // JODO_TRAIN_GEMM_LOG=<file>: one line per product (tools/train_gemm_shapes.py ranks a step's products by shape)
    static FILE* f = [] { const char* n = getenv("JODO_TRAIN_GEMM_LOG"); return n ? fopen(n, "a") : (FILE*)nullptr; }();
    return f;
}

void gemm(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
          const float* bias, int acc, float* ws, size_t ws_floats, const GemmEpi* epi_in) {
    if (M <= 0 || N <= 0) return;
    if (shape_log()) { fprintf(shape_log(), "%d %d %d %d %d %d %d %d\n", tA, tB, M, N, K, lda, ldb, ldc); fflush(shape_log()); }
    GemmEpi epi;
    if (epi_in) epi = *epi_in; else { epi.act = 0; epi.out2 = nullptr; epi.drop.p = 0.f; epi.drop.seed = 0; epi.drop.site = 0; epi.dbias = nullptr; }
    const GemmPlan p = gemm_plan(tA, M, N, K, ws != nullptr, ws_floats);
    float* part = p.nsplit > 1 ? ws : nullptr;
    const dim3 grid((N + 64 * p.rn - 1) / (64 * p.rn), (M + 64 * p.rm - 1) / (64 * p.rm), p.nsplit);
    // 16-byte loads need an aligned base and a row stride that keeps every quad aligned (column offsets of sliced weights included
    // in the base pointer); otherwise the element-wise path
    const int vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
    const int vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0 && (ldb & 3) == 0) ? 1 : 0;
    if (p.rm != 1 || p.rn != 1) { fprintf(stderr, "jodo gemm: tile %d x %d is not instantiated\n", p.rm, p.rn); abort(); }
    // FAST: 16-byte addressable operands with at least one whole quad (m- / n-contiguous layouts) or row (k-contiguous) to clamp to; a
    // ragged end of K is masked per quad, which k-contiguous operands can only do when K is a multiple of 4
    const bool fast = gemm_fast(tA, tB, M, N, K, vecA, vecB);
    const long wgs = (long)grid.x * grid.y * grid.z;
    if (fast && wgs <= 512 && (K < p.kchunk ? K : p.kchunk) > 2 * TK) launch<true, 4>(s, tA, tB, grid, M, N, K, p.kchunk, A, lda, B, ldb, C, ldc, bias, acc, part, vecA, vecB, epi);
    else if (fast) launch<true, 1>(s, tA, tB, grid, M, N, K, p.kchunk, A, lda, B, ldb, C, ldc, bias, acc, part, vecA, vecB, epi);
    else launch<false, 1>(s, tA, tB, grid, M, N, K, p.kchunk, A, lda, B, ldb, C, ldc, bias, acc, part, vecA, vecB, epi);
    if (p.nsplit > 1) {
        const long blocks = ((long)M * N + 31) / 32 + (epi.dbias ? (M + 31) / 32 : 0);
        hipLaunchKernelGGL(k_splitk_sum, dim3((unsigned)blocks), dim3(256), 0, s, M, N, p.nsplit, part, C, ldc, bias, acc, epi);
    }
}

// dW_i[M_i, N_i] += A_i^T B_i (+ dbias_i) for every job: the plans of gemm() — every job is planned against `plan_floats` of scratch,
// as its own launch would be — packed into `ws`, one k_gemm_group launch for the jobs that take the branch-free loader, one for the
// others ([3 x 256]: no whole quad along m), one k_splitk_sum_group.  Groups are cut where the table or the scratch is full.
void gemm_dw_group(hipStream_t s, const GemmJob* jobs, int n, float* ws, size_t ws_floats, size_t plan_floats) {
    int i = 0;
    while (i < n) {
        GemmGroup G;
        GemmGroupJob slow[GEMM_GROUP_MAX];
        int nf = 0, ns = 0, wgf = 0, wgs = 0;
        size_t used = 0;
        for (; i < n && nf + ns < GEMM_GROUP_MAX; ++i) {
            const GemmJob& q = jobs[i];
            if (q.M <= 0 || q.N <= 0) continue;
            const GemmPlan p = gemm_plan(1, q.M, q.N, q.K, true, plan_floats);
            const size_t need = p.nsplit > 1 ? (size_t)p.nsplit * ((size_t)q.M * q.N + q.M) : 0;
            if (used + need > ws_floats) {
                if (nf + ns == 0) { fprintf(stderr, "jodo gemm_dw_group: %zu floats of scratch for one product, %zu given\n", need, ws_floats); abort(); }
                break;
            }
            if (shape_log()) { fprintf(shape_log(), "1 0 %d %d %d %d %d %d\n", q.M, q.N, q.K, q.lda, q.ldb, q.ldc); fflush(shape_log()); }
            GemmGroupJob J;
            J.A = q.A; J.B = q.B; J.C = q.C; J.dbias = q.dbias; J.part = p.nsplit > 1 ? ws + used : nullptr;
            J.M = q.M; J.N = q.N; J.K = q.K; J.kchunk = p.kchunk; J.lda = q.lda; J.ldb = q.ldb; J.ldc = q.ldc;
            J.nx = (q.N + 63) / 64; J.ny = (q.M + 63) / 64; J.nz = p.nsplit;
            J.vecA = ((reinterpret_cast<uintptr_t>(q.A) & 15) == 0 && (q.lda & 3) == 0) ? 1 : 0;
            J.vecB = ((reinterpret_cast<uintptr_t>(q.B) & 15) == 0 && (q.ldb & 3) == 0) ? 1 : 0;
            J.wg0 = 0; J.sb0 = 0;
            used += (need + 63) / 64 * 64;
            if (gemm_fast(1, 0, q.M, q.N, q.K, J.vecA, J.vecB)) { J.wg0 = wgf; wgf += J.nx * J.ny * J.nz; G.j[nf++] = J; }
            else { J.wg0 = wgs; wgs += J.nx * J.ny * J.nz; slow[ns++] = J; }
        }
        for (int k = 0; k < ns; ++k) G.j[nf + k] = slow[k];
        G.n = nf + ns;
        int sb = 0;
        for (int k = 0; k < G.n; ++k) {
            GemmGroupJob& J = G.j[k];
            J.sb0 = sb;
            if (J.nz > 1) sb += (int)(((long)J.M * J.N + 31) / 32) + (J.dbias ? (J.M + 31) / 32 : 0);
        }
        if (wgf) hipLaunchKernelGGL((k_gemm_group<true>), dim3((unsigned)wgf), dim3(256), 0, s, G, 0, nf);
        if (wgs) hipLaunchKernelGGL((k_gemm_group<false>), dim3((unsigned)wgs), dim3(256), 0, s, G, nf, G.n);
        if (sb) hipLaunchKernelGGL(k_splitk_sum_group, dim3((unsigned)sb), dim3(256), 0, s, G);
    }
}

}  // namespace jt
