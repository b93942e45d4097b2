// Training-side kernels of the DGT: forward with saved activations and the backward of every phase of
//   DGT_concat.forward / Cond_DGT_concat.forward     /root/reference/models/mol_gnn.py:491-594, :687-794
//   EquivariantMixBlock.forward                      models/mol_gnn.py:270-322
//   TransMixLayer.forward / message                  models/layers.py:131-186   (F.dropout on alpha, :179, is an identity: the block builds the layer with its default dropout = 0, mol_gnn.py:230)
//   MultiCondEquiUpdate.forward                      models/mol_gnn.py:71-94
//   CondGaussianLayer / gaussian, CoorsNorm          models/layers.py:291-295, :328-334, :344-347
// i.e. what loss.backward() (losses.py:286-385) differentiates.  Everything that is not a dense projection lives here; the
// projections and their two gradient GEMMs are train_gemm.hip.
//
// Layout (dgt_train.hip builds the tables): molecules in caller order, node rows Nn = sum n_b, edge rows R = sum n_b^2 — the dense
// n x n tile of each molecule, row (a, c) at edge_off[b] + a n + c, a = row atom = attention source, c = column = attention
// target; diagonal rows are carried and masked where the reference has no edge.  Per-molecule modulation rows mods[b, .].
//
// Kernel style: one thread per output element (coalesced along the feature index) or per reduced (row | atom | molecule, feature);
// reductions are plain loops in a fixed order (bit-deterministic, no atomics).  No LDS, no barriers, no cross-lane operations —
// first-correct kernels of the training row; the hot inference path (dgt_kernels_*.h) is where the strip model lives.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "train_common.h"

namespace jt {

#define JT_IDX(n)                                                          \
    const long i_ = (long)blockIdx.x * (long)blockDim.x + (long)threadIdx.x; \
    if (i_ >= (long)(n)) return
// one thread per element: n threads in workgroups of 256
#define JT_LAUNCH(kernel, n, stream, ...) hipLaunchKernelGGL(kernel, dim3((unsigned)(((long)(n) + 255) / 256)), dim3(256), 0, stream, __VA_ARGS__)

struct Topo {                 // device tables of one batch
    int B, Nn, R, N;
    const int* node_off;      // [B + 1]
    const int* edge_off;      // [B + 1]
    const int* nn;            // [B]
    const int* node_mol;      // [Nn]
    const int* edge_mol;      // [R]
    const int* edge_a;        // [R] global node row of the row atom
    const int* edge_c;        // [R] global node row of the column atom
    // edge rows of a molecule cut into chunks (at most 64 per molecule, at least 32 rows each): two-level per-molecule sums
    int NC;
    const int* ec_off;        // [NC + 1] first row of every chunk
    const int* ec_mol_off;    // [B + 1] first chunk of every molecule
};

// ================================================================ elementwise =================================================
// y = SiLU(x) * dropout
__global__ void k_silu_fwd(long n, const float* __restrict__ x, float* __restrict__ y, Drop d) {
    JT_IDX(n);
    y[i_] = silu_f(x[i_]) * drop_mul(d, (unsigned long long)i_);
}
// dx = dy * dropout * SiLU'(x)     (dx may alias dy)
__global__ void k_silu_bwd(long n, const float* __restrict__ x, const float* dy, float* dx, Drop d) {
    JT_IDX(n);
    dx[i_] = dy[i_] * drop_mul(d, (unsigned long long)i_) * silu_grad(x[i_]);
}
__global__ void k_tanh_fwd(long n, float* __restrict__ x) {
    JT_IDX(n);
    x[i_] = tanhf(x[i_]);
}
// dpre = dt * (1 - t^2), in place on dt
__global__ void k_tanh_bwd(long n, const float* __restrict__ t, float* __restrict__ dt) {
    JT_IDX(n);
    dt[i_] = dt[i_] * (1.f - t[i_] * t[i_]);
}
__global__ void k_gelu_fwd(long n, const float* __restrict__ x, float* __restrict__ y) {
    JT_IDX(n);
    y[i_] = 0.5f * x[i_] * (1.f + erff(x[i_] * 0.70710678118654752f));
}
__global__ void k_gelu_bwd(long n, const float* __restrict__ x, const float* dy, float* dx) {
    JT_IDX(n);
    const float v = x[i_];
    dx[i_] = dy[i_] * (0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * expf(-0.5f * v * v));
}
// y = x * dropout (y may alias x)
__global__ void k_drop(long n, const float* x, float* y, Drop d) {
    JT_IDX(n);
    y[i_] = x[i_] * drop_mul(d, (unsigned long long)i_);
}
// x = 0 when *flag == 0
__global__ void k_scale_if_zero(long n, float* __restrict__ x, const int* __restrict__ flag) {
    JT_IDX(n);
    if (*flag == 0) x[i_] = 0.f;
}
__global__ void k_add(long n, float* __restrict__ y, const float* __restrict__ x) {
    JT_IDX(n);
    y[i_] += x[i_];
}
// strided copy: dst[r, dcol + f] = src[r, scol + f]   (acc: +=)
__global__ void k_copy2d(long rows, int F, const float* __restrict__ src, int lds, int scol, float* __restrict__ dst, int ldd, int dcol, int acc) {
    JT_IDX(rows * F);
    const long r = i_ / F; const int f = (int)(i_ % F);
    const float v = src[r * lds + scol + f];
    float* o = dst + r * ldd + dcol + f;
    *o = acc ? *o + v : v;
}

// ================================================================ batched modulation projections ===============================
// Every time-MLP of every block (node_time_mlp.1, edge_time_mlp.1, equi_update.time_mlp.1, dist_layer.time_mlp.1, and the top-level
// dist_layer.time_mlp.1) is Linear(SiLU(time_emb)): 4 L + 1 projections of the SAME [B, T] input.  As launches of their own they were
// 33 forward products (+ their split-K sums) and 66 backward products of 128 rows each at QM9's training batch — 13 - 18 us apiece
// whatever their size.  Round 5: the weights are gathered into one [Mtot, T] matrix per step (they change every optimiser step), ONE
// product gives all modulation rows [B, Mtot], the backward writes every modulation gradient into one [B, Mtot] array and ends with ONE
// weight-gradient and ONE input-gradient product; the rows are copied back to the separate gradient tensors.
constexpr int MOD_MAX = 4 * 16 + 1;   // up to 16 blocks (the tables travel as kernel arguments: 2 KB)
struct ModTable {
    int n;
    const float* w[MOD_MAX];      // [F, T]
    const float* bias[MOD_MAX];   // [F]
    float* out[MOD_MAX];          // forward: this projection's rows [B, F]
    int F[MOD_MAX], col[MOD_MAX]; // rows of the projection; its first column inside [., Mtot]
};
struct ModGradTable { int n; float* gw[MOD_MAX]; float* gb[MOD_MAX]; int F[MOD_MAX], col[MOD_MAX]; };
// blockIdx.y = projection; Wall[(col + f) T + k] = w[f T + k], ball[col + f] = bias[f]
__global__ void k_mod_gather(ModTable M, int T, float* __restrict__ Wall, float* __restrict__ ball) {
    const int i = blockIdx.y;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)M.F[i] * T;
    if (e >= n) return;
    Wall[(long)M.col[i] * T + e] = M.w[i][e];
    if (e < M.F[i]) ball[M.col[i] + e] = M.bias[i][e];
}
// blockIdx.y = projection; out_i[b, f] = all[b, col_i + f]
__global__ void k_mod_scatter(ModTable M, int B, int Mtot, const float* __restrict__ all) {
    const int i = blockIdx.y;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)B * M.F[i]) return;
    const long b = e / M.F[i]; const int f = (int)(e % M.F[i]);
    M.out[i][e] = all[b * Mtot + M.col[i] + f];
}
// the inverse of k_mod_scatter: all[b, col_i + f] = out_i[b, f]   (gradient rows of projections that share an input, gathered for ONE product)
__global__ void k_mod_gather_cols(ModTable M, int B, int Mtot, float* __restrict__ all) {
    const int i = blockIdx.y;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)B * M.F[i]) return;
    const long b = e / M.F[i]; const int f = (int)(e % M.F[i]);
    all[b * Mtot + M.col[i] + f] = M.out[i][e];
}
// blockIdx.y = projection; gw_i[f, k] += dWall[(col_i + f) T + k], gb_i[f] += dball[col_i + f]
__global__ void k_mod_scatter_grads(ModGradTable M, int T, const float* __restrict__ dWall, const float* __restrict__ dball) {
    const int i = blockIdx.y;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)M.F[i] * T;
    if (e >= n) return;
    M.gw[i][e] += dWall[(long)M.col[i] * T + e];
    if (e < M.F[i]) M.gb[i][e] += dball[M.col[i] + e];
}

// ================================================================ LayerNorm + modulate ========================================
// per-row mean and rstd (biased variance, eps 1e-6; LayerNorm(elementwise_affine=False), mol_gnn.py:234-245, :64) in two small
// launches: eight threads per row sum a contiguous eighth each (a wave reads whole cache lines; one thread per row walked 64
// different lines per load) of d = x - x[row, 0] and d^2 — shifted, so that the one-pass variance does not cancel — and a second
// launch combines the eight partial sums in double.
__global__ void k_row_part(long rows, int F, const float* __restrict__ x, float* __restrict__ part) {
    JT_IDX(rows * 8);
    const long r = i_ >> 3; const int p = (int)(i_ & 7), w = F / 8;
    const float* q = x + r * F;
    const float x0 = q[0];
    float s = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int f = p * w; f < (p + 1) * w; ++f) { const float d = q[f] - x0; s += d; s2 += d * d; }
    part[i_ * 2] = s; part[i_ * 2 + 1] = s2;
}
__global__ void k_row_stats(long rows, int F, const float* __restrict__ x, const float* __restrict__ part, float* __restrict__ mean, float* __restrict__ rstd) {
    JT_IDX(rows);
    double s = 0.0, s2 = 0.0;
    for (int p = 0; p < 8; ++p) { s += (double)part[(i_ * 8 + p) * 2]; s2 += (double)part[(i_ * 8 + p) * 2 + 1]; }
    const double md = s / F, var = s2 / F - md * md;
    mean[i_] = (float)((double)x[i_ * F] + md);
    rstd[i_] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-6));
}
// xhat = (x - mean) rstd;  y = xhat (1 + sc[mol]) + sh[mol]      (modulate, mol_gnn.py:12-13)
__global__ void k_ln_mod_fwd(long rows, int F, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                             const int* __restrict__ row_mol, const float* __restrict__ mods, int ldm, int sh_off, int sc_off,
                             float* __restrict__ xhat, float* __restrict__ y) {
    JT_IDX(rows * F);
    const long r = i_ / F; const int f = (int)(i_ % F);
    const float* m = mods + (long)row_mol[r] * ldm;
    const float xh = (x[i_] - mean[r]) * rstd[r];
    xhat[i_] = xh;
    y[i_] = xh * (1.f + m[sc_off + f]) + m[sh_off + f];
}
// backward, pass 1: per row c1 = mean_f(g), c2 = mean_f(g xhat) with g = dy (1 + sc) — eight partial sums per row, then combined
__global__ void k_ln_bwd_part(long rows, int F, const float* __restrict__ dy, const float* __restrict__ xhat, const int* __restrict__ row_mol,
                              const float* __restrict__ mods, int ldm, int sc_off, float* __restrict__ part) {
    JT_IDX(rows * 8);
    const long r = i_ >> 3; const int p = (int)(i_ & 7), w = F / 8;
    const float* m = mods + (long)row_mol[r] * ldm + sc_off;
    const float* d = dy + r * F; const float* xh = xhat + r * F;
    float a = 0.f, b = 0.f;
#pragma unroll 8
    for (int f = p * w; f < (p + 1) * w; ++f) { const float g = d[f] * (1.f + m[f]); a += g; b += g * xh[f]; }
    part[i_ * 2] = a; part[i_ * 2 + 1] = b;
}
__global__ void k_ln_bwd_stats(long rows, int F, const float* __restrict__ part, float* __restrict__ c1, float* __restrict__ c2) {
    JT_IDX(rows);
    double a = 0.0, b = 0.0;
    for (int p = 0; p < 8; ++p) { a += (double)part[(i_ * 8 + p) * 2]; b += (double)part[(i_ * 8 + p) * 2 + 1]; }
    c1[i_] = (float)(a / F); c2[i_] = (float)(b / F);
}
// pass 2: dx = rstd (g - c1 - xhat c2)         (acc: dx += ...)
__global__ void k_ln_bwd_apply(long rows, int F, const float* dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
                               const float* __restrict__ c1, const float* __restrict__ c2, const int* __restrict__ row_mol,
                               const float* __restrict__ mods, int ldm, int sc_off, float* dx, int acc) {
    JT_IDX(rows * F);
    const long r = i_ / F; const int f = (int)(i_ % F);
    const float g = dy[i_] * (1.f + mods[(long)row_mol[r] * ldm + sc_off + f]);
    const float v = rstd[r] * (g - c1[r] - xhat[i_] * c2[r]);
    dx[i_] = acc ? dx[i_] + v : v;
}

// ================================================================ segment / column reductions ==================================
// out[s, ocol + f] (+)= sum over rows r in [off[s], off[s + 1]) of a[r, f] * (b ? b[r, f] : 1)
// (db: dropout applied to b on the fly — b * mask, then the product: the same roundings as a k_drop pass over b in front)
__global__ void k_seg_colsum(int S, int F, const int* __restrict__ off, const float* __restrict__ a, const float* __restrict__ b,
                             float* __restrict__ out, int ldo, int ocol, int acc, Drop db) {
    JT_IDX((long)S * F);
    const int s = (int)(i_ / F), f = (int)(i_ % F);
    double t = 0.0;                                  // gradient sums cancel: accumulated in double, stored in float
    if (b && db.p > 0.f) {
#pragma unroll 8
        for (long r = off[s]; r < off[s + 1]; ++r) t += (double)(a[r * F + f] * (b[r * F + f] * drop_mul(db, (unsigned long long)(r * F + f))));
    } else if (b) {
#pragma unroll 8
        for (long r = off[s]; r < off[s + 1]; ++r) t += (double)(a[r * F + f] * b[r * F + f]);
    } else {
#pragma unroll 8
        for (long r = off[s]; r < off[s + 1]; ++r) t += (double)a[r * F + f];
    }
    float* o = out + (long)s * ldo + ocol + f;
    *o = acc ? *o + (float)t : (float)t;
}
// the same over the edge rows of every molecule in two levels (a molecule has up to 181^2 rows): part[c, f] = sum over chunk c,
// then out[mol, ocol + f] = sum over the molecule's chunks — fixed order, no atomics
__global__ void k_seg_part(int NC, int F, const int* __restrict__ ec_off, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part) {
    JT_IDX((long)NC * F);
    const int c = (int)(i_ / F), f = (int)(i_ % F);
    double t = 0.0;
    if (b) {
#pragma unroll 8
        for (long r = ec_off[c]; r < ec_off[c + 1]; ++r) t += (double)(a[r * F + f] * b[r * F + f]);
    } else {
#pragma unroll 8
        for (long r = ec_off[c]; r < ec_off[c + 1]; ++r) t += (double)a[r * F + f];
    }
    part[i_] = (float)t;
}
// the same with b = ehat computed on the fly: ehat[(a, c), f] = p[a, f] + p[c, f] + bias[f] (node2edge_lin of both atoms of the edge)
// — the roundings of k_edge_bcast followed by k_seg_part, without the [R, F] array in between
__global__ void k_seg_part_ehat(Topo t, int F, const float* __restrict__ a, const float* __restrict__ p, const float* __restrict__ bias,
                                float* __restrict__ part) {
    JT_IDX((long)t.NC * F);
    const int c = (int)(i_ / F), f = (int)(i_ % F);
    const float bv = bias[f];
    double s = 0.0;
#pragma unroll 8
    for (long r = t.ec_off[c]; r < t.ec_off[c + 1]; ++r)
        s += (double)(a[r * F + f] * (p[(long)t.edge_a[r] * F + f] + p[(long)t.edge_c[r] * F + f] + bv));
    part[i_] = (float)s;
}
__global__ void k_seg_fin(int S, int F, const int* __restrict__ mol_off, const float* __restrict__ part, float* __restrict__ out, int ldo, int ocol, int acc) {
    JT_IDX((long)S * F);
    const int s = (int)(i_ / F), f = (int)(i_ % F);
    double t = 0.0;
#pragma unroll 8
    for (int c = mol_off[s]; c < mol_off[s + 1]; ++c) t += (double)part[(long)c * F + f];
    float* o = out + (long)s * ldo + ocol + f;
    *o = acc ? *o + (float)t : (float)t;
}
// both sums a LayerNorm + modulate backward needs, in one pass over the rows (round 5: the two used to be two launches each):
// out[s, c1 + f] = sum a[r, f],  out[s, c2 + f] = sum a[r, f] b[r, f]   (same loops, same order: bit-identical to the separate sums)
__global__ void k_seg_colsum2(int S, int F, const int* __restrict__ off, const float* __restrict__ a, const float* __restrict__ b,
                              float* __restrict__ out, int ldo, int c1, int c2) {
    JT_IDX((long)S * F);
    const int s = (int)(i_ / F), f = (int)(i_ % F);
    double t = 0.0, u = 0.0;
#pragma unroll 8
    for (long r = off[s]; r < off[s + 1]; ++r) { const float av = a[r * F + f]; t += (double)av; u += (double)(av * b[r * F + f]); }
    out[(long)s * ldo + c1 + f] = (float)t;
    out[(long)s * ldo + c2 + f] = (float)u;
}
__global__ void k_seg_part2(int NC, int F, const int* __restrict__ ec_off, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part) {
    JT_IDX((long)NC * F);
    const int c = (int)(i_ / F), f = (int)(i_ % F);
    double t = 0.0, u = 0.0;
#pragma unroll 8
    for (long r = ec_off[c]; r < ec_off[c + 1]; ++r) { const float av = a[r * F + f]; t += (double)av; u += (double)(av * b[r * F + f]); }
    part[(long)c * 2 * F + f] = (float)t;
    part[(long)c * 2 * F + F + f] = (float)u;
}
__global__ void k_seg_fin2(int S, int F, const int* __restrict__ mol_off, const float* __restrict__ part, float* __restrict__ out, int ldo, int c1, int c2) {
    JT_IDX((long)S * 2 * F);
    const int s = (int)(i_ / (2 * F)), g = (int)(i_ % (2 * F));
    double t = 0.0;
#pragma unroll 8
    for (int c = mol_off[s]; c < mol_off[s + 1]; ++c) t += (double)part[(long)c * 2 * F + g];
    out[(long)s * ldo + (g < F ? c1 + g : c2 + g - F)] = (float)t;
}
// column sums of a [rows, F] (row stride lda) in two deterministic stages: part[c, f] = sum of chunk c, then out[f] (+)= sum_c
__global__ void k_colsum_part(long rows, int F, int chunk, const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, float* __restrict__ part) {
    const long nchunks = (rows + chunk - 1) / chunk;
    JT_IDX(nchunks * F);
    const long c = i_ / F; const int f = (int)(i_ % F);
    const long r1 = (c + 1) * chunk < rows ? (c + 1) * chunk : rows;
    double t = 0.0;
    if (b) {
#pragma unroll 8
        for (long r = c * chunk; r < r1; ++r) t += (double)(a[r * lda + f] * b[r * ldb + f]);
    } else {
#pragma unroll 8
        for (long r = c * chunk; r < r1; ++r) t += (double)a[r * lda + f];
    }
    part[i_] = (float)t;
}
__global__ void k_colsum_fin(long nchunks, int F, const float* __restrict__ part, float* __restrict__ out, int acc) {
    JT_IDX(F);
    double t = 0.0;
#pragma unroll 8
    for (long c = 0; c < nchunks; ++c) t += (double)part[c * F + i_];
    out[i_] = acc ? out[i_] + (float)t : (float)t;
}

// The second (and third) stages of the reductions above, MANY IN ONE LAUNCH (round 5, second half): the backward queues them — every
// queued reduction keeps its partial sums in a region of its own — and runs the queue once per block.  Same loops as k_seg_fin,
// k_seg_fin2, k_colsum_fin and a second-level k_colsum_part (chunk rows per output row): bit-identical to the launch-each form.
enum { FIN_SEG = 0, FIN_SEG2 = 1, FIN_COL = 2, FIN_COLPART = 3 };
struct FinJob { const float* part; float* out; const int* off; int kind, F, n, ldo, c1, c2, acc, blk0; };
#define FIN_MAX 24
struct FinTable { int n; FinJob j[FIN_MAX]; };
inline long fin_threads(const FinJob& J) {
    if (J.kind == FIN_SEG) return (long)J.n * J.F;
    if (J.kind == FIN_SEG2) return (long)J.n * 2 * J.F;
    if (J.kind == FIN_COL) return J.F;
    return (long)((J.n + J.c2 - 1) / J.c2) * J.F;             // FIN_COLPART: c2 = rows per chunk
}
__global__ void k_fin_group(FinTable T) {
    int ji = 0;
    const int blk = (int)blockIdx.x;
    while (ji + 1 < T.n && blk >= T.j[ji + 1].blk0) ++ji;
    const FinJob& J = T.j[ji];
    const long i_ = (long)(blk - J.blk0) * (long)blockDim.x + (long)threadIdx.x;
    const int F = J.F;
    if (J.kind == FIN_SEG) {
        if (i_ >= (long)J.n * F) return;
        const int s = (int)(i_ / F), f = (int)(i_ % F);
        double t = 0.0;
#pragma unroll 8
        for (int c = J.off[s]; c < J.off[s + 1]; ++c) t += (double)J.part[(long)c * F + f];
        float* o = J.out + (long)s * J.ldo + J.c1 + f;
        *o = J.acc ? *o + (float)t : (float)t;
    } else if (J.kind == FIN_SEG2) {
        if (i_ >= (long)J.n * 2 * F) return;
        const int s = (int)(i_ / (2 * F)), g = (int)(i_ % (2 * F));
        double t = 0.0;
#pragma unroll 8
        for (int c = J.off[s]; c < J.off[s + 1]; ++c) t += (double)J.part[(long)c * 2 * F + g];
        J.out[(long)s * J.ldo + (g < F ? J.c1 + g : J.c2 + g - F)] = (float)t;
    } else if (J.kind == FIN_COL) {
        if (i_ >= F) return;
        double t = 0.0;
#pragma unroll 8
        for (long c = 0; c < J.n; ++c) t += (double)J.part[c * F + i_];
        J.out[i_] = J.acc ? J.out[i_] + (float)t : (float)t;
    } else {
        const long chunk = J.c2, nch = (J.n + chunk - 1) / chunk;
        if (i_ >= nch * F) return;
        const long c = i_ / F; const int f = (int)(i_ % F);
        const long r1 = (c + 1) * chunk < J.n ? (c + 1) * chunk : J.n;
        double t = 0.0;
#pragma unroll 8
        for (long r = c * chunk; r < r1; ++r) t += (double)J.part[r * F + f];
        J.out[i_] = (float)t;
    }
}

// x_i *= scale for up to 16 small arrays of n floats (one launch at the end of the backward: see the node2edge bias gradients)
struct ScaleTable { int n_arrays; float* x[16]; };
__global__ void k_scale_arrays(ScaleTable T, int n, float scale) {
    JT_IDX((long)T.n_arrays * n);
    T.x[i_ / n][i_ % n] *= scale;
}

// ================================================================ gates, broadcasts between node and edge arrays ================
// y = a + g[mol] * b
__global__ void k_gate_add(long rows, int F, const float* __restrict__ a, const float* __restrict__ b, const int* __restrict__ row_mol,
                           const float* __restrict__ mods, int ldm, int g_off, float* __restrict__ y) {
    JT_IDX(rows * F);
    const long r = i_ / F; const int f = (int)(i_ % F);
    y[i_] = a[i_] + mods[(long)row_mol[r] * ldm + g_off + f] * b[i_];
}
// db = g[mol] * dy   (acc: +=)
__global__ void k_gate_bwd(long rows, int F, const float* __restrict__ dy, const int* __restrict__ row_mol, const float* __restrict__ mods,
                           int ldm, int g_off, float* db, int acc) {
    JT_IDX(rows * F);
    const long r = i_ / F; const int f = (int)(i_ % F);
    const float v = mods[(long)row_mol[r] * ldm + g_off + f] * dy[i_];
    db[i_] = acc ? db[i_] + v : v;
}
// db = (g[mol] * dy) * dropout — the gate's backward and the dropout's backward behind it in one pass (same roundings as the two)
__global__ void k_gate_drop_bwd(long rows, int F, const float* __restrict__ dy, const int* __restrict__ row_mol, const float* __restrict__ mods,
                                int ldm, int g_off, float* __restrict__ db, Drop d) {
    JT_IDX(rows * F);
    const long r = i_ / F; const int f = (int)(i_ % F);
    db[i_] = (mods[(long)row_mol[r] * ldm + g_off + f] * dy[i_]) * drop_mul(d, (unsigned long long)i_);
}
// y = a + g[mol] * (b * dropout) — dropout and the gated residual behind it in one pass (same roundings as k_drop + k_gate_add)
__global__ void k_drop_gate_add(long rows, int F, const float* __restrict__ a, const float* __restrict__ b, Drop d, const int* __restrict__ row_mol,
                                const float* __restrict__ mods, int ldm, int g_off, float* __restrict__ y) {
    JT_IDX(rows * F);
    const long r = i_ / F; const int f = (int)(i_ % F);
    y[i_] = a[i_] + mods[(long)row_mol[r] * ldm + g_off + f] * (b[i_] * drop_mul(d, (unsigned long long)i_));
}
// edge rows: y[(a, c), f] = base[(a, c), f] + g * (p[a, f] + q[c, f] + bias[f]);  g = mods[mol, g_off + f] or 1 (mods NULL); base NULL = 0
__global__ void k_edge_bcast(Topo t, int F, const float* __restrict__ base, const float* __restrict__ p, const float* __restrict__ q,
                             const float* __restrict__ bias, const float* __restrict__ mods, int ldm, int g_off, float* __restrict__ y) {
    JT_IDX((long)t.R * F);
    const long r = i_ / F; const int f = (int)(i_ % F);
    const float v = p[(long)t.edge_a[r] * F + f] + q[(long)t.edge_c[r] * F + f] + (bias ? bias[f] : 0.f);
    const float g = mods ? mods[(long)t.edge_mol[r] * ldm + g_off + f] : 1.f;
    y[i_] = (base ? base[i_] : 0.f) + g * v;
}
// node sums of an edge array: rowsum[a, f] = sum_c x[(a, c), f]; colsum[c, f] = sum_a x[(a, c), f]   (either output may be NULL; acc);
// rowsum == colsum: that array gets row sum + column sum (what the two passes with acc = 0, 1 left there: one launch instead of two)
// mods (rowsum == colsum only): the sum is multiplied by the molecule's gate mods[mol, g_off + f] — the gate's backward folded in (the gate
// is the same for every edge of the molecule, so it leaves the sum)
__global__ void k_edge_to_node(Topo t, int F, const float* __restrict__ x, float* rowsum, float* colsum, int acc,
                               const float* __restrict__ mods, int ldm, int g_off) {
    JT_IDX((long)t.Nn * F);
    const int node = (int)(i_ / F), f = (int)(i_ % F);
    const int b = t.node_mol[node], n = t.nn[b], i = node - t.node_off[b];
    const long e0 = t.edge_off[b];
    if (rowsum && rowsum == colsum) {
        float sr = 0.f, sc = 0.f;
#pragma unroll 8
        for (int c = 0; c < n; ++c) sr += x[(e0 + (long)i * n + c) * F + f];
#pragma unroll 8
        for (int a = 0; a < n; ++a) sc += x[(e0 + (long)a * n + i) * F + f];
        float v = sr + sc;
        if (mods) v *= mods[(long)b * ldm + g_off + f];
        rowsum[i_] = acc ? rowsum[i_] + v : v;
        return;
    }
    if (rowsum) {
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < n; ++c) s += x[(e0 + (long)i * n + c) * F + f];
        rowsum[i_] = acc ? rowsum[i_] + s : s;
    }
    if (colsum) {
        float s = 0.f;
#pragma unroll 8
        for (int a = 0; a < n; ++a) s += x[(e0 + (long)a * n + i) * F + f];
        colsum[i_] = acc ? colsum[i_] + s : s;
    }
}

// ================================================================ inputs ========================================================
__global__ void k_pack_nodes(Topo t, int nd, const float* __restrict__ xh, const float* __restrict__ cond_x, float* __restrict__ pos,
                             float* __restrict__ cpos, float* __restrict__ nin) {
    JT_IDX(t.Nn);
    const int b = t.node_mol[i_], i = (int)i_ - t.node_off[b];
    const int F = 3 + nd;
    const float* x = xh + ((long)b * t.N + i) * F;
    const float* c = cond_x ? cond_x + ((long)b * t.N + i) * F : nullptr;
    for (int d = 0; d < 3; ++d) { pos[i_ * 3 + d] = x[d]; cpos[i_ * 3 + d] = c ? c[d] : 0.f; }
    for (int j = 0; j < nd; ++j) { nin[i_ * 2 * nd + j] = x[3 + j]; nin[i_ * 2 * nd + nd + j] = c ? c[3 + j] : 0.f; }
}
// ein[r, 0:ch] = edge_x, [ch:2ch] = cond_edge_x (0 on the first step); adjacency flags and self-conditioning distances
// (mol_gnn.py:516-545: cond_adj_2d = cond_edge_x[..., 0] >= edge_th, or 1 without self-conditioning; adj_spatial = d^2 <= cut-off)
__global__ void k_pack_edges(Topo t, int ch, int ldin, float edge_th, float cutoff, const float* __restrict__ edge_x, const float* __restrict__ cond_edge_x,
                             const float* __restrict__ cpos, float* __restrict__ ein, float* __restrict__ adj2d, float* __restrict__ adjsp,
                             float* __restrict__ d2c, int* __restrict__ flags) {
    JT_IDX(t.R);
    const int b = t.edge_mol[i_], n = t.nn[b];
    const int loc = (int)(i_ - t.edge_off[b]), a = loc / n, c = loc % n;
    const long src = (((long)b * t.N + a) * t.N + c) * ch;
    for (int k = 0; k < ch; ++k) { ein[i_ * ldin + k] = edge_x[src + k]; ein[i_ * ldin + ch + k] = cond_edge_x ? cond_edge_x[src + k] : 0.f; }
    adj2d[i_] = cond_edge_x ? (cond_edge_x[src] >= edge_th ? 1.f : 0.f) : 1.f;
    const float* pa = cpos + (long)t.edge_a[i_] * 3; const float* pc = cpos + (long)t.edge_c[i_] * 3;
    const float dx = pa[0] - pc[0], dy = pa[1] - pc[1], dz = pa[2] - pc[2];
    const float d2 = dx * dx + dy * dy + dz * dz;
    d2c[i_] = d2;
    adjsp[i_] = d2 <= cutoff ? 1.f : 0.f;
    if (d2 != 0.f) flags[0] = 1;              // the batch-global first-step switch of mol_gnn.py:544 (same value from every writer)
}
__global__ void k_dist2(Topo t, const float* __restrict__ pos, float* __restrict__ d2) {
    JT_IDX(t.R);
    const float* pa = pos + (long)t.edge_a[i_] * 3; const float* pc = pos + (long)t.edge_c[i_] * 3;
    const float dx = pa[0] - pc[0], dy = pa[1] - pc[1], dz = pa[2] - pc[2];
    d2[i_] = dx * dx + dy * dy + dz * dz;
}
// time features [x, sin(2 pi x w), cos(2 pi x w)]  (LearnedSinusodialposEmb, layers.py:283-288; product order x * w * 2 * pi)
__global__ void k_time_feat(int B, int half, const float* __restrict__ nl, const float* __restrict__ w, float* __restrict__ feat) {
    JT_IDX((long)B * (2 * half + 1));
    const int F = 2 * half + 1, b = (int)(i_ / F), j = (int)(i_ % F);
    const float x = nl[b];
    if (j == 0) { feat[i_] = x; return; }
    const int k = (j - 1) % half;
    const float fr = x * w[k] * 2.f * 3.14159265358979323846f;
    feat[i_] = j <= half ? sinf(fr) : cosf(fr);
}
// dw[k] = sum_b x_b 2 pi (dsin cos(fr) - dcos sin(fr))
__global__ void k_time_feat_bwd(int B, int half, const float* __restrict__ nl, const float* __restrict__ w, const float* __restrict__ dfeat,
                                float* __restrict__ dw) {
    JT_IDX(half);
    const int F = 2 * half + 1;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float x = nl[b];
        const float fr = x * w[i_] * 2.f * 3.14159265358979323846f;
        s += x * 2.f * 3.14159265358979323846f * (dfeat[(long)b * F + 1 + i_] * cosf(fr) - dfeat[(long)b * F + 1 + half + i_] * sinf(fr));
    }
    dw[i_] += s;
}

// ================================================================ Gaussian basis (CondGaussianLayer, layers.py:328-334) =========
// x' = d2 (1 + scale) + shift with (scale, shift) = gm[mol, 0:2];  out[r, 0] = x', out[r, 1 + k] = N(x'; mean_k, |std_k| + 1e-5), pi = 3.14159
// zero_if: when given and *zero_if == 0 the output is all zeros (first step, mol_gnn.py:544-545)
__global__ void k_gbf_fwd(long rows, int De, const float* __restrict__ d2, const int* __restrict__ row_mol, const float* __restrict__ gm,
                          const float* __restrict__ means, const float* __restrict__ stds, const int* __restrict__ zero_if,
                          float* __restrict__ out, int ldo, int ocol) {
    JT_IDX(rows * De);
    const long r = i_ / De; const int j = (int)(i_ % De);
    float v = 0.f;
    if (!zero_if || *zero_if != 0) {
        const float* g = gm + (long)row_mol[r] * 2;
        const float x = d2[r] * (g[0] + 1.f) + g[1];
        if (j == 0) v = x;
        else {
            const float sd = fabsf(stds[j - 1]) + 1e-5f;
            const float z = (x - means[j - 1]) / sd;
            v = expf(-0.5f * (z * z)) / (2.5066272f * sd);            // (2 * 3.14159) ** 0.5 = 2.50662720...
        }
    }
    out[r * ldo + ocol + j] = v;
}
// per row: dx' = dG[r, 0] + sum_k dG[r, 1 + k] g_k (-(x' - m_k) / s_k^2);  dd2 (+)= dx' (1 + scale); dxp[r] = dx' (for the per-molecule sums)
__global__ void k_gbf_bwd_row(long rows, int De, const float* __restrict__ d2, const int* __restrict__ row_mol, const float* __restrict__ gm,
                              const float* __restrict__ means, const float* __restrict__ stds, const float* __restrict__ dG, int ldg, int gcol,
                              float* __restrict__ dxp, float* dd2, int acc) {
    JT_IDX(rows);
    const float* g = gm + (long)row_mol[i_] * 2;
    const float x = d2[i_] * (g[0] + 1.f) + g[1];
    const float* d = dG + i_ * ldg + gcol;
    double sum = (double)d[0];
    for (int k = 0; k < De - 1; ++k) {
        const float sd = fabsf(stds[k]) + 1e-5f;
        const float z = (x - means[k]) / sd;
        const float gk = expf(-0.5f * (z * z)) / (2.5066272f * sd);
        sum += (double)(d[1 + k] * gk * (-z / sd));
    }
    const float s = (float)sum;
    dxp[i_] = s;
    if (dd2) dd2[i_] = (acc ? dd2[i_] : 0.f) + s * (g[0] + 1.f);
}
// parameter gradients, stage 1 over row chunks: part_m[c, k] (means), part_s[c, k] (stds)
__global__ void k_gbf_bwd_par(long rows, int De, int chunk, const float* __restrict__ d2, const int* __restrict__ row_mol, const float* __restrict__ gm,
                              const float* __restrict__ means, const float* __restrict__ stds, const float* __restrict__ dG, int ldg, int gcol,
                              float* __restrict__ part_m, float* __restrict__ part_s) {
    const int K = De - 1;
    const long nchunks = (rows + chunk - 1) / chunk;
    JT_IDX(nchunks * K);
    const long c = i_ / K; const int k = (int)(i_ % K);
    const long r1 = (c + 1) * chunk < rows ? (c + 1) * chunk : rows;
    const float w = stds[k];
    const float sd = fabsf(w) + 1e-5f, sg = w < 0.f ? -1.f : (w > 0.f ? 1.f : 0.f);
    double dm = 0.0, ds = 0.0;
#pragma unroll 8
    for (long r = c * chunk; r < r1; ++r) {
        const float* g = gm + (long)row_mol[r] * 2;
        const float x = d2[r] * (g[0] + 1.f) + g[1];
        const float z = (x - means[k]) / sd;
        const float gk = expf(-0.5f * (z * z)) / (2.5066272f * sd);
        const float d = dG[r * ldg + gcol + 1 + k] * gk;
        dm += (double)(d * (z / sd));
        ds += (double)(d * ((z * z - 1.f) / sd) * sg);
    }
    part_m[i_] = (float)dm;
    part_s[i_] = (float)ds;
}

// ================================================================ attention (TransMixLayer, layers.py:131-186) =================
// S[(a, c), hd]: hd < XH: adjacency heads (1 or -1e10, :170-174); else sum_sc q[c] k[a] tanh(lin_edge0 et)[(a, c)] / sqrt(C)   (:165-167)
__global__ void k_attn_scores(Topo t, int H, int XH, int SC, float inv_sqrt_c, const float* __restrict__ q, const float* __restrict__ k,
                              const float* __restrict__ t0, const float* __restrict__ adj2d, const float* __restrict__ adjsp, float* __restrict__ S) {
    JT_IDX((long)t.R * H);
    const long r = i_ / H; const int hd = (int)(i_ % H);
    const int QK = (H - XH) * SC;
    float s;
    if (hd < XH) s = ((hd == 0 ? adj2d[r] : adjsp[r]) > 0.f) ? 1.f : -1e10f;
    else {
        const float* qq = q + (long)t.edge_c[r] * QK + (hd - XH) * SC;
        const float* kk = k + (long)t.edge_a[r] * QK + (hd - XH) * SC;
        const float* tt = t0 + r * QK + (hd - XH) * SC;
        float acc = 0.f;
#pragma unroll 8
        for (int j = 0; j < SC; ++j) acc += qq[j] * kk[j] * tt[j];
        s = acc * inv_sqrt_c;
    }
    S[i_] = s;
}
// softmax over the sources a of every (target c, head), a != c; alpha[(c, c)] = 0; in place   (torch_geometric softmax: / (sum + 1e-16))
__global__ void k_attn_softmax(Topo t, int H, float* __restrict__ S) {
    JT_IDX((long)t.Nn * H);
    const int node = (int)(i_ / H), hd = (int)(i_ % H);
    const int b = t.node_mol[node], n = t.nn[b], c = node - t.node_off[b];
    const long e0 = t.edge_off[b];
    float m = -INFINITY;
#pragma unroll 8
    for (int a = 0; a < n; ++a) if (a != c) m = fmaxf(m, S[(e0 + (long)a * n + c) * H + hd]);
    float sum = 0.f;
#pragma unroll 8
    for (int a = 0; a < n; ++a) if (a != c) sum += expf(S[(e0 + (long)a * n + c) * H + hd] - m);
#pragma unroll 8
    for (int a = 0; a < n; ++a) {
        float* p = S + (e0 + (long)a * n + c) * H + hd;
        *p = a == c ? 0.f : expf(*p - m) / (sum + 1e-16f);
    }
}
// hhat[c, f] = sum_a v[a, f] tanh(lin_edge1 et)[(a, c), f] dropout(alpha)[(a, c), f / C]
__global__ void k_attn_msg(Topo t, int D, int H, const float* __restrict__ v, const float* __restrict__ t1, const float* __restrict__ alpha,
                           Drop dr, float* __restrict__ hhat) {
    JT_IDX((long)t.Nn * D);
    const int node = (int)(i_ / D), f = (int)(i_ % D);
    const int b = t.node_mol[node], n = t.nn[b], c = node - t.node_off[b];
    const long e0 = t.edge_off[b];
    const int hd = f / (D / H);
    float s = 0.f;
#pragma unroll 8
    for (int a = 0; a < n; ++a) {
        const long r = e0 + (long)a * n + c;
        s += v[((long)t.node_off[b] + a) * D + f] * t1[r * D + f] * (alpha[r * H + hd] * drop_mul(dr, (unsigned long long)(r * H + hd)));
    }
    hhat[i_] = s;
}
// dv[a, f] = sum_c dhhat[c, f] t1[(a, c), f] ad[(a, c), hd]
__global__ void k_attn_bwd_v(Topo t, int D, int H, const float* __restrict__ dhhat, const float* __restrict__ t1, const float* __restrict__ alpha,
                             Drop dr, float* __restrict__ dv) {
    JT_IDX((long)t.Nn * D);
    const int node = (int)(i_ / D), f = (int)(i_ % D);
    const int b = t.node_mol[node], n = t.nn[b], a = node - t.node_off[b];
    const long e0 = t.edge_off[b];
    const int hd = f / (D / H);
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < n; ++c) {
        const long r = e0 + (long)a * n + c;
        s += dhhat[((long)t.node_off[b] + c) * D + f] * t1[r * D + f] * (alpha[r * H + hd] * drop_mul(dr, (unsigned long long)(r * H + hd)));
    }
    dv[i_] = s;
}
// dt1[(a, c), f] = dhhat[c, f] v[a, f] ad[(a, c), hd] (1 - t1^2)     (pre-activation gradient of lin_edge1)
__global__ void k_attn_bwd_t1(Topo t, int D, int H, const float* __restrict__ dhhat, const float* __restrict__ v, const float* __restrict__ t1,
                              const float* __restrict__ alpha, Drop dr, float* __restrict__ dt1) {
    JT_IDX((long)t.R * D);
    const long r = i_ / D; const int f = (int)(i_ % D);
    const int hd = f / (D / H);
    const float tv = t1[i_];
    dt1[i_] = dhhat[(long)t.edge_c[r] * D + f] * v[(long)t.edge_a[r] * D + f] * (alpha[r * H + hd] * drop_mul(dr, (unsigned long long)(r * H + hd))) * (1.f - tv * tv);
}
// dalpha[(a, c), hd] = dropout * sum_{f in head} dhhat[c, f] v[a, f] t1[(a, c), f]
__global__ void k_attn_bwd_alpha(Topo t, int D, int H, const float* __restrict__ dhhat, const float* __restrict__ v, const float* __restrict__ t1,
                                 Drop dr, float* __restrict__ dalpha) {
    JT_IDX((long)t.R * H);
    const long r = i_ / H; const int hd = (int)(i_ % H);
    const int C = D / H;
    const float* dh = dhhat + (long)t.edge_c[r] * D + hd * C;
    const float* vv = v + (long)t.edge_a[r] * D + hd * C;
    const float* tt = t1 + r * D + hd * C;
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < C; ++j) s += dh[j] * vv[j] * tt[j];
    dalpha[i_] = s * drop_mul(dr, (unsigned long long)i_);
}
// softmax backward over the sources of (c, hd): dS = alpha (dalpha - sum_a alpha dalpha); in place on dalpha
__global__ void k_attn_bwd_softmax(Topo t, int H, const float* __restrict__ alpha, float* __restrict__ dalpha) {
    JT_IDX((long)t.Nn * H);
    const int node = (int)(i_ / H), hd = (int)(i_ % H);
    const int b = t.node_mol[node], n = t.nn[b], c = node - t.node_off[b];
    const long e0 = t.edge_off[b];
    float dot = 0.f;
#pragma unroll 8
    for (int a = 0; a < n; ++a) { const long x = (e0 + (long)a * n + c) * H + hd; dot += alpha[x] * dalpha[x]; }
#pragma unroll 8
    for (int a = 0; a < n; ++a) { const long x = (e0 + (long)a * n + c) * H + hd; dalpha[x] = alpha[x] * (dalpha[x] - dot); }
}
// dq[c, j] = sum_a dS[(a, c), hd] k[a, j] t0[(a, c), j] / sqrt(C);  dk[a, j] = sum_c dS[(a, c), hd] q[c, j] t0[(a, c), j] / sqrt(C)
__global__ void k_attn_bwd_qk(Topo t, int H, int XH, int SC, float inv_sqrt_c, const float* __restrict__ dS, const float* __restrict__ q,
                              const float* __restrict__ k, const float* __restrict__ t0, float* __restrict__ dq, float* __restrict__ dk) {
    const int QK = (H - XH) * SC;
    JT_IDX((long)t.Nn * QK);
    const int node = (int)(i_ / QK), j = (int)(i_ % QK);
    const int b = t.node_mol[node], n = t.nn[b], i = node - t.node_off[b];
    const long e0 = t.edge_off[b], n0 = t.node_off[b];
    const int hd = XH + j / SC;
    float sq = 0.f, sk = 0.f;
#pragma unroll 8
    for (int o = 0; o < n; ++o) {
        const long rq = e0 + (long)o * n + i;          // (a = o, c = i)
        sq += dS[rq * H + hd] * k[(n0 + o) * QK + j] * t0[rq * QK + j];
        const long rk = e0 + (long)i * n + o;          // (a = i, c = o)
        sk += dS[rk * H + hd] * q[(n0 + o) * QK + j] * t0[rk * QK + j];
    }
    dq[i_] = sq * inv_sqrt_c;
    dk[i_] = sk * inv_sqrt_c;
}
// dt0[(a, c), j] = dS q[c, j] k[a, j] / sqrt(C) (1 - t0^2)          (pre-activation gradient of lin_edge0)
__global__ void k_attn_bwd_t0(Topo t, int H, int XH, int SC, float inv_sqrt_c, const float* __restrict__ dS, const float* __restrict__ q,
                              const float* __restrict__ k, const float* __restrict__ t0, float* __restrict__ dt0) {
    const int QK = (H - XH) * SC;
    JT_IDX((long)t.R * QK);
    const long r = i_ / QK; const int j = (int)(i_ % QK);
    const float tv = t0[i_];
    dt0[i_] = dS[r * H + XH + j / SC] * q[(long)t.edge_c[r] * QK + j] * k[(long)t.edge_a[r] * QK + j] * inv_sqrt_c * (1.f - tv * tv);
}

// ================================================================ coordinate update (MultiCondEquiUpdate, mol_gnn.py:85-92) =====
// trans[(a, c)] = CoorsNorm(x_a - x_c) * mean(inv * [1, adj2d, adjsp])    (0 on the diagonal)
__global__ void k_coord_fwd(Topo t, const float* __restrict__ pos, const float* __restrict__ inv, const float* __restrict__ adj2d,
                            const float* __restrict__ adjsp, const float* __restrict__ scale, float* __restrict__ trans) {
    JT_IDX(t.R);
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (t.edge_a[i_] != t.edge_c[i_]) {
        const float* pa = pos + (long)t.edge_a[i_] * 3; const float* pc = pos + (long)t.edge_c[i_] * 3;
        const float dx = pa[0] - pc[0], dy = pa[1] - pc[1], dz = pa[2] - pc[2];
        const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-8f);
        const float iota = (inv[i_ * 3] + inv[i_ * 3 + 1] * adj2d[i_] + inv[i_ * 3 + 2] * adjsp[i_]) / 3.f;
        const float s = scale[0];
        o0 = dx / nrm * s * iota; o1 = dy / nrm * s * iota; o2 = dz / nrm * s * iota;
    }
    trans[i_ * 3] = o0; trans[i_ * 3 + 1] = o1; trans[i_ * 3 + 2] = o2;
}
// x_pre[a] = x[a] + sum_c trans[(a, c)]
__global__ void k_coord_sum(Topo t, const float* __restrict__ pos, const float* __restrict__ trans, float* __restrict__ out) {
    JT_IDX((long)t.Nn * 3);
    const int node = (int)(i_ / 3), d = (int)(i_ % 3);
    const int b = t.node_mol[node], n = t.nn[b], a = node - t.node_off[b];
    const long e0 = t.edge_off[b];
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < n; ++c) s += trans[(e0 + (long)a * n + c) * 3 + d];
    out[i_] = pos[i_] + s;
}
// per molecule: x -= mean(x)   (remove_mean_with_mask, models/utils.py:38-45); zero_if: *zero_if != 0 -> zeros (NaN guard, mol_gnn.py:587-589)
__global__ void k_center(Topo t, const float* __restrict__ x, const int* __restrict__ zero_if, float* __restrict__ out) {
    JT_IDX((long)t.B * 3);
    const int b = (int)(i_ / 3), d = (int)(i_ % 3);
    const int n0 = t.node_off[b], n = t.nn[b];
    if (zero_if && *zero_if != 0) { for (int i = 0; i < n; ++i) out[(long)(n0 + i) * 3 + d] = 0.f; return; }
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < n; ++i) s += x[(long)(n0 + i) * 3 + d];
    const float m = s / (float)n;
    for (int i = 0; i < n; ++i) out[(long)(n0 + i) * 3 + d] = x[(long)(n0 + i) * 3 + d] - m;
}
__global__ void k_nan_flag(long n, const float* __restrict__ x, int* __restrict__ flag) {
    JT_IDX(n);
    if (x[i_] != x[i_]) flag[0] = 1;
}
// backward of the coordinate update for one edge row, given dxp = d loss / d x_pre (already centred):
//   dtrans = dxp[a];  dinv[k] = s (u . dtrans) adj_k / 3;  part of ddiff through CoorsNorm;  ds partial = iota (u . dtrans)
__global__ void k_coord_bwd(Topo t, const float* __restrict__ pos, const float* __restrict__ inv, const float* __restrict__ adj2d,
                            const float* __restrict__ adjsp, const float* __restrict__ scale, const float* __restrict__ dxp,
                            float* __restrict__ dinv, float* __restrict__ ddiff, float* __restrict__ dscale_row) {
    JT_IDX(t.R);
    float di0 = 0.f, di1 = 0.f, di2 = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, ds = 0.f;
    if (t.edge_a[i_] != t.edge_c[i_]) {
        const float* pa = pos + (long)t.edge_a[i_] * 3; const float* pc = pos + (long)t.edge_c[i_] * 3;
        const float dx = pa[0] - pc[0], dy = pa[1] - pc[1], dz = pa[2] - pc[2];
        const float raw = sqrtf(dx * dx + dy * dy + dz * dz);
        const float nrm = fmaxf(raw, 1e-8f);
        const float ux = dx / nrm, uy = dy / nrm, uz = dz / nrm;
        const float iota = (inv[i_ * 3] + inv[i_ * 3 + 1] * adj2d[i_] + inv[i_ * 3 + 2] * adjsp[i_]) / 3.f;
        const float s = scale[0];
        const float* dt = dxp + (long)t.edge_a[i_] * 3;
        const float ud = ux * dt[0] + uy * dt[1] + uz * dt[2];
        const float di = s * ud / 3.f;
        di0 = di; di1 = di * adj2d[i_]; di2 = di * adjsp[i_];
        ds = iota * ud;
        const float w = s * iota / nrm;
        if (raw > 1e-8f) { g0 = w * (dt[0] - ux * ud); g1 = w * (dt[1] - uy * ud); g2 = w * (dt[2] - uz * ud); }
        else { g0 = w * dt[0]; g1 = w * dt[1]; g2 = w * dt[2]; }
    }
    dinv[i_ * 3] = di0; dinv[i_ * 3 + 1] = di1; dinv[i_ * 3 + 2] = di2;
    ddiff[i_ * 3] = g0; ddiff[i_ * 3 + 1] = g1; ddiff[i_ * 3 + 2] = g2;
    dscale_row[i_] = ds;
}
// ddiff[(a, c)] += 2 (x_a - x_c) dd2[(a, c)]
__global__ void k_dist2_bwd(Topo t, const float* __restrict__ pos, const float* __restrict__ dd2, float* __restrict__ ddiff) {
    JT_IDX((long)t.R * 3);
    const long r = i_ / 3; const int d = (int)(i_ % 3);
    ddiff[i_] += 2.f * (pos[(long)t.edge_a[r] * 3 + d] - pos[(long)t.edge_c[r] * 3 + d]) * dd2[r];
}
// dx[a] (+)= base[a] + sum_c ddiff[(a, c)] - sum_c ddiff[(c, a)]
__global__ void k_diff_to_node(Topo t, const float* __restrict__ ddiff, const float* __restrict__ base, float* __restrict__ dx) {
    JT_IDX((long)t.Nn * 3);
    const int node = (int)(i_ / 3), d = (int)(i_ % 3);
    const int b = t.node_mol[node], n = t.nn[b], a = node - t.node_off[b];
    const long e0 = t.edge_off[b];
    float s = base ? base[i_] : 0.f;
#pragma unroll 8
    for (int c = 0; c < n; ++c) s += ddiff[(e0 + (long)a * n + c) * 3 + d] - ddiff[(e0 + (long)c * n + a) * 3 + d];
    dx[i_] = s;
}

// ================================================================ outputs =======================================================
// out_edge[b, a, c, k] = 0.5 (Ep[(a, c), k] + Ep[(c, a), k]) off the diagonal inside the molecule, else 0      (mol_gnn.py:578-582)
__global__ void k_edge_out(Topo t, int ch, const float* __restrict__ Ep, float* __restrict__ out) {
    JT_IDX((long)t.B * t.N * t.N * ch);
    const int k = (int)(i_ % ch);
    long x = i_ / ch;
    const int c = (int)(x % t.N); x /= t.N;
    const int a = (int)(x % t.N); const int b = (int)(x / t.N);
    const int n = t.nn[b];
    float v = 0.f;
    if (a < n && c < n && a != c) {
        const long e0 = t.edge_off[b];
        v = 0.5f * (Ep[(e0 + (long)a * n + c) * ch + k] + Ep[(e0 + (long)c * n + a) * ch + k]);
    }
    out[i_] = v;
}
__global__ void k_edge_out_bwd(Topo t, int ch, const float* __restrict__ dout, float* __restrict__ dEp) {
    JT_IDX((long)t.R * ch);
    const long r = i_ / ch; const int k = (int)(i_ % ch);
    const int b = t.edge_mol[r], n = t.nn[b];
    const int loc = (int)(r - t.edge_off[b]), a = loc / n, c = loc % n;
    float v = 0.f;
    if (a != c) v = 0.5f * (dout[((((long)b * t.N + a) * t.N + c) * ch) + k] + dout[((((long)b * t.N + c) * t.N + a) * ch) + k]);
    dEp[i_] = v;
}
// out_xh[b, i, :] = [pos_final, atom] inside the molecule, 0 on padding
__global__ void k_node_out(Topo t, int nd, const float* __restrict__ posf, const float* __restrict__ atom, float* __restrict__ out) {
    const int F = 3 + nd;
    JT_IDX((long)t.B * t.N * F);
    const int j = (int)(i_ % F);
    long x = i_ / F;
    const int i = (int)(x % t.N), b = (int)(x / t.N);
    float v = 0.f;
    if (i < t.nn[b]) { const long r = t.node_off[b] + i; v = j < 3 ? posf[r * 3 + j] : atom[r * nd + (j - 3)]; }
    out[i_] = v;
}
__global__ void k_node_out_bwd(Topo t, int nd, const float* __restrict__ dout, float* __restrict__ dposf, float* __restrict__ datom) {
    const int F = 3 + nd;
    JT_IDX((long)t.Nn * F);
    const long r = i_ / F; const int j = (int)(i_ % F);
    const int b = t.node_mol[r], i = (int)r - t.node_off[b];
    const float v = dout[((long)b * t.N + i) * F + j];
    if (j < 3) dposf[r * 3 + j] = v; else datom[r * nd + (j - 3)] = v;
}

}  // namespace jt
