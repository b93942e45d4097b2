// Epilogue kernels (prediction heads: dgt_kernels_wide.h): NaN guard, centre-of-mass removal, dense symmetrised output.
// Reference: DGT_concat.forward models/mol_gnn.py:571-594, to_dense_edge_attr models/utils.py:129-137,
// remove_mean_with_mask models/utils.py:38-45.
#pragma once
#include "dgt_kernels_common.h"

namespace jd {

// final positions (before centring): last block's input positions + its partial updates; NaN guard flag
__global__ void k_pos_final(KArgs A) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= A.pd.Nn_pad) return;
    float4 p = reinterpret_cast<const float4*>(A.pos_in)[v];
    if (v < A.pd.Nn) {
        if (A.layer > 0)                                      // at least one block ran
            p = advance_position(A, p, v, v >> 5, A.pd.node_n[v], A.pd.node_i[v], A.pd.node_eoff[v]);
        if (isnan(p.x) || isnan(p.y) || isnan(p.z)) atomicOr(&A.flags[FLAG_NAN], 1);
    }
    reinterpret_cast<float4*>(A.pos_out)[v] = p;
}

// dense outputs: out_xh [B,N,3+nd], out_edge [B,N,N,ch]; zeros on padding; edges symmetrised
__global__ void k_finalize_nodes(KArgs A) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;    // (b, i)
    if (idx >= A.pd.B * A.pd.N) return;
    // sticky count of evaluations in which the NaN guard fired: never reset by the library, read-and-cleared by the
    // caller once per sampling round (the reference prints its warning per forward, mol_gnn.py:588)
    if (idx == 0 && A.flags[FLAG_NAN] != 0) A.flags[FLAG_NAN_COUNT] += 1;
    if (idx == 0) {                                           // a pinned path that this call's inputs did not take (sticky)
        int bad = 0;
        if (A.pin_sym == 1 && A.flags[FLAG_ASYM]) bad |= 1;       // (a directed pin forces FLAG_ASYM in k_flags_init: always valid)
        if ((A.pin_uni == 1 && !A.flags[FLAG_UNIFORM_T]) || (A.pin_uni == 2 && A.flags[FLAG_UNIFORM_T])) bad |= 2;
        if (bad) A.flags[FLAG_PIN_VIOLATED] |= bad;
    }
    const int b = idx / A.pd.N, i = idx % A.pd.N;
    const int n = A.pd.orig_n[b], nd = A.d.nd;
    float* o = A.out_xh + (size_t)idx * (3 + nd);
    if (i >= n) {
        for (int f = 0; f < 3 + nd; ++f) o[f] = 0.f;
        return;
    }
    const int v0 = A.pd.orig_noff[b];
    const bool nan = A.flags[FLAG_NAN] != 0;
    // centre of mass (remove_mean_with_mask, models/utils.py:38-45): the sum over up to 181 positions of size ~ 4 runs in double —
    // in float its partial sums reach 10^2 and every addition rounds at their ulp, not at the positions'
    double mx = 0., my = 0., mz = 0.;
    for (int k = 0; k < n; ++k) {
        const float4 p = reinterpret_cast<const float4*>(A.pos_out)[v0 + k];
        mx += p.x; my += p.y; mz += p.z;
    }
    const float4 p = reinterpret_cast<const float4*>(A.pos_out)[v0 + i];
    const double inv = 1.0 / (double)n;
    o[0] = nan ? 0.f : p.x - (float)(mx * inv);
    o[1] = nan ? 0.f : p.y - (float)(my * inv);
    o[2] = nan ? 0.f : p.z - (float)(mz * inv);
    const float* ap = A.apred + (size_t)(v0 + i) * 32;
    for (int f = 0; f < nd; ++f) o[3 + f] = ap[f];
}

__global__ void k_finalize_edges(KArgs A) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (b, a, c)
    const size_t NN = (size_t)A.pd.N * A.pd.N;
    if (idx >= (size_t)A.pd.B * NN) return;
    const int b = (int)(idx / NN);
    const int a = (int)((idx % NN) / A.pd.N), c = (int)(idx % A.pd.N);
    const int n = A.pd.orig_n[b], ch = A.d.ch;
    float* o = A.out_edge + idx * ch;
    if (a >= n || c >= n || a == c) {
        for (int f = 0; f < ch; ++f) o[f] = 0.f;
        return;
    }
    const size_t e0 = (size_t)A.pd.orig_eoff[b];
    const float4 p = reinterpret_cast<const float4*>(A.epred)[e0 + (size_t)a * n + c];
    const float4 q = reinterpret_cast<const float4*>(A.epred)[e0 + (size_t)c * n + a];
    const float pp[4] = {p.x, p.y, p.z, p.w}, qq[4] = {q.x, q.y, q.z, q.w};
    for (int f = 0; f < ch; ++f) o[f] = 0.5f * (pp[f] + qq[f]);
}

}  // namespace jd
