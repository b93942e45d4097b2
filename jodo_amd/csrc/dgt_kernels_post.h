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
        if (A.layer > 0) {                                    // at least one block ran
            if (A.flags[FLAG_ASYM]) {
                const int parts = A.pd.strip_parts[v >> 5];
                for (int q = 0; q < parts; ++q) {
                    const float4 dp = reinterpret_cast<const float4*>(A.dpos)[(size_t)v * A.pd.max_parts + q];
                    p.x += dp.x; p.y += dp.y; p.z += dp.z;
                }
            } else {
                const int n = A.pd.node_n[v], i = A.pd.node_i[v];
                const float4* row = reinterpret_cast<const float4*>(A.dposE) + (size_t)A.pd.node_eoff[v] + (size_t)i * n;
                // eight, then four rows in flight per step, added in column order (the sum is bit-identical to the one-load-per-iteration
                // loop, which paid one exposed L2 round trip per neighbour: 15 us per launch at QM9 B = 2500, nine launches per forward;
                // GEOM molecules have up to 181 atoms: 23 us per launch at four in flight)
                int c = 0;
                for (; c + 8 <= n; c += 8) {
                    float4 d[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) d[u] = row[c + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (c + u != i) { p.x += d[u].x; p.y += d[u].y; p.z += d[u].z; }
                }
                for (; c + 4 <= n; c += 4) {
                    const float4 d0 = row[c], d1 = row[c + 1], d2 = row[c + 2], d3 = row[c + 3];
                    if (c != i) { p.x += d0.x; p.y += d0.y; p.z += d0.z; }
                    if (c + 1 != i) { p.x += d1.x; p.y += d1.y; p.z += d1.z; }
                    if (c + 2 != i) { p.x += d2.x; p.y += d2.y; p.z += d2.z; }
                    if (c + 3 != i) { p.x += d3.x; p.y += d3.y; p.z += d3.z; }
                }
                for (; c < n; ++c) {
                    if (c == i) continue;
                    const float4 dp = row[c];
                    p.x += dp.x; p.y += dp.y; p.z += dp.z;
                }
            }
        }
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
    float mx = 0.f, my = 0.f, mz = 0.f;
    for (int k = 0; k < n; ++k) {
        const float4 p = reinterpret_cast<const float4*>(A.pos_out)[v0 + k];
        mx += p.x; my += p.y; mz += p.z;
    }
    const float4 p = reinterpret_cast<const float4*>(A.pos_out)[v0 + i];
    const float inv = 1.f / (float)n;
    o[0] = nan ? 0.f : p.x - mx * inv;
    o[1] = nan ? 0.f : p.y - my * inv;
    o[2] = nan ? 0.f : p.z - mz * inv;
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
