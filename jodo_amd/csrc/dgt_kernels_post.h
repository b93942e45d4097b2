// Epilogue kernels: prediction heads, NaN guard, centre-of-mass removal, dense symmetrised output.
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
                for (int c = 0; c < n; ++c) {
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

// node head: [h0 ; node_0(h) ; ... ] (KNH) -> 256 -> SiLU -> 128 -> SiLU -> nd
__global__ __launch_bounds__(64, 1) void k_node_head(KArgs A) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int v = blockIdx.x * 32 + j;
    const int KNH = A.d.KNH;
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned o1 = (unsigned)(A.wg[JW_NH1_W] * 4), o2 = (unsigned)(A.wg[JW_NH2_W] * 4), o3 = (unsigned)(A.wg[JW_NH3_W] * 4);
    WPipe<8> wp;
    wpipe_prime(wp, ws, o1);
    f32x16 o[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) o[b] = zero16();
    {
        const int kq = KNH / 8, nch = KNH / 64;
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            float x[32];
            load_nat<2>(A.ahid + (size_t)v * KNH + c * 64, half, x);
#pragma unroll
            for (int ob = 0; ob < 8; ++ob) {
                const unsigned cur = o1 + (unsigned)(ob * kq + c * 8) * 1024;
                const unsigned nxt = ob < 7 ? o1 + (unsigned)((ob + 1) * kq + c * 8) * 1024
                                            : (c + 1 < nch ? o1 + (unsigned)((c + 1) * 8) * 1024 : o2);
                o[ob] = mfma_block_p<8>(wp, ws, cur, nxt, x, o[ob]);
            }
        }
    }
    float a1[128];
    {
        const float* bias = A.W + A.wg[JW_NH1_B];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float r[16];
            acc_bias(o[b], bias + b * 32 + half * 16, r);
#pragma unroll
            for (int s = 0; s < 16; ++s) a1[b * 16 + s] = silu_f(r[s]);
        }
    }
    float a2[64];
    {
        const float* bias = A.W + A.wg[JW_NH2_B];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned cur = o2 + (unsigned)(b * 32) * 1024;
            f32x16 acc = mfma_block_p<32>(wp, ws, cur, b < 3 ? cur + 32 * 1024 : o3, a1, zero16());
            float r[16];
            acc_bias(acc, bias + b * 32 + half * 16, r);
#pragma unroll
            for (int s = 0; s < 16; ++s) a2[b * 16 + s] = silu_f(r[s]);
        }
    }
    {
        f32x16 acc = mfma_block_p<16>(wp, ws, o3, o3, a2, zero16());
        float r[16];
        acc_bias(acc, A.W + A.wg[JW_NH3_B] + half * 16, r);
        store16(A.apred + (size_t)v * 32 + half * 16, r);
    }
}

// edge heads on dense rows: [e0 ; edge_0(e) ; ...] (KEH) -> [exist 64 | type 64] -> SiLU
//   -> block-diagonal [32 | 32] -> SiLU -> block-diagonal [1 | ch-1]
template <int NBK>     // KEH / 32
__global__ __launch_bounds__(64) void k_edge_head(KArgs A) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const size_t r = (size_t)blockIdx.x * 32 + j;             // workspace rows are padded by 32
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned o1 = (unsigned)(A.wg[JW_EH1_W] * 4), o2 = (unsigned)(A.wg[JW_EH2_W] * 4), o3 = (unsigned)(A.wg[JW_EH3_W] * 4);
    WPipe<4> wp;                                              // NBK * 4 quads per block: any NBK divides
    wpipe_prime(wp, ws, o1);
    float x[NBK * 16];
    load_nat<NBK>(A.ehid + r * (NBK * 32), half, x);
    float a1[64];
    {
        const float* bias = A.W + A.wg[JW_EH1_B];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned cur = o1 + (unsigned)(b * NBK * 4) * 1024;
            f32x16 acc = mfma_block_p<NBK * 4>(wp, ws, cur, b < 3 ? cur + (unsigned)(NBK * 4) * 1024 : o2, x, zero16());
            float rr[16];
            acc_bias(acc, bias + b * 32 + half * 16, rr);
#pragma unroll
            for (int s = 0; s < 16; ++s) a1[b * 16 + s] = silu_f(rr[s]);
        }
    }
    float a2[32];
    {
        const float* bias = A.W + A.wg[JW_EH2_B];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned cur = o2 + (unsigned)(b * 16) * 1024;
            f32x16 acc = mfma_block_p<16>(wp, ws, cur, b < 1 ? cur + 16 * 1024 : o3, a1, zero16());
            float rr[16];
            acc_bias(acc, bias + b * 32 + half * 16, rr);
#pragma unroll
            for (int s = 0; s < 16; ++s) a2[b * 16 + s] = silu_f(rr[s]);
        }
    }
    {
        f32x16 acc = mfma_block_p<8>(wp, ws, o3, o3, a2, zero16());
        float rr[16];
        acc_bias(acc, A.W + A.wg[JW_EH3_B] + half * 16, rr);
        if (half == 0 && r < (size_t)A.pd.rows)
            reinterpret_cast<float4*>(A.epred)[r] = make_float4(rr[0], rr[1], rr[2], rr[3]);
    }
}

// dense outputs: out_xh [B,N,3+nd], out_edge [B,N,N,ch]; zeros on padding; edges symmetrised
__global__ void k_finalize_nodes(KArgs A) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;    // (b, i)
    if (idx >= A.pd.B * A.pd.N) return;
    // sticky count of evaluations in which the NaN guard fired: never reset by the library, read-and-cleared by the
    // caller once per sampling round (the reference prints its warning per forward, mol_gnn.py:588)
    if (idx == 0 && A.flags[FLAG_NAN] != 0) A.flags[FLAG_NAN_COUNT] += 1;
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
