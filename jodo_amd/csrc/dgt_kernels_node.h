// Node-side kernels of a DGT block (EquivariantMixBlock.forward, models/mol_gnn.py:270-322):
//   k_node_pre    LN1 + modulate + q / k / v projections                      (layers.py:147-149)
//   k_node_post1  node2edge_lin per node, gated residual + LN2 + FFN (partial sums over hidden halves)
//   k_node_post2  FFN combine -> h', W_row h' / W_col h' (halves of equi_update.input_lin), readout
// There are only Nn/32 node strips (1409 at QM9 B = 2500 — fewer than two per SIMD), so each kernel is
// cut into independent (strip, piece) work items that redo the cheap LayerNorm prologue; the pieces
// exchange data through small global buffers (FFN partial sums, ping-pong node state).
#pragma once
#include "dgt_kernels_common.h"

namespace jd {

// ------------------------------------------------------------------------------------------------
// piece 0 / 1 / 2 = q / k / v; piece 0 also advances the positions
__global__ __launch_bounds__(64, 1) void k_node_pre(KArgs A) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int strip = blockIdx.x / 3, piece = blockIdx.x % 3;
    const LaneNode L = lane_node(A, strip, j);
    // positions entering this block: previous positions + the contributions of the previous update
    if (piece == 0) {
        float4 p = reinterpret_cast<const float4*>(A.pos_in)[L.v];
        if (A.layer > 0) {
            if (A.flags[FLAG_ASYM]) {
                const int parts = A.pd.strip_parts[strip];
                for (int q = 0; q < parts; ++q) {
                    const float4 dp = reinterpret_cast<const float4*>(A.dpos)[(size_t)L.v * A.pd.max_parts + q];
                    p.x += dp.x; p.y += dp.y; p.z += dp.z;
                }
            } else {                                       // pair path: one contribution per edge row (i, c)
                const float4* row = reinterpret_cast<const float4*>(A.dposE) + (size_t)L.eoff + (size_t)L.i * L.n;
                for (int c = 0; c < L.n; ++c) {
                    if (c == L.i) continue;
                    const float4 dp = row[c];
                    p.x += dp.x; p.y += dp.y; p.z += dp.z;
                }
            }
        }
        if (half == 0) reinterpret_cast<float4*>(A.pos_out)[L.v] = p;
    }
    const float* mr = mod_row(A, L.b) + A.mod_base;          // node chunks: ns1, nc1, ng1, ns2, nc2, ng2
    float hx[128];
    load_nat<8>(A.h + (size_t)L.v * 256, half, hx);
    layer_norm<128>(hx);
    modulate<8>(hx, mr, mr + 256, half);
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned woff = (unsigned)(A.wb[piece == 0 ? JB_WQ : (piece == 1 ? JB_WK : JB_WV)] * 4);
    const float* bias = A.W + A.wb[piece == 0 ? JB_BQ : (piece == 1 ? JB_BK : JB_BV)];
    float* outp = piece == 0 ? A.q : (piece == 1 ? A.k : A.v);
    WPipe<8> wp;
    wpipe_prime(wp, ws, woff);
#pragma unroll 1
    for (int b = 0; b < 8; ++b) {
        const unsigned cur = woff + (unsigned)b * 32 * 1024;
        const unsigned nxt = b < 7 ? cur + 32 * 1024 : woff;
        float bb[16], r[16];
        load16(bias + b * 32 + half * 16, bb);
        f32x16 acc = mfma_block_p<32>(wp, ws, cur, nxt, hx, zero16());
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = acc[s] + bb[s];
        store16(outp + (size_t)L.v * 256 + b * 32 + half * 16, r);
    }
}

// hh = aggregated attention messages (sum of the per-chunk partials), hx = LN(h + ng1 * hh) * (1 + nc2) + ns2
__device__ __forceinline__ void node_mid(const KArgs& A, const LaneNode& L, int strip, int half, const float* mr,
                                         float (&hh)[128], float (&hx)[128]) {
    const float* ng1 = mr + 2 * 256, *ns2 = mr + 3 * 256, *nc2 = mr + 4 * 256;
#pragma unroll
    for (int s = 0; s < 128; ++s) hh[s] = 0.f;
    const int parts = A.pd.strip_parts[strip];
    for (int q = 0; q < parts; ++q) {
        float tmp[128];
        load_nat<8>(A.hhat + ((size_t)L.v * A.pd.max_parts + q) * 256, half, tmp);
#pragma unroll
        for (int s = 0; s < 128; ++s) hh[s] += tmp[s];
    }
    load_nat<8>(A.h + (size_t)L.v * 256, half, hx);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        float g[16];
        load16(ng1 + b * 32 + half * 16, g);
#pragma unroll
        for (int s = 0; s < 16; ++s) hx[b * 16 + s] = fmaf(g[s], hh[b * 16 + s], hx[b * 16 + s]);
    }
    layer_norm<128>(hx);
    modulate<8>(hx, ns2, nc2, half);
}

// ------------------------------------------------------------------------------------------------
// piece p in {0,1}: node2edge_lin output block p; FFN hidden chunks [p*2R, (p+1)*2R) -> partial ff2 sums
template <int R>   // mlp_ratio
__global__ __launch_bounds__(64, 1) void k_node_post1(KArgs A) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int strip = blockIdx.x >> 1, piece = blockIdx.x & 1;
    const LaneNode L = lane_node(A, strip, j);
    const float* mr = mod_row(A, L.b) + A.mod_base;
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned oN2E = (unsigned)(A.wb[JB_N2E_W] * 4) + (unsigned)piece * 32 * 1024;
    const unsigned oF1 = (unsigned)(A.wb[JB_FF1_W] * 4), oF2 = (unsigned)(A.wb[JB_FF2_W] * 4);
    WPipe<8> wp;
    wpipe_prime(wp, ws, oN2E);
    float hh[128], hx[128];
    node_mid(A, L, strip, half, mr, hh, hx);
    const int c0 = piece * 2 * R, c1 = c0 + 2 * R;
    {   // node2edge_lin applied per node (bias added on the edge side)
        f32x16 acc = mfma_block_p<32>(wp, ws, oN2E, oF1 + (unsigned)(c0 * 2) * 32 * 1024, hh, zero16());
        float r[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = acc[s];
        store16(A.n2e + (size_t)L.v * 64 + piece * 32 + half * 16, r);
    }
    f32x16 o[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) o[b] = zero16();
    {
        const float* b1 = A.W + A.wb[JB_FF1_B];
        constexpr int KQ2 = R * 256 / 8;                      // quads per ff2 output block
#pragma unroll 1
        for (int c = c0; c < c1; ++c) {
            float hid[32];
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const unsigned cur = oF1 + (unsigned)(c * 2 + b2) * 32 * 1024;
                const unsigned nxt = b2 == 0 ? cur + 32 * 1024 : oF2 + (unsigned)(c * 8) * 1024;
                float bb[16];
                load16(b1 + (c * 2 + b2) * 32 + half * 16, bb);
                f32x16 acc = mfma_block_p<32>(wp, ws, cur, nxt, hx, zero16());
#pragma unroll
                for (int s = 0; s < 16; ++s) hid[b2 * 16 + s] = silu_f(acc[s] + bb[s]);
            }
#pragma unroll
            for (int ob = 0; ob < 8; ++ob) {
                const unsigned cur = oF2 + (unsigned)(ob * KQ2 + c * 8) * 1024;
                const unsigned nxt = ob < 7 ? oF2 + (unsigned)((ob + 1) * KQ2 + c * 8) * 1024
                                            : oF1 + (unsigned)((c + 1 < c1 ? c + 1 : c0) * 2) * 32 * 1024;
                o[ob] = mfma_block_p<8>(wp, ws, cur, nxt, hid, o[ob]);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        float r[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = o[b][s];
        store16(A.ffp + ((size_t)L.v * 2 + piece) * 256 + b * 32 + half * 16, r);
    }
}

// ------------------------------------------------------------------------------------------------
// piece 0: h' -> h_out, W_row h' + b ; piece 1: W_col h', readout node_l(h')
__global__ __launch_bounds__(64, 1) void k_node_post2(KArgs A) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int strip = blockIdx.x >> 1, piece = blockIdx.x & 1;
    const LaneNode L = lane_node(A, strip, j);
    const float* mr = mod_row(A, L.b) + A.mod_base;
    const float* ng2 = mr + 5 * 256;
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned oW = (unsigned)(A.wb[piece == 0 ? JB_ROW_W : JB_COL_W] * 4), oNro = (unsigned)(A.wb[JB_NRO_W] * 4);
    WPipe<8> wp;
    wpipe_prime(wp, ws, oW);
    float hx[128];
    {
        float hh[128];
        node_mid(A, L, strip, half, mr, hh, hx);
    }
    {   // h' = hx + ng2 * (ff2 partial 0 + partial 1 + bias)
        const float* b2 = A.W + A.wb[JB_FF2_B];
        const float* p0 = A.ffp + (size_t)L.v * 2 * 256, *p1 = p0 + 256;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float a0[16], a1[16], bb[16], g[16];
            load16(p0 + b * 32 + half * 16, a0);
            load16(p1 + b * 32 + half * 16, a1);
            load16(b2 + b * 32 + half * 16, bb);
            load16(ng2 + b * 32 + half * 16, g);
#pragma unroll
            for (int s = 0; s < 16; ++s) hx[b * 16 + s] = fmaf(g[s], a0[s] + a1[s] + bb[s], hx[b * 16 + s]);
        }
    }
    if (piece == 0) store_nat<8>(A.h_out + (size_t)L.v * 256, half, hx);
    const float* bin = A.W + A.wb[JB_IN_B];
    float* outp = piece == 0 ? A.wrow : A.wcol;
#pragma unroll 1
    for (int b = 0; b < 8; ++b) {
        const unsigned cur = oW + (unsigned)b * 32 * 1024;
        const unsigned nxt = b < 7 ? cur + 32 * 1024 : oNro;
        float bb[16], r[16];
        load16(bin + b * 32 + half * 16, bb);
        f32x16 acc = mfma_block_p<32>(wp, ws, cur, nxt, hx, zero16());
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = acc[s] + (piece == 0 ? bb[s] : 0.f);
        store16(outp + (size_t)L.v * 256 + b * 32 + half * 16, r);
    }
    if (piece == 1) {   // readout node_l(h') -> atom_hids[:, D + l*64 ...]
        const float* bias = A.W + A.wb[JB_NRO_B];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned cur = oNro + (unsigned)b * 32 * 1024;
            float bb[16], r[16];
            load16(bias + b * 32 + half * 16, bb);
            f32x16 acc = mfma_block_p<32>(wp, ws, cur, b == 0 ? cur + 32 * 1024 : oNro, hx, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) r[s] = acc[s] + bb[s];
            store16(A.ahid + (size_t)L.v * A.d.KNH + 256 + A.layer * 64 + b * 32 + half * 16, r);
        }
    }
}

}  // namespace jd
