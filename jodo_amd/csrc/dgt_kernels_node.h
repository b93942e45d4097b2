// Node-side kernels of a DGT block (EquivariantMixBlock.forward, models/mol_gnn.py:270-322):
//   k_node_pre    LN1 + modulate + q / k / v projections                      (layers.py:147-149)
//   k_node_post   node2edge_lin per node, gated residual + LN2 + FFN, W_row h' / W_col h' (halves of
//                 equi_update.input_lin), readout
//   k_node_postw  the same with a workgroup of 2 or 4 waves per strip (few strips / remainder strips)
// There are only Nn/32 node strips (1409 at QM9 B = 2500 — fewer than two per SIMD); k_node_pre is cut
// into independent (strip, piece) items (q / k / v), which measured faster.  k_node_post keeps one item per strip
// for every full round of 1024 strips and hands the remainder to k_node_postw; with >= 1024 strips it also
// produces the next block's q / k / v (node_next_*), see the launcher in dgt_forward.hip.
#pragma once
#include "dgt_kernels_attn.h"

namespace jd {

// ------------------------------------------------------------------------------------------------
// piece 0 / 1 / 2 = q / k / v; piece 0 also advances the positions
// NEXT: the items of the FOLLOWING block (weights A.wbn, modulation row A.mod_base_next; h already holds that block's input),
// issued from the launch that carries this block's k_node_ab items (k_node_ab_pre, dgt_forward.hip); positions are not touched
template <bool NEXT = false>
__device__ __forceinline__ void node_pre_body(const KArgs& A, int blk) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int strip = blk / 3, piece = blk % 3;
    const LaneNode L = lane_node(A, strip, j);
    // positions entering this block: previous positions + the contributions of the previous update
    if (!NEXT && piece == 0 && A.pre_mode == 0) {
        float4 p = reinterpret_cast<const float4*>(A.pos_in)[L.v];
        if (A.layer > 0) p = advance_position(A, p, L.v, strip, L.n, L.i, L.eoff);
        if (half == 0) reinterpret_cast<float4*>(A.pos_out)[L.v] = p;
    }
    const float* mr = mod_row(A, L.b) + (NEXT ? A.mod_base_next : A.mod_base);          // node chunks: ns1, nc1, ng1, ns2, nc2, ng2
    float hx[128];
    load_nat<8>(A.h + (size_t)L.v * 256, half, hx);
    layer_norm<128>(hx);
    modulate<8>(hx, mr, mr + 256, half);
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned woff = (unsigned)((NEXT ? A.wbn[2 * piece] : A.wb[piece == 0 ? JB_WQ : (piece == 1 ? JB_WK : JB_WV)]) * 4);
    const float* bias = A.W + (NEXT ? A.wbn[2 * piece + 1] : A.wb[piece == 0 ? JB_BQ : (piece == 1 ? JB_BK : JB_BV)]);
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
        store16T(outp, 8, L.v, half, b, r);
    }
}
__global__ __launch_bounds__(64, 1) void k_node_pre(KArgs A) { node_pre_body<false>(A, (int)blockIdx.x); }
// hh = aggregated attention messages: flash-style merge of the per-item partials of the fused attention kernel
// (attn_merge, dgt_kernels_attn.h), fixed order
__device__ __forceinline__ void node_load_hh(const KArgs& A, const LaneNode& L, int half, float (&hh)[128]) {
    attn_merge<256>(A, L.v, half, hh);
}

// in place: x = LN(h + ng1 * x) * (1 + nc2) + ns2   (x holds hh on entry, the FFN input on exit)
__device__ __forceinline__ void node_residual_ln(const KArgs& A, const LaneNode& L, int half, const float* mr, float (&x)[128]) {
    const float* ng1 = mr + 2 * 256, *ns2 = mr + 3 * 256, *nc2 = mr + 4 * 256;
    const float* hrow = A.h + (size_t)L.v * 256;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        float g[16], h0[16];
        load16(ng1 + b * 32 + half * 16, g);
        load16(hrow + b * 32 + half * 16, h0);
#pragma unroll
        for (int s = 0; s < 16; ++s) x[b * 16 + s] = fmaf(g[s], x[b * 16 + s], h0[s]);
    }
    layer_norm<128>(x);
    modulate<8>(x, ns2, nc2, half);
}

// ---- the next block's k_node_pre work, fused behind k_node_post (which holds h' in registers) -------
// x: h' in, LN1(h') * (1 + nc1) + ns1 of the NEXT block out
__device__ __forceinline__ void node_next_ln(const KArgs& A, const LaneNode& L, int half, float (&x)[128]) {
    const float* mr = mod_row(A, L.b) + A.mod_base_next;
    layer_norm<128>(x);
    modulate<8>(x, mr, mr + 256, half);
}
// output blocks [g0, g1) of the concatenated projection q (0-7) | k (8-15) | v (16-23); the weight ring must
// already hold the first quads of block g0 (the caller chained its previous block to it)
__device__ __forceinline__ void node_next_qkv(const KArgs& A, const LaneNode& L, int half, const float (&x)[128], const WSrc& ws,
                                              WPipe<8>& wp, int g0, int g1) {
#pragma unroll 1
    for (int g = g0; g < g1; ++g) {
        const int piece = g >> 3, b = g & 7;
        const unsigned cur = (unsigned)(A.wbn[2 * piece] * 4) + (unsigned)b * 32 * 1024;
        const int gn = g + 1 < g1 ? g + 1 : g0;
        const unsigned nxt = (unsigned)(A.wbn[2 * (gn >> 3)] * 4) + (unsigned)(gn & 7) * 32 * 1024;
        const float* bias = A.W + A.wbn[2 * piece + 1];
        float* outp = piece == 0 ? A.q : (piece == 1 ? A.k : A.v);
        float bb[16], r[16];
        load16(bias + b * 32 + half * 16, bb);
        f32x16 acc = mfma_block_p<32>(wp, ws, cur, nxt, x, zero16());
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = acc[s] + bb[s];
        store16T(outp, 8, L.v, half, b, r);
    }
}

// ------------------------------------------------------------------------------------------------
// One item per strip.  (Splitting this kernel into (strip, piece) items was measured slower: every item
// pays the same latency-bound prologue — partial-sum reduction, LayerNorm — which is 20-50 % of a piece.)
template <int R>   // mlp_ratio
__global__ __launch_bounds__(64, 1) void k_node_post(KArgs A) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int strip = blockIdx.x + A.strip0;
    const LaneNode L = lane_node(A, strip, j);
    const float* mr = mod_row(A, L.b) + A.mod_base;
    const float* ng2 = mr + 5 * 256;
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned oN2E = (unsigned)(A.wb[JB_N2E_W] * 4), oF1 = (unsigned)(A.wb[JB_FF1_W] * 4), oF2 = (unsigned)(A.wb[JB_FF2_W] * 4);
    const bool rot = rot_active(A);                           // rotated statistics: Q P (W_row h + b), Q P W_col h instead (dgt_pack.cpp rot_stats)
    const unsigned oRow = (unsigned)(A.wb[rot ? JB_ROWQ_W : JB_ROW_W] * 4), oCol = (unsigned)(A.wb[rot ? JB_COLQ_W : JB_COL_W] * 4), oNro = (unsigned)(A.wb[JB_NRO_W] * 4);
    WPipe<8> wp;
    wpipe_prime(wp, ws, oN2E);
    float hx[128];                                            // first the messages, then (in place) the FFN input
    node_load_hh(A, L, half, hx);
    // node2edge_lin applied per node (bias added on the edge side)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const unsigned cur = oN2E + (unsigned)b * 32 * 1024;
        f32x16 acc = mfma_block_p<32>(wp, ws, cur, b == 0 ? cur + 32 * 1024 : oF1, hx, zero16());
        float r[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = acc[s];
        store16T(A.n2e, 2, L.v, half, b, r);
    }
    node_residual_ln(A, L, half, mr, hx);
    // FFN: hidden R*256 in chunks of 64 features; ff2 accumulates over the chunks
    f32x16 o[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) o[b] = zero16();
    {
        const float* b1 = A.W + A.wb[JB_FF1_B];
        constexpr int KQ2 = R * 256 / 8;                      // quads per ff2 output block
#pragma unroll 1
        for (int c = 0; c < R * 4; ++c) {
            float hid[32];
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const unsigned cur = oF1 + (unsigned)(c * 2 + b2) * 32 * 1024;
                const unsigned nxt = b2 == 0 ? cur + 32 * 1024 : oF2 + (unsigned)(c * 8) * 1024;
                float bb[16];
                load16(b1 + (c * 2 + b2) * 32 + half * 16, bb);
                f32x16 acc = mfma_block_p<32>(wp, ws, cur, nxt, hx, zero16());
                silu_bias16(acc, bb, hid + b2 * 16);
            }
#pragma unroll
            for (int ob = 0; ob < 8; ++ob) {
                const unsigned cur = oF2 + (unsigned)(ob * KQ2 + c * 8) * 1024;
                const unsigned nxt = ob < 7 ? oF2 + (unsigned)((ob + 1) * KQ2 + c * 8) * 1024
                                            : (c + 1 < R * 4 ? oF1 + (unsigned)((c + 1) * 2) * 32 * 1024 : oRow);
                o[ob] = mfma_block_p<8>(wp, ws, cur, nxt, hid, o[ob]);
            }
        }
    }
    {
        const float* b2 = A.W + A.wb[JB_FF2_B];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float bb[16], g[16];
            load16(b2 + b * 32 + half * 16, bb);
            load16(ng2 + b * 32 + half * 16, g);
#pragma unroll
            for (int s = 0; s < 16; ++s) hx[b * 16 + s] = fmaf(g[s], o[b][s] + bb[s], hx[b * 16 + s]);
        }
    }
    store_nat<8>(A.h + (size_t)L.v * 256, half, hx);
    // per-node halves of equi_update.input_lin: W_row h (+ bias), W_col h
    {
        const float* bin = A.W + A.wb[rot ? JB_INQ_B : JB_IN_B];
#pragma unroll 1
        for (int b = 0; b < 8; ++b) {
            const unsigned cr = oRow + (unsigned)b * 32 * 1024, cc = oCol + (unsigned)b * 32 * 1024;
            float bb[16], r[16];
            load16(bin + b * 32 + half * 16, bb);
            f32x16 acc = mfma_block_p<32>(wp, ws, cr, cc, hx, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) r[s] = acc[s] + bb[s];
            store16T(A.wrow, 8, L.v, half, b, r);
            acc = mfma_block_p<32>(wp, ws, cc, b < 7 ? cr + 32 * 1024 : oNro, hx, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) r[s] = acc[s];
            store16T(A.wcol, 8, L.v, half, b, r);
        }
    }
    // readout node_l(h) -> atom_hids[:, D + l*64 ...]
    {
        const float* bias = A.W + A.wb[JB_NRO_B];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned cur = oNro + (unsigned)b * 32 * 1024;
            float bb[16], r[16];
            load16(bias + b * 32 + half * 16, bb);
            f32x16 acc = mfma_block_p<32>(wp, ws, cur, b == 0 ? cur + 32 * 1024 : (A.fuse_next ? (unsigned)(A.wbn[0] * 4) : oNro), hx, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) r[s] = acc[s] + bb[s];
            store16(A.ahid + (size_t)L.v * A.d.KNH + 256 + A.layer * 64 + b * 32 + half * 16, r);
        }
    }
    if (A.fuse_next) {                                        // next block's LN1 + modulate + q / k / v, h' still in registers
        node_next_ln(A, L, half, hx);
        node_next_qkv(A, L, half, hx, ws, wp, 0, 24);
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-wave variant: one workgroup of NW = 2 or 4 waves per strip.  The waves share the
// strip's work — FFN hidden chunks c = w, w + 4, ..; afterwards 4 of the 16 W_row / W_col blocks each —
// and exchange through LDS once: the partial FFN sums (4 x 32 KiB) are reduced by the wave that owns the
// output block, which then publishes its two blocks of h' in place of partial 0.  The cheap
// latency-bound prologue (message partial sums, residual, LayerNorm) is simply repeated by every wave.
// Why: with one wave per strip a launch has only Nn/32 items (1409 at QM9 B = 2500: 1.4 per SIMD -> two
// full rounds; 177 at B = 313: 83 % of the chip idle); here an item is ~3.7x shorter and the unit of
// scheduling is a CU.
template <int W, int NW>   // reduce + publish the 8 / NW output blocks owned by wave W
__device__ __forceinline__ void node_postw_reduce(const KArgs& A, const LaneNode& L, int half, int lane, float4* part,
                                                  const float* ng2, const float (&hx)[128]) {
    constexpr int NBW = 8 / NW;
    const float* b2 = A.W + A.wb[JB_FF2_B];
#pragma unroll
    for (int bb = 0; bb < NBW; ++bb) {
        const int b = NBW * W + bb;
        float sum[16], bias[16], g[16], r[16];
        load16(b2 + b * 32 + half * 16, bias);
        load16(ng2 + b * 32 + half * 16, g);
#pragma unroll
        for (int s = 0; s < 16; ++s) sum[s] = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w)                            // fixed order: bit-deterministic
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = part[((w * 8 + b) * 4 + q) * 64 + lane];
                sum[q * 4 + 0] += v.x; sum[q * 4 + 1] += v.y; sum[q * 4 + 2] += v.z; sum[q * 4 + 3] += v.w;
            }
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = fmaf(g[s], sum[s] + bias[s], hx[(NBW * W + bb) * 16 + s]);
        store16(A.h + (size_t)L.v * 256 + b * 32 + half * 16, r);
#pragma unroll
        for (int q = 0; q < 4; ++q)       // only this wave ever reads column b of the partials: safe to overwrite
            part[((0 * 8 + b) * 4 + q) * 64 + lane] = make_float4(r[q * 4 + 0], r[q * 4 + 1], r[q * 4 + 2], r[q * 4 + 3]);
    }
}

template <int R, int NW>   // mlp_ratio, waves per strip (2 or 4); part: [NW waves][8 blocks][4 quads][64 lanes] float4 = NW x 32 KiB of LDS
__device__ __forceinline__ void node_postw_body(const KArgs& A, int strip, float4* part) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    const LaneNode L = lane_node(A, strip, j);
    const float* mr = mod_row(A, L.b) + A.mod_base;
    const float* ng2 = mr + 5 * 256;
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned oN2E = (unsigned)(A.wb[JB_N2E_W] * 4), oF1 = (unsigned)(A.wb[JB_FF1_W] * 4), oF2 = (unsigned)(A.wb[JB_FF2_W] * 4);
    const bool rot = rot_active(A);                           // rotated statistics: Q P (W_row h + b), Q P W_col h instead (dgt_pack.cpp rot_stats)
    const unsigned oRow = (unsigned)(A.wb[rot ? JB_ROWQ_W : JB_ROW_W] * 4), oCol = (unsigned)(A.wb[rot ? JB_COLQ_W : JB_COL_W] * 4), oNro = (unsigned)(A.wb[JB_NRO_W] * 4);
    constexpr int NCH = R * 4, CPW = NCH / NW;                // hidden chunks of 64, chunks per wave
    constexpr int KQ2 = R * 256 / 8;
    constexpr int PER = 16 / NW;                              // W_row / W_col blocks per wave in the last phase
    constexpr int ROW_WAVES = NW / 2;                         // waves 0 .. NW/2-1 take W_row, the others W_col + readout
    constexpr int NRO_PER = 2 / ROW_WAVES;                    // readout blocks per W_col wave
    const bool is_row = wave < ROW_WAVES;
    const int b0 = (wave % ROW_WAVES) * PER;
    const unsigned oLast = (is_row ? oRow : oCol) + (unsigned)b0 * 32 * 1024;
    const unsigned oFirstFF = oF1 + (unsigned)(wave * 2) * 32 * 1024;
    WPipe<8> wp;
    wpipe_prime(wp, ws, wave < 2 ? oN2E + (unsigned)wave * 32 * 1024 : oFirstFF);
    float hx[128];
    node_load_hh(A, L, half, hx);                             // every wave merges the (few, L2-resident) partials itself
    if (wave < 2) {                                           // node2edge_lin: one block each on waves 0 and 1
        f32x16 acc = mfma_block_p<32>(wp, ws, oN2E + (unsigned)wave * 32 * 1024, oFirstFF, hx, zero16());
        float r[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = acc[s];
        store16T(A.n2e, 2, L.v, half, wave, r);
    }
    node_residual_ln(A, L, half, mr, hx);
    f32x16 o[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) o[b] = zero16();
    {
        const float* b1 = A.W + A.wb[JB_FF1_B];
#pragma unroll 1
        for (int ci = 0; ci < CPW; ++ci) {
            const int c = wave + NW * ci;
            float hid[32];
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const unsigned cur = oF1 + (unsigned)(c * 2 + b2) * 32 * 1024;
                const unsigned nxt = b2 == 0 ? cur + 32 * 1024 : oF2 + (unsigned)(c * 8) * 1024;
                float bb[16];
                load16(b1 + (c * 2 + b2) * 32 + half * 16, bb);
                f32x16 acc = mfma_block_p<32>(wp, ws, cur, nxt, hx, zero16());
                silu_bias16(acc, bb, hid + b2 * 16);
            }
#pragma unroll
            for (int ob = 0; ob < 8; ++ob) {
                const unsigned cur = oF2 + (unsigned)(ob * KQ2 + c * 8) * 1024;
                const unsigned nxt = ob < 7 ? oF2 + (unsigned)((ob + 1) * KQ2 + c * 8) * 1024
                                            : (ci + 1 < CPW ? oF1 + (unsigned)((c + NW) * 2) * 32 * 1024 : oLast);
                o[ob] = mfma_block_p<8>(wp, ws, cur, nxt, hid, o[ob]);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            part[((wave * 8 + b) * 4 + q) * 64 + lane] = make_float4(o[b][q * 4 + 0], o[b][q * 4 + 1], o[b][q * 4 + 2], o[b][q * 4 + 3]);
    __syncthreads();
    if (wave == 0) node_postw_reduce<0, NW>(A, L, half, lane, part, ng2, hx);
    else if (wave == 1) node_postw_reduce<1, NW>(A, L, half, lane, part, ng2, hx);
    else if (NW > 2 && wave == 2) node_postw_reduce<2 % NW, NW>(A, L, half, lane, part, ng2, hx);
    else if (NW > 2) node_postw_reduce<3 % NW, NW>(A, L, half, lane, part, ng2, hx);
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 8; ++b)                               // every wave picks up the full h'
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = part[((0 * 8 + b) * 4 + q) * 64 + lane];
            hx[b * 16 + q * 4 + 0] = v.x; hx[b * 16 + q * 4 + 1] = v.y; hx[b * 16 + q * 4 + 2] = v.z; hx[b * 16 + q * 4 + 3] = v.w;
        }
    {   // this wave's share of the 16 W_row / W_col blocks; the W_col waves then take the readout blocks
        const float* bin = A.W + A.wb[rot ? JB_INQ_B : JB_IN_B];
        float* dst = is_row ? A.wrow : A.wcol;
        const int ro0 = (wave % ROW_WAVES) * NRO_PER;         // first readout block of this (W_col) wave
        const unsigned oAfter = is_row ? oLast : oNro + (unsigned)ro0 * 32 * 1024;
        // first q / k / v block of this wave when the next block's projections are fused in (24 blocks over NW waves)
        const int g0 = wave * (24 / NW);
        const unsigned oNextQ = A.fuse_next ? (unsigned)(A.wbn[2 * (g0 >> 3)] * 4) + (unsigned)(g0 & 7) * 32 * 1024 : oLast;
#pragma unroll 1
        for (int bi = 0; bi < PER; ++bi) {
            const int b = b0 + bi;
            const unsigned cur = oLast + (unsigned)bi * 32 * 1024;
            float bb[16], r[16];
            load16(bin + b * 32 + half * 16, bb);
            f32x16 acc = mfma_block_p<32>(wp, ws, cur, bi + 1 < PER ? cur + 32 * 1024 : (is_row ? oNextQ : oAfter), hx, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) r[s] = acc[s] + (is_row ? bb[s] : 0.f);
            store16T(dst, 8, L.v, half, b, r);
        }
        if (!is_row) {                                        // readout node_l(h') -> atom_hids[:, D + l*64 ...]
            const float* bias = A.W + A.wb[JB_NRO_B];
#pragma unroll 1
            for (int k = 0; k < NRO_PER; ++k) {
                const int b = ro0 + k;
                const unsigned cur = oAfter + (unsigned)k * 32 * 1024;
                float bb[16], r[16];
                load16(bias + b * 32 + half * 16, bb);
                f32x16 acc = mfma_block_p<32>(wp, ws, cur, k + 1 < NRO_PER ? cur + 32 * 1024 : oNextQ, hx, zero16());
#pragma unroll
                for (int s = 0; s < 16; ++s) r[s] = acc[s] + bb[s];
                store16(A.ahid + (size_t)L.v * A.d.KNH + 256 + A.layer * 64 + b * 32 + half * 16, r);
            }
        }
        if (A.fuse_next) {                                    // next block's q / k / v: 24 / NW blocks per wave
            node_next_ln(A, L, half, hx);
            node_next_qkv(A, L, half, hx, ws, wp, g0, g0 + 24 / NW);
        }
    }
}
template <int R, int NW>
__global__ __launch_bounds__(NW * 64, 1) void k_node_postw(KArgs A) {
    __shared__ float4 part[NW * 8 * 4 * 64];
    node_postw_body<R, NW>(A, (int)blockIdx.x + A.strip0, part);
}

}  // namespace jd
