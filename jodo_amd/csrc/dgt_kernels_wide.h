// Width-generic kernel set (template <int D>): every per-block kernel of dgt_kernels_{pre,node,block,post}.h
// restated for any node width D = 128 * k (the README's GEOM Base model runs nf = 128, BASELINE config 4 nf = 384: De = 96, T = 1536, 14 x 27
// score channels, 16 x 24 value channels).  Same strip model, same HBM layouts, same launch sequence;
// what differs from the tuned nf = 256 set:
//   * all weights are streamed from L2 through the software-pipelined ring (the K = De projections of
//     the attention kernels no longer fit in LDS at De = 96), prefetch group of 4 quads at D = 384 (K = 96 gives
//     12-quad blocks), 8 at D = 256;
//   * score heads use the "one head per 32-row block" arrangement (csrc/dgt_pack.cpp
//     qk_out_map_wide): SC = 27 has no 16 + 2 split, and padding a head to 32 rows keeps its reduction
//     in-lane (the attention edge phase itself is the width-generic k_edge_attn, dgt_kernels_attn.h);
//   * the pair update parks half of the shared part of input_lin in LDS at D = 384 (registers hold the rest).
// Instantiated for D = 384; D = 256 is instantiated too so that the whole set can be pinned against
// the tuned kernels and the nf = 256 fixtures (jodo_cfg.layout = 1, tests only).
#pragma once
#include "dgt_kernels_common.h"
#include "dgt_kernels_attn.h"
#ifndef JODO_X_UPD_EARLY
#define JODO_X_UPD_EARLY 0        // experiment switch of the pair update's item top (see k_edge_update_sym)
#endif
#ifndef JODO_X_UPD_PERS
#define JODO_X_UPD_PERS 0         // pair update (folded form, one wave per item): persistent workgroups, the NEXT item's descriptor chain in flight
#endif                            // under the current item (1), its partner rows too (2); see k_edge_update_sym.  Measured: 1 % / nothing
#ifndef JODO_X_UPD_NEXT
#define JODO_X_UPD_NEXT 0         // pair update: the NEXT offset's partner position / edge row / node2edge row requested behind the last
#endif                            // weight prefetch of the current offset (items of several offsets; see k_edge_update_sym)
#define JODO_X_UPD_ROWS (JODO_X_UPD_NEXT || JODO_X_UPD_PERS >= 2)   // PERS = 2: the same across the items of a persistent workgroup

namespace jd {
namespace wide {

constexpr int NHEAD_BLOCKS = 14; // learned score heads, one 32-row block each

template <int D_>
struct Dim {
    static constexpr int D = D_, De = D_ / 4, ND = D_ / 32, NE = D_ / 128;
    static constexpr int HD = D_ / 2, HE = D_ / 8;                // registers per half-lane: node / edge vector
    static constexpr int KQD = D_ / 8, KQE = D_ / 32;             // weight quads per output block for K = D / K = De
    static constexpr int PG = (D_ % 256 == 0) ? 8 : 4;             // weight quads in flight per prefetch group (must divide D/32)
    static constexpr int C = D_ / 16;                              // value channels per head
    static constexpr float INV_SQRT_C = D_ == 256 ? 0.25f : (D_ == 384 ? 0.20412414523193150f : (D_ == 128 ? 0.35355339059327379f : 0.f));
    // (the padded readout widths cnp / cep depend on n_layers too: run-time values A.d.cnp, A.d.cep — dgt_plan.cpp)
    // modulation slice of one block: node 6 x D | edge 6 x De | equi (shift, scale) 2 x D | gbf (scale, shift)
    static constexpr int M_EDGE = 6 * D_, M_EQUI = 6 * D_ + 6 * (D_ / 4), M_GBF = 6 * D_ + 6 * (D_ / 4) + 2 * D_;
    static constexpr int M_WG = M_GBF + 32, M_BS = M_WG + D_;      // coord_mlp.0 pushed through the LayerNorm: W0 (1 + sc) | W0 sh + b0
};

// ------------------------------------------------------------------------------------------------
// prologue
template <int D, int KQ>
__global__ __launch_bounds__(64) void k_embed_nodes(KArgs A) {
    using X = Dim<D>;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int v = blockIdx.x * 32 + j;
    float x[KQ * 4];
    const float4* src = reinterpret_cast<const float4*>(A.feat + (size_t)v * (KQ * 8) + half * (KQ * 4));
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const float4 t = src[q];
        x[q * 4 + 0] = t.x; x[q * 4 + 1] = t.y; x[q * 4 + 2] = t.z; x[q * 4 + 3] = t.w;
    }
    const float4* w = wq(A, A.wg[JW_NODE_EMB_W], lane);
    const float* bias = A.W + A.wg[JW_NODE_EMB_B];
#pragma unroll
    for (int b = 0; b < X::ND; ++b) {
        f32x16 acc = mfma_block<KQ>(w + (size_t)b * KQ * 64, x, zero16());
        float r[16];
        acc_bias(acc, bias + b * 32 + half * 16, r);
        store16(A.h + (size_t)v * D + b * 32 + half * 16, r);
        store16(A.ahid + (size_t)v * A.d.KNH + b * 32 + half * 16, r);
    }
}

template <int D>
__device__ __forceinline__ void embed_edges_body(const KArgs& A, int it) {
    using X = Dim<D>;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int strip = A.pd.item_strip[it], t0 = A.pd.item_t0[it], t1 = A.pd.item_t1[it];
    const LaneNode L = lane_node(A, strip, j);
    const int ch = A.d.ch;
    const bool first = A.flags[FLAG_COND_NONZERO] == 0;
    const float* mr = mod_row(A, L.b);
    const float gscale = mr[0], gshift = mr[1];
    const float4 pc = reinterpret_cast<const float4*>(A.cpos)[L.v];
    const float* tab = A.W + A.wg[JW_GBF_TOP];
    const float4* w = wq(A, A.wg[JW_EDGE_EMB_W], lane);
    const float* bias = A.W + A.wg[JW_EDGE_EMB_B];
    for (int t = t0; t < t1; ++t) {
        const bool ok = L.valid && t < L.n;
        const int tc = ok ? t : 0;
        const int u = L.noff + tc;
        const size_t r = (size_t)L.eoff + (size_t)L.i * L.n + tc;
        // half_rows: the state is only kept for the row a pair's evaluating lane reads back, (i, i + d) with the offsets of pair_of()
        int dd = tc - L.i;
        if (dd < 0) dd += L.n;
        const bool wr = ok && (!A.half_rows || L.n > PAIR_GROUP_LANES || (dd != 0 && (2 * dd < L.n || (2 * dd == L.n && 2 * L.i < L.n))));
        const float4 pu = reinterpret_cast<const float4*>(A.cpos)[u];
        const float dx = pc.x - pu.x, dy = pc.y - pu.y, dz = pc.z - pu.z;
        const float d2c = dx * dx + dy * dy + dz * dz;
        const size_t din = (((size_t)L.b * A.pd.N + L.i) * A.pd.N + tc) * ch;
        float ein[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int f = half * 4 + s;
            float val = 0.f;
            if (f < ch) val = A.edge_x[din + f];
            else if (f < 2 * ch && A.cond_edge_x) val = A.cond_edge_x[din + (f - ch)];
            ein[s] = val;
        }
        int adj2d = 1;
        if (A.cond_edge_x) adj2d = A.cond_edge_x[din] >= A.d.edge_th ? 1 : 0;
        const int adjsp = d2c <= A.d.cutoff ? 1 : 0;
        float G[X::HE];
        if (first) {
#pragma unroll
            for (int s = 0; s < X::HE; ++s) G[s] = 0.f;
        } else {
            gbf_n<X::NE>(d2c, gscale, gshift, tab, half, G);
        }
#pragma unroll
        for (int b = 0; b < X::NE; ++b) {
            f32x16 acc = mfma_block<X::KQE>(w + (size_t)(b * (X::KQE + 1)) * 64, G, zero16());
            acc = mfma_block<1>(w + (size_t)(b * (X::KQE + 1) + X::KQE) * 64, ein, acc);
            float rr[16];
            acc_bias(acc, bias + b * 32 + half * 16, rr);
            if (wr) {
                store16(A.e + r * X::De + b * 32 + half * 16, rr);
                store16(A.ehid + r * A.d.KEH + b * 32 + half * 16, rr);
            }
        }
        if (ok && half == 0) A.eflag[r] = adj2d | (adjsp << 1);
    }
}
template <int D>
__global__ __launch_bounds__(64) void k_embed_edges(KArgs A) { embed_edges_body<D>(A, (int)blockIdx.x); }

// ------------------------------------------------------------------------------------------------
// node side
// NEXT: the items of the FOLLOWING block (weights A.wbn, modulation row A.mod_base_next; h already holds that block's input), issued
// from the launch that carries this block's k_node_ab items (k_node_ab_pre_w, dgt_forward.hip); positions are not touched.
// A.pre_mode == 1: the positions come from k_pos_final (every block when the items ride with k_node_ab)
template <int D, bool NEXT = false>
__device__ __forceinline__ void node_pre_body(const KArgs& A, int blk) {
    using X = Dim<D>;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int strip = blk / 3, piece = blk % 3;
    const LaneNode L = lane_node(A, strip, j);
    if (!NEXT && piece == 0 && A.pre_mode == 0) {
        float4 p = reinterpret_cast<const float4*>(A.pos_in)[L.v];
        if (A.layer > 0) p = advance_position(A, p, L.v, strip, L.n, L.i, L.eoff);
        if (half == 0) reinterpret_cast<float4*>(A.pos_out)[L.v] = p;
    }
    const float* mr = mod_row(A, L.b) + (NEXT ? A.mod_base_next : A.mod_base);
    float hx[X::HD];
    load_nat<X::ND>(A.h + (size_t)L.v * D, half, hx);
    layer_norm<X::HD>(hx);
    modulate<X::ND>(hx, mr, mr + D, half);
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned woff = (unsigned)((NEXT ? A.wbn[2 * piece] : A.wb[piece == 0 ? JB_WQ : (piece == 1 ? JB_WK : JB_WV)]) * 4);
    const float* bias = A.W + (NEXT ? A.wbn[2 * piece + 1] : A.wb[piece == 0 ? JB_BQ : (piece == 1 ? JB_BK : JB_BV)]);
    float* outp = piece == 0 ? A.q : (piece == 1 ? A.k : A.v);
    const int nb = piece == 2 ? X::ND : NHEAD_BLOCKS;
    WPipe<X::PG> wp;
    wpipe_prime(wp, ws, woff);
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
        const unsigned cur = woff + (unsigned)b * X::KQD * 1024;
        const unsigned nxt = b + 1 < nb ? cur + X::KQD * 1024 : woff;
        float bb[16], r[16];
        load16(bias + b * 32 + half * 16, bb);
        f32x16 acc = mfma_block_p<X::KQD>(wp, ws, cur, nxt, hx, zero16());
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = acc[s] + bb[s];
        store16T(outp, nb, L.v, half, b, r);
    }
}
template <int D>
__global__ __launch_bounds__(64, 1) void k_node_pre(KArgs A) { node_pre_body<D, false>(A, (int)blockIdx.x); }

template <int D, int R>
__global__ __launch_bounds__(64, 1) void k_node_post(KArgs A) {
    using X = Dim<D>;
    constexpr int NPASS = 1, NOB = X::ND / NPASS;        // ff2 output blocks per pass (one pass fits at D = 384: 192 inputs in
                                                          // VGPRs + 192 accumulators in AGPRs, no spills; two passes cost 27 % more MFMAs)
    constexpr int NCH = R * D / 64;                                       // hidden chunks of 64 features
    constexpr int KQ2 = R * D / 8;                                        // quads per ff2 output block
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int strip = blockIdx.x;
    const LaneNode L = lane_node(A, strip, j);
    const float* mr = mod_row(A, L.b) + A.mod_base;
    const float* ng1 = mr + 2 * D, *ns2 = mr + 3 * D, *nc2 = mr + 4 * D, *ng2 = mr + 5 * D;
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned oN2E = (unsigned)(A.wb[JB_N2E_W] * 4), oF1 = (unsigned)(A.wb[JB_FF1_W] * 4), oF2 = (unsigned)(A.wb[JB_FF2_W] * 4);
    const bool rot = rot_active(A);                        // rotated statistics: Q P (W_row h + b), Q P W_col h instead (dgt_pack.cpp rot_stats)
    const unsigned oRow = (unsigned)(A.wb[rot ? JB_ROWQ_W : JB_ROW_W] * 4), oCol = (unsigned)(A.wb[rot ? JB_COLQ_W : JB_COL_W] * 4), oNro = (unsigned)(A.wb[JB_NRO_W] * 4);
    WPipe<X::PG> wp;
    wpipe_prime(wp, ws, oN2E);
    float hx[X::HD];
    attn_merge<D>(A, L.v, half, hx);                                      // aggregated attention messages (dgt_kernels_attn.h)
#pragma unroll
    for (int b = 0; b < X::NE; ++b) {                                     // node2edge_lin per node
        const unsigned cur = oN2E + (unsigned)b * X::KQD * 1024;
        f32x16 acc = mfma_block_p<X::KQD>(wp, ws, cur, b + 1 < X::NE ? cur + X::KQD * 1024 : oF1, hx, zero16());
        float r[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = acc[s];
        store16T(A.n2e, X::NE, L.v, half, b, r);
    }
    {   // x = LN(h + ng1 * hh) * (1 + nc2) + ns2
        const float* hrow = A.h + (size_t)L.v * D;
#pragma unroll
        for (int b = 0; b < X::ND; ++b) {
            float g[16], h0[16];
            load16(ng1 + b * 32 + half * 16, g);
            load16(hrow + b * 32 + half * 16, h0);
#pragma unroll
            for (int s = 0; s < 16; ++s) hx[b * 16 + s] = fmaf(g[s], hx[b * 16 + s], h0[s]);
        }
        layer_norm<X::HD>(hx);
        modulate<X::ND>(hx, ns2, nc2, half);
    }
    // FFN: hidden R*D in chunks of 64; ff2 accumulates NOB output blocks per pass (the hidden chunk is
    // recomputed in every pass — cheaper than spilling D/2 accumulators per lane).  hx must stay intact
    // until the last pass (it is the ff1 input), so finished output blocks go straight to A.h and are
    // read back (same thread) as the B operand of the projections below.
    {
        const float* b1 = A.W + A.wb[JB_FF1_B];
        const float* b2 = A.W + A.wb[JB_FF2_B];
        float* hrow = A.h + (size_t)L.v * D;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            f32x16 o[NOB];
#pragma unroll
            for (int b = 0; b < NOB; ++b) o[b] = zero16();
#pragma unroll 1
            for (int c = 0; c < NCH; ++c) {
                float hid[32];
#pragma unroll
                for (int b2i = 0; b2i < 2; ++b2i) {
                    const unsigned cur = oF1 + (unsigned)(c * 2 + b2i) * X::KQD * 1024;
                    const unsigned nxt = b2i == 0 ? cur + X::KQD * 1024 : oF2 + (unsigned)((ps * NOB) * KQ2 + c * 8) * 1024;
                    float bb[16];
                    load16(b1 + (c * 2 + b2i) * 32 + half * 16, bb);
                    f32x16 acc = mfma_block_p<X::KQD>(wp, ws, cur, nxt, hx, zero16());
                    silu_bias16(acc, bb, hid + b2i * 16);
                }
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) {
                    const unsigned cur = oF2 + (unsigned)((ps * NOB + ob) * KQ2 + c * 8) * 1024;
                    const unsigned nxt = ob + 1 < NOB ? oF2 + (unsigned)((ps * NOB + ob + 1) * KQ2 + c * 8) * 1024
                                                      : (c + 1 < NCH ? oF1 + (unsigned)((c + 1) * 2) * X::KQD * 1024
                                                                     : (ps + 1 < NPASS ? oF1 : oRow));
                    o[ob] = mfma_block_p<8>(wp, ws, cur, nxt, hid, o[ob]);
                }
            }
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
                const int b = ps * NOB + ob;
                float bb[16], g[16], r[16];
                load16(b2 + b * 32 + half * 16, bb);
                load16(ng2 + b * 32 + half * 16, g);
#pragma unroll
                for (int s = 0; s < 16; ++s) r[s] = fmaf(g[s], o[ob][s] + bb[s], hx[b * 16 + s]);
                store16(hrow + b * 32 + half * 16, r);
            }
        }
        load_nat<X::ND>(hrow, half, hx);                                  // hx = h_out from here on
    }
    {   // per-node halves of equi_update.input_lin: W_row h (+ bias), W_col h
        const float* bin = A.W + A.wb[rot ? JB_INQ_B : JB_IN_B];
#pragma unroll 1
        for (int b = 0; b < X::ND; ++b) {
            const unsigned cr = oRow + (unsigned)b * X::KQD * 1024, cc = oCol + (unsigned)b * X::KQD * 1024;
            float bb[16], r[16];
            load16(bin + b * 32 + half * 16, bb);
            f32x16 acc = mfma_block_p<X::KQD>(wp, ws, cr, cc, hx, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) r[s] = acc[s] + bb[s];
            store16T(A.wrow, X::ND, L.v, half, b, r);
            acc = mfma_block_p<X::KQD>(wp, ws, cc, b + 1 < X::ND ? cr + X::KQD * 1024 : oNro, hx, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) r[s] = acc[s];
            store16T(A.wcol, X::ND, L.v, half, b, r);
        }
    }
    {   // readout node_l(h) -> atom_hids[:, D + l*CNP ...]
        const float* bias = A.W + A.wb[JB_NRO_B];
        const int nro = A.d.cnp / 32;                       // 2 D / n_layers features padded to whole blocks
#pragma unroll 1
        for (int b = 0; b < nro; ++b) {
            const unsigned cur = oNro + (unsigned)b * X::KQD * 1024;
            float bb[16], r[16];
            load16(bias + b * 32 + half * 16, bb);
            f32x16 acc = mfma_block_p<X::KQD>(wp, ws, cur, b + 1 < nro ? cur + X::KQD * 1024 : oNro, hx, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) r[s] = acc[s] + bb[s];
            store16(A.ahid + (size_t)L.v * A.d.KNH + D + A.layer * A.d.cnp + b * 32 + half * 16, r);
        }
    }
}

// coord_mlp.0 pushed through the LayerNorm of equi_update, per-node part (see k_edge_update_sym): piece 0: A = W0 (R (1 + sc)),
// R = W_row h + b (A.wrow); piece 1: B = W0 (C (1 + sc)), C = W_col h (A.wcol); also the feature means of R and C, which
// add up to the LayerNorm mean of a directed edge.  One (strip, piece) item per wave: 2 x n_strips items.
// Rotated statistics (rot_active): the rows are Q P R / Q P C; A' = F (Q P R) with the folded F = W0 diag(1 + sc) Q^T of this
// forward (k_fold_coord, A.ffold) — no scaling, no means (P removed them).
template <int D>
__device__ __forceinline__ void node_ab_body(const KArgs& A, int item) {
    if (D != 256 && !A.flags[FLAG_UNIFORM_T]) return;       // nf = 384 pushes coord_mlp.0 through only with a shared modulation row
    using X = Dim<D>;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int strip = item >> 1, piece = item & 1;
    const LaneNode L = lane_node(A, strip, j);
    const bool rot = rot_active(A);
    const float* qsc = mod_row(A, L.b) + A.mod_base + X::M_EQUI + D;       // equi_update.time_mlp: (shift, scale)
    const TRow src = trow(piece == 0 ? A.wrow : A.wcol, X::ND, L.v, half);
    float x[X::HD];
    float sum = 0.f;
    if (rot) {                                              // (one branch around two straight-line loops: a branch per block
#pragma unroll                                              //  serialised the row loads — 116 -> 161 us per launch)
        for (int b = 0; b < X::ND; ++b) {
            float t[16];
            load16T(src, b, t);
#pragma unroll
            for (int s = 0; s < 16; ++s) x[b * 16 + s] = t[s];
        }
    } else {
#pragma unroll
        for (int b = 0; b < X::ND; ++b) {
            float t[16], g[16];
            load16T(src, b, t);
            load16(qsc + b * 32 + half * 16, g);
#pragma unroll
            for (int s = 0; s < 16; ++s) { sum += t[s]; x[b * 16 + s] = t[s] * (1.f + g[s]); }
        }
    }
    if (!rot) {
        const float red = pair_sum(sum);
        if (half == 0) A.rmean[(size_t)L.v * 2 + piece] = red * (1.f / D);
    }
    // (the base pointer is selected, not the descriptor: a select between two buffer resources went through scratch)
    const float* wbase = rot ? A.ffold + (size_t)A.layer * D * D : A.W;
    const WSrc ws = make_wsrc(wbase, lane);
    const unsigned o0 = rot ? 0u : (unsigned)(A.wb[JB_C0_W] * 4);
    WPipe<X::PG> wp;
    wpipe_prime(wp, ws, o0);
    float* dst = piece == 0 ? A.ua : A.ub;
#pragma unroll 1
    for (int b = 0; b < X::ND; ++b) {
        const unsigned cur = o0 + (unsigned)b * X::KQD * 1024;
        f32x16 acc = mfma_block_p<X::KQD>(wp, ws, cur, b + 1 < X::ND ? cur + X::KQD * 1024 : o0, x, zero16());
        float r[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = acc[s];
        store16T(dst, X::ND, L.v, half, b, r);
    }
}
template <int D>
__global__ __launch_bounds__(64, 1) void k_node_ab(KArgs A) { node_ab_body<D>(A, A.ab0 + (int)blockIdx.x); }

// Rotated statistics, the features a pair's [e ; G] projection cannot reach: |(Q P R_a + Q P C_c)[2 De:]|^2 for every directed edge
// (a, c) of a molecule.  With R'' / C'' the upper D - 2 De features of the rotated rows the plain Gram form |R''_a|^2 + |C''_c|^2 +
// 2 <R''_a, C''_c> loses kappa^2 eps when the two rows nearly cancel (kappa = |R''| / |R'' + C''|): a distance-like input_lin,
// W_row ~ -W_col, on atoms whose features share a large common component — which trained weights may well be and random init never
// is (round-3 review).  The sum is therefore taken around a reference atom of the molecule (its first atom, ref):
//     R''_a + C''_c = (R''_a + C''_ref) + (C''_c - C''_ref) = a' + c',     T2 = |a'|^2 + |c'|^2 + 2 <a', c'>
// a' and c' are formed as vectors (one rounding each, like the plain path's R_a + C_c), the common component is gone before
// anything is squared, and the loss is back to kappa eps with kappa measured on the deviations from the reference atom.
// The inner products of a 32 x 32 tile of atoms are ONE chain of (D - 2 De) / 2 MFMAs: the strip-transposed row arrays are at once
// the A operand (atom a = lane & 31 supplies its k-slot) and the B operand (atom c).  One tile per wave; tiles = ordered pairs of
// strips that share a molecule (plan list gt_sa / gt_sc).  The reference rows sit in a strip <= the tile's own strips (molecules are
// contiguous in packed order), so the tiles' place in the merged launches (k_node_mix) is unchanged.
// A.rot == 2 (tests): the uncentred form of round 3.
template <int D>
__device__ __forceinline__ void node_gram_body(const KArgs& A, int tile) {
    if (!rot_active(A)) return;
    using X = Dim<D>;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int sa = A.pd.gt_sa[tile], sc = A.pd.gt_sc[tile];
    const float4* ra = reinterpret_cast<const float4*>(A.wrow) + (size_t)sa * X::ND * 256 + lane;
    const float4* cc = reinterpret_cast<const float4*>(A.wcol) + (size_t)sc * X::ND * 256 + lane;
    const int vc = sc * 32 + j, val = sa * 32 + j;
    const int nc = A.pd.node_n[vc], noffc = A.pd.node_noff[vc], ic = A.pd.node_i[vc];
    const bool centre = A.rot != 2;
    // C'' of the first atom of this lane's A-side molecule / B-side molecule (padding lanes: their own row, never used)
    const int refa = (centre && A.pd.node_n[val] > 0) ? A.pd.node_noff[val] : val, refc = (centre && nc > 0) ? noffc : vc;
    const float4* ma = reinterpret_cast<const float4*>(A.wcol) + (size_t)(refa >> 5) * X::ND * 256 + (refa & 31) + 32 * half;
    const float4* mc = reinterpret_cast<const float4*>(A.wcol) + (size_t)(refc >> 5) * X::ND * 256 + (refc & 31) + 32 * half;
    const float sgn = centre ? 1.f : 0.f;
    f32x16 acc = zero16();
    float na[4] = {0.f, 0.f, 0.f, 0.f}, nc4[4] = {0.f, 0.f, 0.f, 0.f};      // squared norms of this lane's halves of a', c'
#pragma unroll
    for (int b = 2 * X::NE; b < X::ND; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a0 = ra[(b * 4 + q) * 64], c0 = cc[(b * 4 + q) * 64], pa = ma[(b * 4 + q) * 64], pc = mc[(b * 4 + q) * 64];
            const float4 a = make_float4(fmaf(sgn, pa.x, a0.x), fmaf(sgn, pa.y, a0.y), fmaf(sgn, pa.z, a0.z), fmaf(sgn, pa.w, a0.w));
            const float4 c = make_float4(fmaf(-sgn, pc.x, c0.x), fmaf(-sgn, pc.y, c0.y), fmaf(-sgn, pc.z, c0.z), fmaf(-sgn, pc.w, c0.w));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, c.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, c.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, c.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, c.w, acc, 0, 0, 0);
            na[0] = fmaf(a.x, a.x, na[0]); na[1] = fmaf(a.y, a.y, na[1]); na[2] = fmaf(a.z, a.z, na[2]); na[3] = fmaf(a.w, a.w, na[3]);
            nc4[0] = fmaf(c.x, c.x, nc4[0]); nc4[1] = fmaf(c.y, c.y, nc4[1]); nc4[2] = fmaf(c.z, c.z, nc4[2]); nc4[3] = fmaf(c.w, c.w, nc4[3]);
        }
    const float nR = pair_sum((na[0] + na[1]) + (na[2] + na[3]));           // |a'|^2 of atom sa * 32 + (lane & 31), in both halves
    const float nC = pair_sum((nc4[0] + nc4[1]) + (nc4[2] + nc4[3]));
    const size_t eoffc = (size_t)A.pd.node_eoff[vc];
#pragma unroll
    for (int s = 0; s < 16; ++s) {                          // accumulator register s of half h holds row 8 (s / 4) + 4 h + s % 4
        const int m = 8 * (s >> 2) + 4 * half + (s & 3), va = sa * 32 + m;
        const float nRa = __shfl(nR, m);
        if (nc > 0 && A.pd.node_n[va] > 0 && A.pd.node_noff[va] == noffc)
            A.gramE[eoffc + (size_t)A.pd.node_i[va] * nc + ic] = nRa + nC + 2.f * acc[s];
    }
}
template <int D>
__global__ __launch_bounds__(64) void k_node_gram(KArgs A) { node_gram_body<D>(A, A.g0 + (int)blockIdx.x); }

// M_l = coord_mlp.0 diag(1 + sc_l) input_lin[:, e ; G] for every block l, in the streaming layout of input_lin's [e ; G]
// part (pack_projection: [out block][quad][lane] float4) — valid when all molecules share one modulation row (device flag
// UNIFORM_T), which is how sampling runs an unconditional model.  Both factors are read in their packed layouts: row o of
// coord_mlp.0 sits at lane (s & 3) + 4 h + 8 (s >> 2) of block o / 32 (o % 32 = 16 h + s), its column j at register
// m = 16 (j / 32) + j % 16 of half (j % 32) / 16.  One float4 of M per thread, sums in double; 2 x 134 MFLOP per forward.
template <int D, int KQI>     // KQI = quads per output block of the right-hand factor: 2 KQE ([e ; G] part of input_lin), KQD (Q^T)
__global__ __launch_bounds__(64) void k_fold_coord(KArgs A, FoldOffs F, float* out, int need_rot, unsigned short* out_s = nullptr) {
    if (!A.flags[FLAG_UNIFORM_T] || (need_rot && !rot_active(A))) return;
    using X = Dim<D>;
    constexpr int PER_L = X::ND * KQI;
    const int l = blockIdx.x / PER_L, r = blockIdx.x % PER_L, nb = r / KQI, q = r % KQI;
    const int lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const float* sc = A.mods + 32 + (size_t)l * A.d.MB + X::M_EQUI + D;
    const float4* c0 = reinterpret_cast<const float4*>(A.W + F.c0[l]) + (size_t)nb * X::KQD * 64;
    const float4* ine = reinterpret_cast<const float4*>(A.W + F.ine[l]);
    double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
#pragma unroll 2
    for (int qj = 0; qj < X::KQD; ++qj) {
#pragma unroll
        for (int khj = 0; khj < 2; ++khj) {
            const float4 w = c0[qj * 64 + i + 32 * khj];
            const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) {
                const int m = 4 * qj + cj, j = (m >> 4) * 32 + khj * 16 + (m & 15);
                const int s_ = j & 15, h_ = (j >> 4) & 1, ij = (s_ & 3) + 4 * h_ + 8 * (s_ >> 2);
                const float4 v = ine[((size_t)(j >> 5) * KQI + q) * 64 + ij + 32 * kh];
                const double f = (double)wv[cj] * (1.0 + (double)sc[j]);
                a0 += f * v.x; a1 += f * v.y; a2 += f * v.z; a3 += f * v.w;
            }
        }
    }
    const float m4[4] = {(float)a0, (float)a1, (float)a2, (float)a3};
    reinterpret_cast<float4*>(out)[((size_t)l * PER_L + r) * 64 + lane] = make_float4(m4[0], m4[1], m4[2], m4[3]);
    if (out_s) {
        // split image for the opt-in split-bf16 pair update (dgt_split.h): the SAME fp32 matrix as three bf16 terms, hi + mid + lo == m
        // exactly; quad q holds the f32 k-steps 4q .. 4q + 3 = elements 4 (q & 1) .. + 3 of K16 step q / 2
        // layout [l][out block][K16 step][term][lane][8]
        unsigned short t[3][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const __bf16 h = (__bf16)m4[c];
            const float r1 = m4[c] - (float)h;
            const __bf16 m = (__bf16)r1;
            const __bf16 lo = (__bf16)(r1 - (float)m);
            t[0][c] = __builtin_bit_cast(unsigned short, h); t[1][c] = __builtin_bit_cast(unsigned short, m); t[2][c] = __builtin_bit_cast(unsigned short, lo);
        }
        constexpr int NS = KQI / 2;
        const size_t stepbase = (((size_t)l * X::ND + nb) * NS + (q >> 1)) * 3;
#pragma unroll
        for (int term = 0; term < 3; ++term) {
            unsigned short* dst = out_s + ((stepbase + term) * 64 + lane) * 8 + 4 * (q & 1);
            *reinterpret_cast<uint2*>(dst) = make_uint2((unsigned)t[term][0] | ((unsigned)t[term][1] << 16), (unsigned)t[term][2] | ((unsigned)t[term][3] << 16));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// edge side (directed: rows r = eoff + a*n + c, a = source / row atom, c = target / column atom)
template <int D, int R>
__global__ __launch_bounds__(64, 1) void k_edge_update(KArgs A) {
    if (!A.flags[FLAG_ASYM]) return;                            // symmetric inputs: k_edge_update_sym runs instead
    using X = Dim<D>;
    constexpr int NCH = R * X::De / 64;                   // edge FFN hidden chunks of 64
    constexpr int KQ4 = R * X::De / 8;                    // quads per ff4 output block
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int it = blockIdx.x;
    const int strip = A.pd.item_strip[it], t0 = A.pd.item_t0[it], t1 = A.pd.item_t1[it], part = A.pd.item_part[it];
    const LaneNode L = lane_node(A, strip, j);
    const float* mrow = mod_row(A, L.b) + A.mod_base;
    const float* eg1 = mrow + X::M_EDGE + 2 * X::De;      // edge chunks: es1, ec1, eg1, es2, ec2, eg2
    const float* qsh = mrow + X::M_EQUI;                  // equi_update.time_mlp: (shift, scale)
    const float gscale = mrow[X::M_GBF + 0], gshift = mrow[X::M_GBF + 1];
    const float4 pv = reinterpret_cast<const float4*>(A.pos_out)[L.v];
    const float cscale = A.W[A.wb[JB_CSCALE]];
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned o3 = (unsigned)(A.wb[JB_FF3_W] * 4), o4 = (unsigned)(A.wb[JB_FF4_W] * 4);
    const unsigned oro = (unsigned)(A.wb[JB_ERO_W] * 4), oi = (unsigned)(A.wb[JB_INE_W] * 4), o0 = (unsigned)(A.wb[JB_C0_W] * 4);
    WPipe<X::PG> wp;
    wpipe_prime(wp, ws, o3);
    float dax = 0.f, day = 0.f, daz = 0.f;
    for (int t = t0; t < t1; ++t) {
        const bool inr = L.valid && t < L.n;
        const bool ok = inr && t != L.i;
        const int tc = inr ? t : 0;
        const int u = L.noff + tc;
        const size_t r = (size_t)L.eoff + (size_t)L.i * L.n + tc;
        const float* eg1_ = launder(eg1);
        const float* es2_ = eg1_ + X::De, *ec2_ = es2_ + X::De, *eg2_ = ec2_ + X::De;
        const float* qsh_ = launder(qsh);
        const float* qsc_ = qsh_ + D;
        const float* cst = launder(A.W);
        const float* n2bias_ = cst + A.wb[JB_N2E_B], *b3_ = cst + A.wb[JB_FF3_B], *b4_ = cst + A.wb[JB_FF4_B];
        const float* b0_ = cst + A.wb[JB_C0_B], *w2_ = cst + A.wb[JB_C2_W], *tab_ = cst + A.wb[JB_GBF];
        const float* bro_ = cst + A.wb[JB_ERO_B];
        TRow wrow = trow(A.wrow, X::ND, L.v, half);
        wrow.p = launder(wrow.p);
        const TRow wcol = trow(A.wcol, X::ND, u, half);
        const float4 pu = reinterpret_cast<const float4*>(A.pos_out)[u];
        const float dx = pv.x - pu.x, dy = pv.y - pu.y, dz = pv.z - pu.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        float G[X::HE];
        gbf_n<X::NE>(d2, gscale, gshift, tab_, half, G);
        // ---- edge residual + LN2 + modulate ----
        float en[X::HE];
        {
            const TRow ra = trow(A.n2e, X::NE, L.v, half), rc = trow(A.n2e, X::NE, u, half);
#pragma unroll
            for (int b = 0; b < X::NE; ++b) {
                float e[16], ta[16], tc2[16], g[16], bb[16];
                load16(A.e + r * X::De + b * 32 + half * 16, e);
                load16T(ra, b, ta);
                load16T(rc, b, tc2);
                load16(eg1_ + b * 32 + half * 16, g);
                load16(n2bias_ + b * 32 + half * 16, bb);
#pragma unroll
                for (int s = 0; s < 16; ++s) en[b * 16 + s] = fmaf(g[s], ta[s] + tc2[s] + bb[s], e[s]);
            }
        }
        layer_norm<X::HE>(en);
        modulate<X::NE>(en, es2_, ec2_, half);
        // ---- edge FFN ----
        {
            f32x16 o[X::NE];
#pragma unroll
            for (int b = 0; b < X::NE; ++b) o[b] = zero16();
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float hid[32];
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    const unsigned wcur = o3 + (unsigned)(c * 2 + b2) * X::KQE * 1024;
                    const unsigned wnx = b2 == 0 ? wcur + X::KQE * 1024 : o4 + (unsigned)(c * 8) * 1024;
                    float bb[16];
                    load16(b3_ + (c * 2 + b2) * 32 + half * 16, bb);
                    f32x16 acc = mfma_block_p<X::KQE>(wp, ws, wcur, wnx, en, zero16());
                    silu_bias16(acc, bb, hid + b2 * 16);
                }
#pragma unroll
                for (int ob = 0; ob < X::NE; ++ob) {
                    const unsigned wcur = o4 + (unsigned)(ob * KQ4 + c * 8) * 1024;
                    const unsigned wnx = ob + 1 < X::NE ? o4 + (unsigned)((ob + 1) * KQ4 + c * 8) * 1024
                                                        : (c + 1 < NCH ? o3 + (unsigned)((c + 1) * 2) * X::KQE * 1024 : oro);
                    o[ob] = mfma_block_p<8>(wp, ws, wcur, wnx, hid, o[ob]);
                }
            }
#pragma unroll
            for (int b = 0; b < X::NE; ++b) {
                float ob4[16], og2[16];
                load16(b4_ + b * 32 + half * 16, ob4);
                load16(eg2_ + b * 32 + half * 16, og2);
#pragma unroll
                for (int s = 0; s < 16; ++s) en[b * 16 + s] = fmaf(og2[s], o[b][s] + ob4[s], en[b * 16 + s]);
            }
        }
        if (inr) store_nat<X::NE>(A.e_out + r * X::De, half, en);
        // ---- readout edge_l(e) -> edge_hids[:, De + l*CEP ...] ----
        {
            float bb[16];
            load16(bro_ + half * 16, bb);
            f32x16 acc = mfma_block_p<X::KQE>(wp, ws, oro, oi, en, zero16());
            float rr[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) rr[s] = acc[s] + bb[s];
            if (inr && (half == 0 || A.d.cep == 32)) store16(A.ehid + r * A.d.KEH + X::De + A.layer * A.d.cep + half * 16, rr);
        }
        // ---- equivariant update: u = W_e e + W_d G + (W_row h_a + b) + W_col h_c ----
        float uu[X::HD];
#pragma unroll
        for (int b = 0; b < X::ND; ++b) {
            const unsigned we = oi + (unsigned)(b * 2 * X::KQE) * 1024, wg_ = we + X::KQE * 1024;
            float a1[16], a2[16];
            load16T(wrow, b, a1);
            load16T(wcol, b, a2);
            f32x16 acc = mfma_block_p<X::KQE>(wp, ws, we, wg_, en, zero16());
            acc = mfma_block_p<X::KQE>(wp, ws, wg_, b + 1 < X::ND ? wg_ + X::KQE * 1024 : o0, G, acc);
#pragma unroll
            for (int s = 0; s < 16; ++s) uu[b * 16 + s] = acc[s] + a1[s] + a2[s];
        }
        layer_norm<X::HD>(uu);
        modulate<X::ND>(uu, qsh_, qsc_, half);
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll 1
        for (int b = 0; b < X::ND; ++b) {
            const unsigned wcur = o0 + (unsigned)b * X::KQD * 1024;
            const unsigned wnx = b + 1 < X::ND ? wcur + X::KQD * 1024 : o3;
            float bb[16], k0[16], k1[16], k2[16];
            load16(b0_ + b * 32 + half * 16, bb);
            load16(w2_ + b * 32 + half * 16, k0);
            load16(w2_ + D + b * 32 + half * 16, k1);
            load16(w2_ + 2 * D + b * 32 + half * 16, k2);
            f32x16 acc = mfma_block_p<X::KQD>(wp, ws, wcur, wnx, uu, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float ys = silu_f(acc[s] + bb[s]);
                c0 = fmaf(ys, k0[s], c0);
                c1 = fmaf(ys, k1[s], c1);
                c2 = fmaf(ys, k2[s], c2);
            }
        }
        c0 = tanh_f(pair_sum(c0));
        c1 = tanh_f(pair_sum(c1));
        c2 = tanh_f(pair_sum(c2));
        const int fl = A.eflag[r];
        const float iota = (c0 + ((fl & 1) ? c1 : 0.f) + ((fl & 2) ? c2 : 0.f)) * (1.f / 3.f);
        const float nrm = fmaxf(sqrtf(d2), 1e-8f);
        const float f = ok ? cscale * iota / nrm : 0.f;
        dax = fmaf(dx, f, dax);
        day = fmaf(dy, f, day);
        daz = fmaf(dz, f, daz);
    }
    if (half == 0)
        reinterpret_cast<float4*>(A.dpos)[(size_t)L.v * A.pd.max_parts + part] = make_float4(dax, day, daz, 0.f);
}

// ------------------------------------------------------------------------------------------------
// Pair (symmetric-input) variants, see dgt_kernels_sym.h for the enumeration and the argument why the edge
// state stays exactly symmetric.  Weights are streamed; the symmetric part S of input_lin (D/2 values per
// lane) is kept in a per-lane array for both directions (registers at D = 256, partly spilled to scratch at
// D = 384 — four waves' LDS slabs of 48 KiB would not fit the CU's 160 KiB), and each direction requests its
// per-node rows with buffer loads pinned ahead of their use (BRow, dgt_device.h).
// ZW > 1 (round 5): the items of a launch's sparsely filled LAST round (launch_update_sym, dgt_edge.hip) run as workgroups of ZW
// waves.  Every wave evaluates the shared trunk and the LayerNorm statistics; the output blocks of the per-pair coord_mlp.0
// projection Z and their SiLU / coord_mlp.2 tails — half of a folded item's matrix work, 56 % of an unfolded one's — are dealt to
// the waves, whose partial dot products meet in LDS in a fixed order (wave 0 finishes the item and does every store).
template <int D, int R, bool FOLD = false, bool ROT = false, int ZW = 1>
__global__ __launch_bounds__(ZW * 64, 1) void k_edge_update_sym(KArgs A) {
    static_assert(FOLD || !ROT, "the rotated statistics need the shared modulation row");
    static_assert(ZW == 1 || D == 256 || FOLD, "the Z split needs the hoisted form (one coord_mlp.0 per pair)");
    if (A.flags[FLAG_ASYM]) return;
    // FOLD: every molecule shares one modulation row (unconditional sampling, one noise level per batch), so
    // coord_mlp.0 (1 + sc) input_lin[e ; G] is ONE D x 2De matrix per block (k_fold_coord) — see the hoist below.
    // Both variants are launched; the device flag picks the one that works.
    if ((A.flags[FLAG_UNIFORM_T] != 0) != FOLD) return;
    using X = Dim<D>;
    constexpr int NCH = R * X::De / 64;
    constexpr int KQ4 = R * X::De / 8;
    const int lane = threadIdx.x & 63, jl = lane & 31, half = lane >> 5;
    const int zw = ZW > 1 ? (int)(threadIdx.x >> 6) : 0;    // this wave's share of the Z blocks: [zb0, zb1)
    const int zb0 = zw * (X::ND / ZW), zb1 = zb0 + X::ND / ZW;
    static_assert(X::ND % ZW == 0, "Z blocks per wave");
    // dir_split: the items of a launch's sparsely filled last round get two workgroups each; both compute the
    // shared trunk, each evaluates one direction (item time x 0.64) — the trunk's stores come from direction 0
#if JODO_X_UPD_PERS
    // PERS (the launcher starts min(items, 1024) workgroups and sets A.pers_n = items of the launch): a workgroup walks the items block index
    // + k * grid.  What a new workgroup pays per item — kernel
    // arguments, coord_mlp.2 to LDS, the first weight group, item -> strip -> lane descriptors -> own position — is paid once, and the
    // next item's descriptors are requested in the middle of the current one.
    constexpr bool PERS = FOLD && ZW == 1;                   // (every launch of these instances is persistent: launch_sym_variant)
    constexpr bool pers = PERS;
    int it_c = A.item0 + (A.dir_split ? (int)(blockIdx.x >> 1) : (int)blockIdx.x);
    int strip_c = A.pd.pitem_strip[it_c], t0_c = A.pd.pitem_t0[it_c], t1_c = A.pd.pitem_t1[it_c];
    LaneNode L_c = lane_node(A, strip_c, jl);
    float4 pv_c = reinterpret_cast<const float4*>(A.pos_out)[L_c.v];
    const int dsel = A.dir_split ? (int)(blockIdx.x & 1) : -1;
    const float* mrow = mod_row(A, L_c.b) + A.mod_base;     // (PERS: the shared row — FOLD runs under FLAG_UNIFORM_T only)
#else
    const int it = A.item0 + (A.dir_split ? (int)(blockIdx.x >> 1) : (int)blockIdx.x);
    const int dsel = A.dir_split ? (int)(blockIdx.x & 1) : -1;
    const int strip = A.pd.pitem_strip[it], t0 = A.pd.pitem_t0[it], t1 = A.pd.pitem_t1[it];
    const LaneNode L = lane_node(A, strip, jl);
    const float* mrow = mod_row(A, L.b) + A.mod_base;
#endif
    const float* eg1 = mrow + X::M_EDGE + 2 * X::De;
    const float* qsh = mrow + X::M_EQUI;
    const float gscale = mrow[X::M_GBF + 0], gshift = mrow[X::M_GBF + 1];
#if !JODO_X_UPD_PERS
    const float4 pv = reinterpret_cast<const float4*>(A.pos_out)[L.v];
#endif
    const float cscale = A.W[A.wb[JB_CSCALE]];
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned o3 = (unsigned)(A.wb[JB_FF3_W] * 4), o4 = (unsigned)(A.wb[JB_FF4_W] * 4);
    const unsigned oro = (unsigned)(A.wb[JB_ERO_W] * 4), oi = (unsigned)(A.wb[JB_INE_W] * 4), o0 = (unsigned)(A.wb[JB_C0_W] * 4);
    WPipe<X::PG> wp;
    wpipe_prime(wp, ws, o3);
    // S = shared part of input_lin, kept for both directions: in registers; at nf = 384 (192 values per lane next to the 192
    // of u) the upper half is parked in LDS instead (24 KiB per one-wave workgroup, four per CU) — spilled to scratch before
    constexpr bool HOIST = D == 256 || FOLD;                // coord_mlp.0 once per pair (below); nf = 384 without a shared
                                                            // modulation row keeps one per direction (S (1 + sc) does not fit)
    constexpr int PLB = D > 256 ? X::ND / 2 : 0;            // blocks parked in LDS
    __shared__ float4 pl[PLB > 0 ? PLB * 4 * 64 : 1];
    // coord_mlp.2 (3 x D, the same for every item): read 16 times per pair offset by the tails right where it is needed; from
    // global memory each read was an exposed L1 round trip between two MFMA blocks, from LDS it is a short ds_read
    __shared__ float4 w2s[HOIST ? 3 * D / 4 : 1];
    __shared__ float zred[ZW > 1 ? (ZW - 1) * 12 * 64 : 1];     // partial coord_mlp.2 dot products of waves 1 .. ZW - 1
#if (JODO_X_UPD_EARLY & 2)
    // experiment (round 6, cycle budget of profiles/r06_cycle_budget.txt: the item's top holds 10.9 k of its 24.6 k waiting cycles):
    // coord_mlp.2 is requested here and parked in LDS only when the first tail is near — nothing waits for it at the top
    constexpr int W2N = (3 * D / 4 + ZW * 64 - 1) / (ZW * 64);
    float4 w2r[HOIST ? W2N : 1];
    if constexpr (HOIST) {
        const float4* src = reinterpret_cast<const float4*>(A.W + A.wb[JB_C2_W]);
#pragma unroll
        for (int i = 0; i < W2N; ++i) { const int k = (int)threadIdx.x + i * ZW * 64; w2r[i] = src[k < 3 * D / 4 ? k : 0]; }
    }
#else
    if constexpr (HOIST) {
        const float4* src = reinterpret_cast<const float4*>(A.W + A.wb[JB_C2_W]);
        for (int i = threadIdx.x; i < 3 * D / 4; i += ZW * 64) w2s[i] = src[i];
        __syncthreads();
    }
#endif
    // (Tried and dropped: the trunk's other item-invariant vectors — modulation chunks, biases, Gaussian table, 3 KiB — from an
    // LDS page as well: pair update 6.18 -> 6.77 ms/step at QM9 B = 2500, 55.3 -> 58.3 at nf = 384.  Their global loads are
    // requested far ahead and overlap; the ds_reads wait in order behind each other.)
    float park[HOIST ? 1 : (X::ND - PLB) * 16];
#if JODO_X_UPD_ROWS
    // NEXT: an item of several pair offsets requests the rows of offset t + 1 — partner position, the pair's edge row, the partner's
    // node2edge row: 68 registers — behind the LAST weight prefetch of offset t (the hook of the last Z block): they travel under that
    // block's MFMAs, its tails and the item end (~ 5 k cycles) instead of being waited for at the top of offset t + 1, and they stand
    // behind every weight group offset t still waits for (loads return in order: requested any earlier they delay those).
    constexpr bool NEXT = FOLD && ZW == 1;              // (the un-folded form keeps S (1 + sc) per lane: no room)
    float er[NEXT ? X::HE : 1], tcr[NEXT ? X::HE : 1];
    u32x4 pur = {0u, 0u, 0u, 0u};
    const __amdgpu_buffer_rsrc_t rpos_n = __builtin_amdgcn_make_buffer_rsrc(A.pos_out, 0, 0x7fffffff, 0x00020000);
    auto request_rows = [&](const LaneNode& Lq, int tn) {   // rows of offset tn of the strip whose lanes are Lq
        if constexpr (NEXT) {
            // edge rows of a strip's molecules lie within a few hundred MB of the strip's first molecule: descriptor at that row, 32-bit offsets
            const int eoff0_n = __builtin_amdgcn_readfirstlane(Lq.eoff);
            const __amdgpu_buffer_rsrc_t re_n = __builtin_amdgcn_make_buffer_rsrc(A.e + (size_t)eoff0_n * X::De, 0, 0x7fffffff, 0x00020000);
            const PairLane Pn = pair_of(Lq, tn + 1);
            pur = __builtin_amdgcn_raw_buffer_load_b128(rpos_n, (unsigned)Pn.u * 16u, 0, 0);
            const size_t rn = Lq.valid ? Pn.rij - (size_t)eoff0_n : 0;     // (padding lanes carry node 0's descriptors)
            const unsigned evoff = (unsigned)((rn * X::De + half * 16) * 4);
            const BRow rc = brow(A.n2e, X::NE, Pn.u, half);
#pragma unroll
            for (int b = 0; b < X::NE; ++b) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(re_n, evoff, (unsigned)(b * 128 + q * 16), 0);
                    er[b * 16 + q * 4 + 0] = __uint_as_float(v.x); er[b * 16 + q * 4 + 1] = __uint_as_float(v.y);
                    er[b * 16 + q * 4 + 2] = __uint_as_float(v.z); er[b * 16 + q * 4 + 3] = __uint_as_float(v.w);
                }
                float t16[16];
                bload16(rc, b, t16);
#pragma unroll
                for (int s = 0; s < 16; ++s) tcr[b * 16 + s] = t16[s];
            }
        }
    };
#endif
#if JODO_X_UPD_PERS >= 2
    request_rows(L_c, t0_c);                                // the first item's rows; every later item's come from the item before it
#endif
#if JODO_X_UPD_PERS
    for (;;) {                                              // one pass per item (not persistent: exactly one)
    const int strip = strip_c, t0 = t0_c, t1 = t1_c;
    const LaneNode L = L_c;
    const float4 pv = pv_c;
#if JODO_X_UPD_ROWS
    const int t_end = t1;                               // (the block loops below have locals named t0 / t1)
#endif
    // the next item of this workgroup: block index + k * grid (a ticket counter was tried first: hipcc turns the atomic into its wave-reduced
    // form and waits for it with vmcnt(0) on the spot — one exposed atomic round trip and a drained weight pipe at every item top)
    const int it_n = it_c + (int)gridDim.x;
    const bool more = pers && it_n - A.item0 < A.pers_n;
    it_c = more ? it_n : it_c;
    const int strip_n = A.pd.pitem_strip[it_c], t0_n = A.pd.pitem_t0[it_c], t1_n = A.pd.pitem_t1[it_c];   // scalar loads, back long before the hook below
#endif
#if JODO_X_UPD_ROWS && !JODO_X_UPD_PERS
    const int t_end = t1;
#endif
    PT_INIT
    for (int t = t0; t < t1; ++t) {
        const PairLane P = pair_of(L, t + 1);
#if JODO_X_UPD_PERS >= 2
        constexpr bool pre = NEXT;                         // every offset's rows were requested by its predecessor (or the prologue)
#elif JODO_X_UPD_ROWS
        const bool pre = NEXT && t > t0;                   // this offset's rows were requested by the previous one
#endif
        const float* eg1_ = launder(eg1);
        const float* es2_ = eg1_ + X::De, *ec2_ = es2_ + X::De, *eg2_ = ec2_ + X::De;
        const float* qsh_ = launder(qsh);
        const float* qsc_ = qsh_ + D;
        const float* cst = launder(A.W);
        const float* n2bias_ = cst + A.wb[JB_N2E_B], *b3_ = cst + A.wb[JB_FF3_B], *b4_ = cst + A.wb[JB_FF4_B];
        const float* b0_ = cst + A.wb[JB_C0_B], *w2_ = cst + A.wb[JB_C2_W], *tab_ = cst + A.wb[JB_GBF];
        const float* bro_ = cst + A.wb[JB_ERO_B];
        const BRow wrow_i = brow(A.wrow, X::ND, L.v, half), wcol_i = brow(A.wcol, X::ND, L.v, half);
        const BRow wrow_j = brow(A.wrow, X::ND, P.u, half), wcol_j = brow(A.wcol, X::ND, P.u, half);
#if (JODO_X_UPD_EARLY & 1)
        // the partner's position first, then the pair's edge row and the two node2edge rows: all requested NOW through buffer
        // descriptors (ordered by the fence; plain global loads are sunk to their use), consumed behind the Gaussian basis
        const __amdgpu_buffer_rsrc_t rpos = __builtin_amdgcn_make_buffer_rsrc(A.pos_out, 0, 0x7fffffff, 0x00020000);
        const u32x4 pur = __builtin_amdgcn_raw_buffer_load_b128(rpos, (unsigned)P.u * 16u, 0, 0);
        // edge rows of a strip's molecules lie within a few hundred MB of the strip's first molecule: descriptor at that row, 32-bit offsets
        const int eoff0 = __builtin_amdgcn_readfirstlane(L.eoff);
        const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(A.e + (size_t)eoff0 * X::De, 0, 0x7fffffff, 0x00020000);
        const unsigned evoff = (unsigned)(((P.rij - (size_t)eoff0) * X::De + half * 16) * 4);
        float er[X::HE], tar[X::HE], tcr[X::HE];
        {
            const BRow ra = brow(A.n2e, X::NE, L.v, half), rc = brow(A.n2e, X::NE, P.u, half);
#pragma unroll
            for (int b = 0; b < X::NE; ++b) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(re, evoff, (unsigned)(b * 128 + q * 16), 0);
                    er[b * 16 + q * 4 + 0] = __uint_as_float(v.x); er[b * 16 + q * 4 + 1] = __uint_as_float(v.y);
                    er[b * 16 + q * 4 + 2] = __uint_as_float(v.z); er[b * 16 + q * 4 + 3] = __uint_as_float(v.w);
                }
                float t16[16];
                bload16(ra, b, t16);
#pragma unroll
                for (int s = 0; s < 16; ++s) tar[b * 16 + s] = t16[s];
                bload16(rc, b, t16);
#pragma unroll
                for (int s = 0; s < 16; ++s) tcr[b * 16 + s] = t16[s];
            }
        }
        pipeline_fence();
        const float4 pu = make_float4(__uint_as_float(pur.x), __uint_as_float(pur.y), __uint_as_float(pur.z), __uint_as_float(pur.w));
#elif JODO_X_UPD_ROWS
        float4 pu;
        if (pre) pu = make_float4(__uint_as_float(pur.x), __uint_as_float(pur.y), __uint_as_float(pur.z), __uint_as_float(pur.w));
        else pu = reinterpret_cast<const float4*>(A.pos_out)[P.u];
#else
        const float4 pu = reinterpret_cast<const float4*>(A.pos_out)[P.u];
#endif
        const float dx = pv.x - pu.x, dy = pv.y - pu.y, dz = pv.z - pu.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        float G[X::HE];
        gbf_n<X::NE>(d2, gscale, gshift, tab_, half, G);
        // ---- edge residual + LN2 + modulate (symmetric) ----
        float en[X::HE];
        {
#if !(JODO_X_UPD_EARLY & 1)
            const TRow ra = trow(A.n2e, X::NE, L.v, half), rc = trow(A.n2e, X::NE, P.u, half);
#endif
#pragma unroll
            for (int b = 0; b < X::NE; ++b) {
                float e[16], ta[16], tc2[16], g[16], bb[16];
#if (JODO_X_UPD_EARLY & 1)
#pragma unroll
                for (int s = 0; s < 16; ++s) { e[s] = er[b * 16 + s]; ta[s] = tar[b * 16 + s]; tc2[s] = tcr[b * 16 + s]; }
#elif JODO_X_UPD_ROWS
                if (pre) {
#pragma unroll
                    for (int s = 0; s < 16; ++s) { e[s] = er[NEXT ? b * 16 + s : 0]; tc2[s] = tcr[NEXT ? b * 16 + s : 0]; }
                } else {
                    load16(A.e + P.rij * X::De + b * 32 + half * 16, e);
                    load16T(rc, b, tc2);
                }
                load16T(ra, b, ta);
#else
                load16(A.e + P.rij * X::De + b * 32 + half * 16, e);
                load16T(ra, b, ta);
                load16T(rc, b, tc2);
#endif
                load16(eg1_ + b * 32 + half * 16, g);
                load16(n2bias_ + b * 32 + half * 16, bb);
#pragma unroll
                for (int s = 0; s < 16; ++s) en[b * 16 + s] = fmaf(g[s], ta[s] + tc2[s] + bb[s], e[s]);
            }
        }
        layer_norm<X::HE>(en);
        modulate<X::NE>(en, es2_, ec2_, half);
        PT(0);
        // ---- edge FFN ----
        {
            f32x16 o[X::NE];
#pragma unroll
            for (int b = 0; b < X::NE; ++b) o[b] = zero16();
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float hid[32];
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    const unsigned wcur = o3 + (unsigned)(c * 2 + b2) * X::KQE * 1024;
                    const unsigned wnx = b2 == 0 ? wcur + X::KQE * 1024 : o4 + (unsigned)(c * 8) * 1024;
                    float bb[16];
                    load16(b3_ + (c * 2 + b2) * 32 + half * 16, bb);
                    f32x16 acc = mfma_block_p<X::KQE>(wp, ws, wcur, wnx, en, zero16());
                    silu_bias16(acc, bb, hid + b2 * 16);
                }
#pragma unroll
                for (int ob = 0; ob < X::NE; ++ob) {
                    const unsigned wcur = o4 + (unsigned)(ob * KQ4 + c * 8) * 1024;
                    const unsigned wnx = ob + 1 < X::NE ? o4 + (unsigned)((ob + 1) * KQ4 + c * 8) * 1024
                                                        : (c + 1 < NCH ? o3 + (unsigned)((c + 1) * 2) * X::KQE * 1024 : oro);
                    o[ob] = mfma_block_p<8>(wp, ws, wcur, wnx, hid, o[ob]);
                }
            }
#pragma unroll
            for (int b = 0; b < X::NE; ++b) {
                float ob4[16], og2[16];
                load16(b4_ + b * 32 + half * 16, ob4);
                load16(eg2_ + b * 32 + half * 16, og2);
#pragma unroll
                for (int s = 0; s < 16; ++s) en[b * 16 + s] = fmaf(og2[s], o[b][s] + ob4[s], en[b * 16 + s]);
            }
        }
        if (P.ok && dsel != 1 && zw == 0) {
            store_nat<X::NE>(A.e_out + P.rij * X::De, half, en);
            if (!A.half_rows || L.n > PAIR_GROUP_LANES) store_nat<X::NE>(A.e_out + P.rji * X::De, half, en);
        }
        PT(1);
#if JODO_X_UPD_PERS
        if constexpr (pers) {                               // the next item's lane descriptors and own position (its strip is known since the item top)
            strip_c = strip_n; t0_c = t0_n; t1_c = t1_n;
            L_c = lane_node(A, strip_c, jl);
            pv_c = reinterpret_cast<const float4*>(A.pos_out)[L_c.v];
            pipeline_fence();
        }
#endif
        // per-node rows of the coord_mlp.0 hoist (own rows do not depend on the pair offset: without an opaque offset LICM
        // hoists and spills them)
        unsigned opq = 0;
        asm volatile("" : "+v"(opq));
        BRow ua_i = brow(A.ua, X::ND, L.v, half), ub_i = brow(A.ub, X::ND, L.v, half);
        const BRow ua_j = brow(A.ua, X::ND, P.u, half), ub_j = brow(A.ub, X::ND, P.u, half);
        BRow own_r = wrow_i, own_c = wcol_i;
        ua_i.voff += opq; ub_i.voff += opq; own_r.voff += opq; own_c.voff += opq;
        float n0[16], n1[16], n2[16], n3[16];
        constexpr int NB2 = 2 * X::NE;                          // blocks of the triangular factor L (K = 2 De)
        constexpr int KQL = 2 * X::KQE;                         // quads per block row of the packed L
        const unsigned oL = (unsigned)(A.wb[JB_LQ_W] * 4);
        float gr0 = 0.f, gr1 = 0.f;
        if constexpr (ROT) {                                    // rows of the first (shortest) L block: requested ahead of the readout
            bload16(own_r, NB2 - 1, n0); bload16(wcol_j, NB2 - 1, n1); bload16(wrow_j, NB2 - 1, n2); bload16(own_c, NB2 - 1, n3);
            gr0 = A.gramE[P.rij]; gr1 = A.gramE[P.rji];
        }
        // ---- readout ----
        {
            float bb[16];
            load16(bro_ + half * 16, bb);
            f32x16 acc;
            if constexpr (ROT) acc = mfma_block_g<X::KQE, (4 < X::PG ? 4 : X::PG)>(wp, ws, oro, ws, oL + (unsigned)((NB2 - 1) * KQL + 4 * (NB2 - 1)) * 1024, en, zero16());
            else acc = mfma_block_p<X::KQE>(wp, ws, oro, oi, en, zero16());
            float rr[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) rr[s] = acc[s] + bb[s];
            if (P.ok && dsel != 1 && zw == 0 && (half == 0 || A.d.cep == 32)) {
                store16(A.ehid + P.rij * A.d.KEH + X::De + A.layer * A.d.cep + half * 16, rr);
                if (!A.half_rows || L.n > PAIR_GROUP_LANES) store16(A.ehid + P.rji * A.d.KEH + X::De + A.layer * A.d.cep + half * 16, rr);
            }
        }
        PT(2);
#if (JODO_X_UPD_EARLY & 2)
        if constexpr (HOIST) {
            if (t == t0) {
#pragma unroll
                for (int i = 0; i < W2N; ++i) { const int k = (int)threadIdx.x + i * ZW * 64; if (k < 3 * D / 4) w2s[k] = w2r[i]; }
                __syncthreads();
            }
        }
#endif
        if constexpr (HOIST) {
            // ---- coord_mlp.0 pushed through the LayerNorm: ONE D x D projection per pair instead of one per direction ----
            // u = LN(pre) (1 + sc) + sh with pre = S + R_a + C_c (R = W_row h + b, C = W_col h) is affine in pre once its
            // mean / rstd are known:  y = coord_mlp.0(u) = [Z + A_a + B_c - mean * wg] * rstd + bs  with
            //   Z = W0 (S (1 + sc))                  computed here, once per pair
            //   A_a = W0 (R_a (1 + sc)), B_c = ...   per node, k_node_ab
            //   wg = W0 (1 + sc), bs = W0 sh + b0    per molecule, rows of the modulation table (dgt_pack.cpp)
            // What remains per direction is vector work: the LayerNorm statistics of pre and the SiLU / coord_mlp.2 tail.
            // The statistics pass rides on the S projection and the tail pass on the Z projection, block by block: the
            // per-node rows they gather are requested one block ahead and arrive behind 4k / 8k cycles of MFMAs.  The mean
            // of pre is meanS + mean(R_a) + mean(C_c); only meanS is unknown while S is being produced, so the variance is
            // accumulated around m0 = mean(R_a) + mean(C_c) and corrected: var = E[(pre - m0)^2] - meanS^2 (meanS is the
            // mean of a bias-free projection of a normalised vector: small against the spread, no cancellation).
            float sg[FOLD ? 1 : X::HD];
            const WSrc wm = make_wsrc(A.mfold + (size_t)A.layer * D * 2 * X::De, lane);      // FOLD: W0 (1 + sc) W_in[e ; G]
            f32x2 q02 = {0.f, 0.f}, q12 = {0.f, 0.f};
            float rstd0, rstd1, mr0 = 0.f, mr1 = 0.f;
            if constexpr (ROT) {
                // ---- rotated statistics (dgt_pack.cpp rot_stats) ----
                // The rows are Rq = Q P (W_row h + b), Cq = Q P W_col h with Q (P W_in[:, e ; G]) = [L ; 0], L upper triangular
                // (2 De x 2 De):  D var(pre) = |L z + Rq_a[:2De] + Cq_c[:2De]|^2 + |Rq_a[2De:] + Cq_c[2De:]|^2.  The first term costs
                // a TRIANGULAR projection (160 instead of 512 MFMAs at nf 256) and half of the row gathers; the second is one number
                // per directed edge from k_node_gram.  No means anywhere: P removed them (the folded matrix and F are built from the
                // centred factors).  Shortest block first, so that the rows of the tail's first block, requested behind the longest
                // block's last weight prefetch, get its 4k cycles of cover.
                float z[2 * X::HE];
#pragma unroll
                for (int s = 0; s < X::HE; ++s) { z[s] = en[s]; z[X::HE + s] = G[s]; }
                static_for<NB2>([&](auto kc) {
                    constexpr int k = decltype(kc)::value, b = NB2 - 1 - k;
                    constexpr int KQ = 4 * (NB2 - b), KQN = 4 * (NB2 - b + 1);
                    constexpr int NXT = (k + 1 < NB2 && KQN < X::PG) ? KQN : X::PG;
                    float t0[16], t1[16];
#pragma unroll
                    for (int s = 0; s < 16; s += 2) {
                        const f32x2 a = pk2(n0[s], n0[s + 1]) + pk2(n1[s], n1[s + 1]), c = pk2(n2[s], n2[s + 1]) + pk2(n3[s], n3[s + 1]);
                        t0[s] = a.x; t0[s + 1] = a.y; t1[s] = c.x; t1[s + 1] = c.y;
                    }
                    auto next_rows = [&]() {
                        if constexpr (k + 1 < NB2) { bload16(own_r, b - 1, n0); bload16(wcol_j, b - 1, n1); bload16(wrow_j, b - 1, n2); bload16(own_c, b - 1, n3); }
                        else { bload16(ua_i, zb0, n0); bload16(ub_j, zb0, n1); bload16(ua_j, zb0, n2); bload16(ub_i, zb0, n3); }    // first block of the tail
                    };
                    const unsigned cur = oL + (unsigned)(b * KQL + 4 * b) * 1024;
                    f32x16 acc;
                    if constexpr (k + 1 < NB2) acc = mfma_block_g<KQ, NXT>(wp, ws, cur, ws, oL + (unsigned)((b - 1) * KQL + 4 * (b - 1)) * 1024, z + 16 * b, zero16(), next_rows);
                    else acc = mfma_block_g<KQ, NXT>(wp, ws, cur, wm, (unsigned)zb0 * 2 * X::KQE * 1024, z + 16 * b, zero16(), next_rows);
                    PT(3);
#pragma unroll
                    for (int s = 0; s < 16; s += 2) {
                        const f32x2 sv = pk2(acc[s], acc[s + 1]);
                        const f32x2 d0 = sv + pk2(t0[s], t0[s + 1]), d1 = sv + pk2(t1[s], t1[s + 1]);
                        q02 = __builtin_elementwise_fma(d0, d0, q02);
                        q12 = __builtin_elementwise_fma(d1, d1, q12);
                    }
                    PT(4);
                });
                // (max with 0: the Gram form |R''|^2 + |C''|^2 + 2 <R'', C''> can round a hair below zero when the variance itself is ~ 0)
                rstd0 = __builtin_amdgcn_rsqf(fmaxf((pair_sum(q02.x + q02.y) + gr0) * (1.f / D), 0.f) + 1e-6f);
                rstd1 = __builtin_amdgcn_rsqf(fmaxf((pair_sum(q12.x + q12.y) + gr1) * (1.f / D), 0.f) + 1e-6f);
            } else {
            const float m00 = A.rmean[(size_t)L.v * 2] + A.rmean[(size_t)P.u * 2 + 1];          // direction 0: a = i, c = j
            const float m01 = A.rmean[(size_t)P.u * 2] + A.rmean[(size_t)L.v * 2 + 1];          // direction 1: a = j, c = i
            f32x2 ssum2 = {0.f, 0.f};
            bload16(own_r, 0, n0); bload16(wcol_j, 0, n1); bload16(wrow_j, 0, n2); bload16(own_c, 0, n3);
#pragma unroll
            for (int b = 0; b < X::ND; ++b) {
                const unsigned we = oi + (unsigned)(b * 2 * X::KQE) * 1024, wg_ = we + X::KQE * 1024;
                float g[16], t0[16], t1[16];
                if constexpr (!FOLD) load16(qsc_ + b * 32 + half * 16, g);
#pragma unroll
                for (int s = 0; s < 16; s += 2) {
                    const f32x2 a = (pk2(n0[s], n0[s + 1]) + pk2(n1[s], n1[s + 1])) - m00, c = (pk2(n2[s], n2[s + 1]) + pk2(n3[s], n3[s + 1])) - m01;
                    t0[s] = a.x; t0[s + 1] = a.y; t1[s] = c.x; t1[s + 1] = c.y;
                }
                // the next block's rows are requested behind the last weight prefetch of this block (see mfma_block_p2)
                auto next_rows = [&]() {
                    if (b + 1 < X::ND) { bload16(own_r, b + 1, n0); bload16(wcol_j, b + 1, n1); bload16(wrow_j, b + 1, n2); bload16(own_c, b + 1, n3); }
                    else { bload16(ua_i, zb0, n0); bload16(ub_j, zb0, n1); bload16(ua_j, zb0, n2); bload16(ub_i, zb0, n3); }    // first block of the tail
                };
                // (a second weight pipe — each part of a block prefetching its own part of the next block, two groups of
                // cover — measured no gain: the blocks are bound by instruction issue, not by the wait for L2)
                f32x16 acc = mfma_block_p<X::KQE>(wp, ws, we, wg_, en, zero16());
                if (FOLD && b + 1 == X::ND) acc = mfma_block_p2<X::KQE>(wp, ws, wg_, wm, (unsigned)zb0 * 2 * X::KQE * 1024, G, acc, next_rows);
                else acc = mfma_block_p2<X::KQE>(wp, ws, wg_, ws, b + 1 < X::ND ? wg_ + X::KQE * 1024 : o0 + (unsigned)zb0 * X::KQD * 1024, G, acc, next_rows);
                PT(3);
#pragma unroll
                for (int s = 0; s < 16; s += 2) {                // element pairs on the packed fp32 pipe
                    const f32x2 sv = pk2(acc[s], acc[s + 1]);
                    ssum2 = ssum2 + sv;
                    if constexpr (!FOLD) { const f32x2 m = sv * (pk2(g[s], g[s + 1]) + 1.f); sg[b * 16 + s] = m.x; sg[b * 16 + s + 1] = m.y; }
                    const f32x2 d0 = sv + pk2(t0[s], t0[s + 1]), d1 = sv + pk2(t1[s], t1[s + 1]);
                    q02 = __builtin_elementwise_fma(d0, d0, q02);
                    q12 = __builtin_elementwise_fma(d1, d1, q12);
                }
                PT(4);
            }
            const float meanS = pair_sum(ssum2.x + ssum2.y) * (1.f / D);
            rstd0 = __builtin_amdgcn_rsqf(fmaxf(pair_sum(q02.x + q02.y) * (1.f / D) - meanS * meanS, 0.f) + 1e-6f);
            rstd1 = __builtin_amdgcn_rsqf(fmaxf(pair_sum(q12.x + q12.y) * (1.f / D) - meanS * meanS, 0.f) + 1e-6f);
            mr0 = (meanS + m00) * rstd0; mr1 = (meanS + m01) * rstd1;
            }
            PT(4);
            const float* wg_v = launder(mrow + X::M_WG);
            const float* bs_v = wg_v + D;
            f32x2 c00 = {0.f, 0.f}, c01 = c00, c02 = c00, c10 = c00, c11 = c00, c12 = c00;
#pragma unroll 1
            for (int b = zb0; b < zb1; ++b) {
                float t0[16], t1[16], wgb[16], bsb[16];
#pragma unroll
                for (int s = 0; s < 16; s += 2) {
                    const f32x2 a = pk2(n0[s], n0[s + 1]) + pk2(n1[s], n1[s + 1]), c = pk2(n2[s], n2[s + 1]) + pk2(n3[s], n3[s + 1]);
                    t0[s] = a.x; t0[s + 1] = a.y; t1[s] = c.x; t1[s + 1] = c.y;
                }
                auto next_rows = [&]() {                         // next block's rows (the last iteration re-requests its own),
                    const int bn = b + 1 < zb1 ? b + 1 : b;      // behind the last weight prefetch of this block
                    bload16(ua_i, bn, n0); bload16(ub_j, bn, n1); bload16(ua_j, bn, n2); bload16(ub_i, bn, n3);
#if JODO_X_UPD_PERS >= 2
                    if constexpr (NEXT) {                       // the next offset of this item, or the first one of the workgroup's next item
                        if (b + 1 == zb1) {
                            const bool in = t + 1 < t_end;
                            LaneNode Lq;
                            Lq.v = in ? L.v : L_c.v; Lq.b = in ? L.b : L_c.b; Lq.i = in ? L.i : L_c.i; Lq.n = in ? L.n : L_c.n;
                            Lq.noff = in ? L.noff : L_c.noff; Lq.eoff = in ? L.eoff : L_c.eoff; Lq.valid = in ? L.valid : L_c.valid;
                            request_rows(Lq, in ? t + 1 : t0_c);
                        }
                    }
#elif JODO_X_UPD_ROWS
                    if constexpr (NEXT) { if (b + 1 == zb1 && t + 1 < t_end) request_rows(L, t + 1); }
#endif
                };
                if constexpr (!ROT) load16(wg_v + b * 32 + half * 16, wgb);
                load16(bs_v + b * 32 + half * 16, bsb);
                f32x16 z;
                if constexpr (FOLD) {                            // Z = M [e ; G]: K = 2 De instead of D, and no S to keep
                    const unsigned mcur = (unsigned)b * 2 * X::KQE * 1024;
                    WSrc wn = wm;
                    unsigned noff = mcur + 2 * X::KQE * 1024;
                    if (b + 1 == zb1) { wn = ws; noff = o3; }
                    z = mfma_block_p<X::KQE>(wp, wm, mcur, mcur + X::KQE * 1024, en, zero16());
                    z = mfma_block_p2<X::KQE>(wp, wm, mcur + X::KQE * 1024, wn, noff, G, z, next_rows);
                } else {
                    const unsigned wcur = o0 + (unsigned)b * X::KQD * 1024;
                    z = mfma_block_p2<X::KQD>(wp, ws, wcur, ws, b + 1 < zb1 ? wcur + X::KQD * 1024 : o3, sg, zero16(), next_rows);
                }
                PT(5);
                // vector tail of this block, one direction after the other and eight registers at a time (fences keep the
                // compiler from evaluating all 32 SiLUs at once: their temporaries would not fit the 256 arch VGPRs)
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    float k0[8], k1[8], k2[8], ca[8];
                    auto ld8 = [&](const float* p8, float (&r)[8]) {
                        const float4 a = reinterpret_cast<const float4*>(p8)[0], c = reinterpret_cast<const float4*>(p8)[1];
                        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = c.x; r[5] = c.y; r[6] = c.z; r[7] = c.w;
                    };
                    const int fo = b * 32 + half * 16 + hq * 8;
                    const float* w2l = reinterpret_cast<const float*>(w2s);
                    ld8(w2l + fo, k0); ld8(w2l + D + fo, k1); ld8(w2l + 2 * D + fo, k2);
#pragma unroll
                    for (int s = 0; s < 8; s += 2) {               // element pairs on the packed fp32 pipe; the three dot products
                        const f32x2 bs2 = pk2(bsb[hq * 8 + s], bsb[hq * 8 + s + 1]);   // keep even / odd partial sums
                        f32x2 c = bs2;                                  // ROT: the mean is gone (centred factors)
                        if constexpr (!ROT) c = __builtin_elementwise_fma((f32x2)(-mr0), pk2(wgb[hq * 8 + s], wgb[hq * 8 + s + 1]), bs2);
                        ca[s] = c.x; ca[s + 1] = c.y;
                    }
                    pipeline_fence();
#pragma unroll
                    for (int s = 0; s < 8; s += 2) {
                        const f32x2 pre = pk2(z[hq * 8 + s], z[hq * 8 + s + 1]) + pk2(t0[hq * 8 + s], t0[hq * 8 + s + 1]);
                        const f32x2 ys0 = silu_f2(__builtin_elementwise_fma(pre, (f32x2)(rstd0), pk2(ca[s], ca[s + 1])));
                        c00 = __builtin_elementwise_fma(ys0, pk2(k0[s], k0[s + 1]), c00);
                        c01 = __builtin_elementwise_fma(ys0, pk2(k1[s], k1[s + 1]), c01);
                        c02 = __builtin_elementwise_fma(ys0, pk2(k2[s], k2[s + 1]), c02);
                    }
                    pipeline_fence();
#pragma unroll
                    for (int s = 0; s < 8; s += 2) {
                        const f32x2 bs2 = pk2(bsb[hq * 8 + s], bsb[hq * 8 + s + 1]);
                        f32x2 c = bs2;
                        if constexpr (!ROT) c = __builtin_elementwise_fma((f32x2)(-mr1), pk2(wgb[hq * 8 + s], wgb[hq * 8 + s + 1]), bs2);
                        ca[s] = c.x; ca[s + 1] = c.y;
                    }
#pragma unroll
                    for (int s = 0; s < 8; s += 2) {
                        const f32x2 pre = pk2(z[hq * 8 + s], z[hq * 8 + s + 1]) + pk2(t1[hq * 8 + s], t1[hq * 8 + s + 1]);
                        const f32x2 ys1 = silu_f2(__builtin_elementwise_fma(pre, (f32x2)(rstd1), pk2(ca[s], ca[s + 1])));
                        c10 = __builtin_elementwise_fma(ys1, pk2(k0[s], k0[s + 1]), c10);
                        c11 = __builtin_elementwise_fma(ys1, pk2(k1[s], k1[s + 1]), c11);
                        c12 = __builtin_elementwise_fma(ys1, pk2(k2[s], k2[s + 1]), c12);
                    }
                    pipeline_fence();
                }
                PT(7);
            }
            PT(7);
            if constexpr (ZW > 1) {                              // the waves' partial dot products: added by wave 0 in wave order
                float* zr = zred + lane;
                if (zw > 0) {
                    float* o = zr + (zw - 1) * 12 * 64;
                    o[0 * 64] = c00.x; o[1 * 64] = c00.y; o[2 * 64] = c01.x; o[3 * 64] = c01.y; o[4 * 64] = c02.x; o[5 * 64] = c02.y;
                    o[6 * 64] = c10.x; o[7 * 64] = c10.y; o[8 * 64] = c11.x; o[9 * 64] = c11.y; o[10 * 64] = c12.x; o[11 * 64] = c12.y;
                }
                __syncthreads();
                if (zw == 0) {
#pragma unroll
                    for (int w = 1; w < ZW; ++w) {
                        const float* o = zr + (w - 1) * 12 * 64;
                        c00.x += o[0 * 64]; c00.y += o[1 * 64]; c01.x += o[2 * 64]; c01.y += o[3 * 64]; c02.x += o[4 * 64]; c02.y += o[5 * 64];
                        c10.x += o[6 * 64]; c10.y += o[7 * 64]; c11.x += o[8 * 64]; c11.y += o[9 * 64]; c12.x += o[10 * 64]; c12.y += o[11 * 64];
                    }
                }
                if (t + 1 < t1) __syncthreads();                 // the next offset's partials may be written
            }
            const float nrm = fmaxf(sqrtf(d2), 1e-8f);
#pragma unroll
            for (int dir = 0; dir < 2; ++dir) {
                const float c0 = tanh_f(pair_sum(dir == 0 ? c00.x + c00.y : c10.x + c10.y));
                const float c1 = tanh_f(pair_sum(dir == 0 ? c01.x + c01.y : c11.x + c11.y));
                const float c2 = tanh_f(pair_sum(dir == 0 ? c02.x + c02.y : c12.x + c12.y));
                const size_t rr = dir == 0 ? P.rij : P.rji;
                const int fl = A.eflag[rr];
                const float iota = (c0 + ((fl & 1) ? c1 : 0.f) + ((fl & 2) ? c2 : 0.f)) * (1.f / 3.f);
                const float f = cscale * iota / nrm;
                const float sgn = dir == 0 ? 1.f : -1.f;          // x_a - x_c
                if (P.ok && half == 0 && zw == 0)
                    reinterpret_cast<float4*>(A.dposE)[rr] = make_float4(sgn * dx * f, sgn * dy * f, sgn * dz * f, 0.f);
            }
            PT(6);
        } else {
            // ---- S = W_e e + W_d G, shared by both directions: parked per lane ----
    #pragma unroll
            for (int b = 0; b < X::ND; ++b) {
                const unsigned we = oi + (unsigned)(b * 2 * X::KQE) * 1024, wg_ = we + X::KQE * 1024;
                f32x16 acc = mfma_block_p<X::KQE>(wp, ws, we, wg_, en, zero16());
                acc = mfma_block_p<X::KQE>(wp, ws, wg_, b + 1 < X::ND ? wg_ + X::KQE * 1024 : o0, G, acc);
    #pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if (b < X::ND - PLB) park[(b < X::ND - PLB ? b : 0) * 16 + s] = acc[s];
                }
                if (b >= X::ND - PLB) {
    #pragma unroll
                    for (int q = 0; q < 4; ++q)
                        pl[((b - (X::ND - PLB)) * 4 + q) * 64 + lane] = make_float4(acc[q * 4 + 0], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]);
                }
            }
            PT(3);
            // ---- two directed evaluations: u = S + W_row h_a + W_col h_c -> LN -> modulate -> coord_mlp ----
    #pragma unroll 1
            for (int dir = 0; dir < 2; ++dir) {
                if (dsel >= 0 && dir != dsel) continue;
                BRow ra = wrow_i, rc = wcol_j;
                if (dir == 1) { ra.voff = wrow_j.voff; rc.voff = wcol_i.voff; }
                float uu[X::HD];
                // the per-node rows are requested GB blocks at a time (buffer loads, pinned by the fence) and only
                // then consumed — see BRow in dgt_device.h
                constexpr int GB = D == 256 ? 4 : 2;          // blocks per gather group (registers are shorter at nf = 384)
    #pragma unroll
                for (int g = 0; g < X::ND / GB; ++g) {
                    float a1[GB * 16], a2[GB * 16];
    #pragma unroll
                    for (int k = 0; k < GB; ++k) {
                        float t1[16], t2[16];
                        bload16(ra, g * GB + k, t1);
                        bload16(rc, g * GB + k, t2);
    #pragma unroll
                        for (int s = 0; s < 16; ++s) { a1[k * 16 + s] = t1[s]; a2[k * 16 + s] = t2[s]; }
                    }
                    pipeline_fence();
    #pragma unroll
                    for (int k = 0; k < GB; ++k) {
                        const int b = g * GB + k;
                        float pk[16];
                        if (b < X::ND - PLB) {
    #pragma unroll
                            for (int s = 0; s < 16; ++s) pk[s] = park[(b < X::ND - PLB ? b : 0) * 16 + s];
                        } else {
    #pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 t = pl[((b - (X::ND - PLB)) * 4 + q) * 64 + lane];
                                pk[q * 4 + 0] = t.x; pk[q * 4 + 1] = t.y; pk[q * 4 + 2] = t.z; pk[q * 4 + 3] = t.w;
                            }
                        }
    #pragma unroll
                        for (int s = 0; s < 16; ++s) uu[b * 16 + s] = pk[s] + (a1[k * 16 + s] + a2[k * 16 + s]);
                    }
                }
                PT(6);
                layer_norm<X::HD>(uu);
                modulate<X::ND>(uu, qsh_, qsc_, half);
                PT(4);
                float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    #pragma unroll 1
                for (int b = 0; b < X::ND; ++b) {
                    const unsigned wcur = o0 + (unsigned)b * X::KQD * 1024;
                    const unsigned wnx = b + 1 < X::ND ? wcur + X::KQD * 1024 : ((dir == 0 && dsel < 0) ? o0 : o3);
                    float bb[16], k0[16], k1[16], k2[16];
                    load16(b0_ + b * 32 + half * 16, bb);
                    load16(w2_ + b * 32 + half * 16, k0);
                    load16(w2_ + D + b * 32 + half * 16, k1);
                    load16(w2_ + 2 * D + b * 32 + half * 16, k2);
                    f32x16 acc = mfma_block_p<X::KQD>(wp, ws, wcur, wnx, uu, zero16());
    #pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const float ys = silu_f(acc[s] + bb[s]);
                        c0 = fmaf(ys, k0[s], c0);
                        c1 = fmaf(ys, k1[s], c1);
                        c2 = fmaf(ys, k2[s], c2);
                    }
                }
                PT(dir == 0 ? 5 : 7);
                c0 = tanh_f(pair_sum(c0));
                c1 = tanh_f(pair_sum(c1));
                c2 = tanh_f(pair_sum(c2));
                const size_t rr = dir == 0 ? P.rij : P.rji;
                const int fl = A.eflag[rr];
                const float iota = (c0 + ((fl & 1) ? c1 : 0.f) + ((fl & 2) ? c2 : 0.f)) * (1.f / 3.f);
                const float nrm = fmaxf(sqrtf(d2), 1e-8f);
                const float f = cscale * iota / nrm;
                const float sgn = dir == 0 ? 1.f : -1.f;          // x_a - x_c
                if (P.ok && half == 0)
                    reinterpret_cast<float4*>(A.dposE)[rr] = make_float4(sgn * dx * f, sgn * dy * f, sgn * dz * f, 0.f);
                PT(6);
            }
        }
    }
#if JODO_X_UPD_PERS
    if (!more) break;
    }
#endif
    PT_FLUSH;
}

// ------------------------------------------------------------------------------------------------
// heads
template <int D>
__device__ __forceinline__ void node_head_body(const KArgs& A, int strip) {
    using X = Dim<D>;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int v = strip * 32 + j;
    const int KNH = A.d.KNH;
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned o1 = (unsigned)(A.wg[JW_NH1_W] * 4), o2 = (unsigned)(A.wg[JW_NH2_W] * 4), o3 = (unsigned)(A.wg[JW_NH3_W] * 4);
    WPipe<4> wp;
    wpipe_prime(wp, ws, o1);
    f32x16 o[X::ND];
#pragma unroll
    for (int b = 0; b < X::ND; ++b) o[b] = zero16();
    {
        const int kq = KNH / 8, nch = KNH / 64;
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            float x[32];
            load_nat<2>(A.ahid + (size_t)v * KNH + c * 64, half, x);
#pragma unroll
            for (int ob = 0; ob < X::ND; ++ob) {
                const unsigned cur = o1 + (unsigned)(ob * kq + c * 8) * 1024;
                const unsigned nxt = ob + 1 < X::ND ? o1 + (unsigned)((ob + 1) * kq + c * 8) * 1024
                                                    : (c + 1 < nch ? o1 + (unsigned)((c + 1) * 8) * 1024 : o2);
                o[ob] = mfma_block_p<8>(wp, ws, cur, nxt, x, o[ob]);
            }
        }
    }
    float a1[X::HD];
    {
        const float* bias = A.W + A.wg[JW_NH1_B];
#pragma unroll
        for (int b = 0; b < X::ND; ++b) {
            float r[16];
            acc_bias(o[b], bias + b * 32 + half * 16, r);
#pragma unroll
            for (int s = 0; s < 16; ++s) a1[b * 16 + s] = silu_f(r[s]);
        }
    }
    float a2[X::HD / 2];
    {
        const float* bias = A.W + A.wg[JW_NH2_B];
#pragma unroll
        for (int b = 0; b < X::ND / 2; ++b) {
            const unsigned cur = o2 + (unsigned)(b * X::KQD) * 1024;
            f32x16 acc = mfma_block_p<X::KQD>(wp, ws, cur, b + 1 < X::ND / 2 ? cur + X::KQD * 1024 : o3, a1, zero16());
            float r[16];
            acc_bias(acc, bias + b * 32 + half * 16, r);
#pragma unroll
            for (int s = 0; s < 16; ++s) a2[b * 16 + s] = silu_f(r[s]);
        }
    }
    {
        f32x16 acc = mfma_block_p<X::KQD / 2>(wp, ws, o3, o3, a2, zero16());
        float r[16];
        acc_bias(acc, A.W + A.wg[JW_NH3_B] + half * 16, r);
        store16(A.apred + (size_t)v * 32 + half * 16, r);
    }
}
template <int D>
__global__ __launch_bounds__(64, 1) void k_node_head(KArgs A) { node_head_body<D>(A, (int)blockIdx.x); }

// SYM: the edge state and the readouts of (a, c) and (c, a) are bit-identical for symmetric inputs, so the head runs once per
// unordered pair (plan list ut_rows) and writes both rows; the directed form covers asymmetric inputs (device flag).
template <int D, int NBK, bool SYM>     // NBK = KEH / 32
__device__ __forceinline__ void edge_head_body(const KArgs& A, int blk) {
    if ((A.flags[FLAG_ASYM] != 0) == SYM) return;
    using X = Dim<D>;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int ur = SYM ? A.pd.ut_rows[((size_t)blk * 32 + j) * 2] : 0, um = SYM ? A.pd.ut_rows[((size_t)blk * 32 + j) * 2 + 1] : 0;
    const size_t r = SYM ? (size_t)(ur < 0 ? 0 : ur) : (size_t)blk * 32 + j;
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned o1 = (unsigned)(A.wg[JW_EH1_W] * 4), o2 = (unsigned)(A.wg[JW_EH2_W] * 4), o3 = (unsigned)(A.wg[JW_EH3_W] * 4);
    WPipe<4> wp;
    wpipe_prime(wp, ws, o1);
    float x[NBK * 16];
    load_nat<NBK>(A.ehid + r * (NBK * 32), half, x);
    float a1[X::De];                                       // [exist De | type De] hidden, half of it per half-lane
    {
        const float* bias = A.W + A.wg[JW_EH1_B];
#pragma unroll
        for (int b = 0; b < 2 * X::NE; ++b) {
            const unsigned cur = o1 + (unsigned)(b * NBK * 4) * 1024;
            f32x16 acc = mfma_block_p<NBK * 4>(wp, ws, cur, b + 1 < 2 * X::NE ? cur + (unsigned)(NBK * 4) * 1024 : o2, x, zero16());
            float rr[16];
            acc_bias(acc, bias + b * 32 + half * 16, rr);
#pragma unroll
            for (int s = 0; s < 16; ++s) a1[b * 16 + s] = silu_f(rr[s]);
        }
    }
    float a2[X::De / 2];
    {
        const float* bias = A.W + A.wg[JW_EH2_B];
#pragma unroll
        for (int b = 0; b < X::NE; ++b) {
            const unsigned cur = o2 + (unsigned)(b * (X::De / 4)) * 1024;
            f32x16 acc = mfma_block_p<X::De / 4>(wp, ws, cur, b + 1 < X::NE ? cur + (unsigned)(X::De / 4) * 1024 : o3, a1, zero16());
            float rr[16];
            acc_bias(acc, bias + b * 32 + half * 16, rr);
#pragma unroll
            for (int s = 0; s < 16; ++s) a2[b * 16 + s] = silu_f(rr[s]);
        }
    }
    {
        f32x16 acc = mfma_block_p<X::De / 8>(wp, ws, o3, o3, a2, zero16());
        float rr[16];
        acc_bias(acc, A.W + A.wg[JW_EH3_B] + half * 16, rr);
        if (SYM) {
            if (half == 0 && ur >= 0) {
                reinterpret_cast<float4*>(A.epred)[ur] = make_float4(rr[0], rr[1], rr[2], rr[3]);
                reinterpret_cast<float4*>(A.epred)[um] = make_float4(rr[0], rr[1], rr[2], rr[3]);
            }
        } else if (half == 0 && r < (size_t)A.pd.rows) {
            reinterpret_cast<float4*>(A.epred)[r] = make_float4(rr[0], rr[1], rr[2], rr[3]);
        }
    }
}
template <int D, int NBK, bool SYM>
__global__ __launch_bounds__(64, 1) void k_edge_head(KArgs A) { edge_head_body<D, NBK, SYM>(A, (int)blockIdx.x); }

// Both heads in one launch (pinned symmetric inputs: the pair form of the edge head is known to be the one that works).  The node
// head has one long item per strip — 1 409 at QM9 B = 2500: 1.38 rounds of one-wave items, the second round leaves 639 SIMDs idle
// for ~100 us — and the edge head tens of thousands of short ones that fill them: node strips first, pair items behind.
template <int D, int NBK>
__global__ __launch_bounds__(64, 1) void k_heads_sym(KArgs A) {
    const int ns = A.pd.n_strips;
    if ((int)blockIdx.x < ns) node_head_body<D>(A, (int)blockIdx.x);
    else edge_head_body<D, NBK, true>(A, (int)blockIdx.x - ns);
}

}  // namespace wide
}  // namespace jd
