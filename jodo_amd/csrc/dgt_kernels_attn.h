// Fused attention edge phase of a DGT block (TransMixLayer.forward / message, models/layers.py:131-186, fed by the
// block's edge_emb + LN1 + modulate, models/mol_gnn.py:284-297):
//
//     et   = LN(edge_emb([GBF(|x_a - x_c|^2) ; e]))(1 + ec1) + es1                       once per unordered pair
//     T0   = tanh(lin_edge0 et),  T1 = tanh(lin_edge1 et)                                once per unordered pair
//     S    = sum_ch q_c k_a T0 / sqrt(C)  (+ the two adjacency heads, 0 -> -1e10)        both directions
//     hhat_c = sum_a softmax_a(S)[a,c] * v_a * T1                                        both directions
//
// in ONE kernel: nothing per-edge is written to HBM (round 1 stored et and the scores — 480 MB per block at QM9
// B = 2500 — and read them back twice, k_edge_scores_sym -> k_softmax -> k_edge_msgs).  The softmax is carried
// flash-style: every lane owns one target atom and keeps a running (max, sum, 128 message accumulators) for each of
// its 8 heads; k_node_post merges the partials of the items that shared a target.
//
// Pair mode (symmetric edge inputs, i.e. sampling).  The edge state is exactly symmetric, so lane i evaluates the pair
// {i, j = (i + d) mod n} once (circulant walk d = 1 .. n/2, dgt_kernels_sym.h).  Target i's new source is j, computed
// in place.  Target j's new source is i: its score and its unweighted message v_i * T1 are handed to j's lane through
// LDS — the sender writes its own slot, the receiver reads the slot of lane (i - d) mod n.  That needs both atoms of a
// pair in one workgroup, hence the plan's groups: whole molecules, up to 128 lanes, one 4-wave workgroup
// (dgt_plan.cpp).  Nine workgroup barriers per offset (one for the scores, one per 32-feature message block, double
// buffered); all waves of an item run the same offsets, so the barriers are cheap.
//
// Directed mode (asymmetric caller inputs; molecules larger than a group): lane = target, the wave visits every
// source itself — no hand-over, no barriers, twice the matrix work per edge.
//
// Weights: edge_emb + lin_edge0 (96 KiB) live in LDS for the whole item, lin_edge1 (64 KiB) streams from L2 through
// the software-pipelined ring (dgt_device.h).  40 KiB of LDS carry the hand-over buffers.
#pragma once
#include "dgt_kernels_sym.h"

namespace jd {

constexpr int ATT_WAVES = 4;
constexpr int ATT_LANES = 128;
constexpr float ATT_NEG = -3.0e38f;                    // "no source yet": exp(ATT_NEG - m) == 0 for every real m

struct AttnLane {                                      // softmax + message state of one target atom (this half's 8 heads)
    float m[8], l[8];
    float acc[128];
};

// scores of one pair for both directions from T0 = tanh(lin_edge0 x): S1 = edge (j -> i), S2 = edge (i -> j), in the
// slot order of this half (slot 0 = adjacency head `half`, slot b = learned head 2(b-1) + half)
__device__ __forceinline__ void attn_scores(const float4* wL0, const float (&x)[32], const BRow& qi, const BRow& ki,
                                            const BRow& qj, const BRow& kj, int half, int f1, int f2, bool both,
                                            float (&S1)[8], float (&S2)[8]) {
    float m1[7], m2[7];
    float qin[16], kin[16], qjn[16], kjn[16];
    bload16(qi, 0, qin); bload16(kj, 0, kjn);
    if (both) { bload16(qj, 0, qjn); bload16(ki, 0, kin); }
#pragma unroll
    for (int b = 0; b < 7; ++b) {
        float a1[16], a2[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) { a1[s] = qin[s] * kjn[s]; a2[s] = both ? qjn[s] * kin[s] : 0.f; }
        bload16(qi, b + 1, qin); bload16(kj, b + 1, kjn);
        if (both) { bload16(qj, b + 1, qjn); bload16(ki, b + 1, kin); }
        pipeline_fence();
        f32x16 acc = mfma_block_lds_p<8>(wL0 + (b * 8) * 64, x, zero16());
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float tt = tanh_f(acc[s]);
            s1 = fmaf(tt, a1[s], s1);
            s2 = fmaf(tt, a2[s], s2);
        }
        m1[b] = s1; m2[b] = s2;
        pipeline_fence();
    }
    float tl1[14], tl2[14];
    {
        f32x16 acc = mfma_block_lds_p<8>(wL0 + (7 * 8) * 64, x, zero16());
#pragma unroll
        for (int g = 0; g < 14; ++g) {
            const float tt = tanh_f(acc[g]);
            tl1[g] = tt * qin[g] * kjn[g];
            tl2[g] = both ? tt * qjn[g] * kin[g] : 0.f;
        }
    }
    float Sg1[14], Sg2[14];
#pragma unroll
    for (int g = 0; g < 14; ++g) {
        const float o1 = ((g & 1) == half) ? m1[g >> 1] : 0.f;
        const float o2 = ((g & 1) == half) ? m2[g >> 1] : 0.f;
        Sg1[g] = pair_sum(o1 + tl1[g]) * 0.25f;                                // / sqrt(out_channels = 16), layers.py:167
        Sg2[g] = both ? pair_sum(o2 + tl2[g]) * 0.25f : 0.f;
    }
    S1[0] = half == 0 ? ((f1 & 1) ? 1.f : -1e10f) : ((f1 & 2) ? 1.f : -1e10f);   // extra heads, 0 -> -1e10 (layers.py:170-174)
    S2[0] = half == 0 ? ((f2 & 1) ? 1.f : -1e10f) : ((f2 & 2) ? 1.f : -1e10f);
#pragma unroll
    for (int b = 1; b < 8; ++b) {
        S1[b] = half == 0 ? Sg1[2 * (b - 1)] : Sg1[2 * (b - 1) + 1];
        S2[b] = half == 0 ? Sg2[2 * (b - 1)] : Sg2[2 * (b - 1) + 1];
    }
}

// et of an edge row from its state e and squared length d2 (GBF -> edge_emb -> LN1 -> modulate)
__device__ __forceinline__ void attn_edge_input(const KArgs& A, const float4* wEE, const float* erow, float d2, float gscale,
                                                float gshift, const float* mrow, int half, float (&x)[32]) {
    const float* es1 = launder(mrow + 6 * 256);
    const float* ec1 = es1 + 64;
    const float* cst = launder(A.W);
    const float* tab = cst + A.wb[JB_GBF];
    const float* bEE = cst + A.wb[JB_EE_B];
    float G[32], e[32];
    gbf64(d2, gscale, gshift, tab, half, G);
    load_nat<2>(erow, half, e);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        float bb[16];
        load16(bEE + b * 32 + half * 16, bb);
        f32x16 acc = mfma_block_lds_p<8>(wEE + (b * 16) * 64, G, zero16());
        acc = mfma_block_lds_p<8>(wEE + (b * 16 + 8) * 64, e, acc);
#pragma unroll
        for (int s = 0; s < 16; ++s) x[b * 16 + s] = acc[s] + bb[s];
    }
    layer_norm<32>(x);
    modulate<2>(x, es1, ec1, half);
}

// PAIR = true: pair-mode items (ai_*), exits when the inputs are asymmetric; PAIR = false: directed-mode items (ad_*),
// runs when the inputs are asymmetric or the item belongs to a molecule that spans several groups
template <bool PAIR>
__global__ __launch_bounds__(ATT_WAVES * 64, 1) void k_edge_attn(KArgs A) {
    const int it = blockIdx.x;
    const bool asym = A.flags[FLAG_ASYM] != 0;
    if (PAIR ? asym : !(asym || A.pd.ad_big[it])) return;
    __shared__ float4 wl[(32 + 64) * 64];               // edge_emb (2 x 16 quads) | lin_edge0 (8 x 8 quads)
    __shared__ float4 sx[PAIR ? 2 * 256 : 1];           // scores handed to the partner: [quad][half * 128 + lane]
    __shared__ float4 ux[PAIR ? 2 * 4 * 256 : 1];       // unweighted messages of one 32-feature block, double buffered
    stage_weights<32, ATT_WAVES>(wl, reinterpret_cast<const float4*>(A.W + A.wb[JB_EE_W]));
    stage_weights<64, ATT_WAVES>(wl + 32 * 64, reinterpret_cast<const float4*>(A.W + A.wb[JB_LE0_W]));
    __syncthreads();
    const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
    const int ln = wave * 32 + (lane & 31);             // lane of the group
    const int grp = PAIR ? A.pd.ai_group[it] : A.pd.ad_group[it];
    const int t0 = PAIR ? A.pd.ai_t0[it] : A.pd.ad_t0[it], t1 = PAIR ? A.pd.ai_t1[it] : A.pd.ad_t1[it];
    const int part = PAIR ? A.pd.ai_part[it] : A.pd.ad_part[it];
    const int vraw = A.pd.ag_node[grp * ATT_LANES + ln];
    LaneNode L;
    L.valid = vraw >= 0;
    L.v = L.valid ? vraw : 0;
    L.b = A.pd.node_b[L.v]; L.i = A.pd.node_i[L.v]; L.n = L.valid ? A.pd.node_n[L.v] : 0;
    L.noff = A.pd.node_noff[L.v]; L.eoff = A.pd.node_eoff[L.v];
    const int lbase = ln - L.i;                         // group lane of atom 0 of this lane's molecule (pair mode)
    const float* mrow = mod_row(A, L.b) + A.mod_base;
    const float gscale = mrow[6 * 256 + 6 * 64 + 2 * 256 + 0], gshift = mrow[6 * 256 + 6 * 64 + 2 * 256 + 1];
    const float4 pv = reinterpret_cast<const float4*>(A.pos_out)[L.v];
    const float4* wEE = wl + lane;
    const float4* wL0 = wl + 32 * 64 + lane;
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned oL1 = (unsigned)(A.wb[JB_LE1_W] * 4);
    WPipe<8> wp;
    wpipe_prime(wp, ws, oL1);
    AttnLane st;
#pragma unroll
    for (int b = 0; b < 8; ++b) { st.m[b] = ATT_NEG; st.l[b] = 0.f; }
#pragma unroll
    for (int s = 0; s < 128; ++s) st.acc[s] = 0.f;
    const int slot = half * ATT_LANES + ln;             // this lane's slot in the hand-over buffers
    for (int t = t0; t < t1; ++t) {
        // ---- who is the source ----
        bool ok, rok = false;
        int u, rslot = slot;
        size_t r_in, r_out = 0;                         // edge rows: (source -> this target), (this atom -> partner)
        if (PAIR) {
            const PairLane P = pair_of(L, t + 1);
            ok = P.ok; u = P.u; r_in = P.rji; r_out = P.rij;
            const int d = t + 1;
            int rr = L.i - d;
            if (rr < 0) rr += L.n;
            rok = L.valid && L.n > 1 && (2 * d < L.n || (2 * d == L.n && 2 * rr < L.n));   // did lane (i - d) evaluate {i - d, i}?
            rslot = half * ATT_LANES + (rok ? lbase + rr : ln);
        } else {
            const bool inr = L.valid && t < L.n;
            ok = inr && t != L.i;
            const int tc = inr ? t : 0;
            u = L.noff + tc;
            r_in = (size_t)L.eoff + (size_t)tc * L.n + L.i;            // edge (source a = t) -> (target c = i)
        }
        const float4 pu = reinterpret_cast<const float4*>(A.pos_out)[u];
        const float dx = pv.x - pu.x, dy = pv.y - pu.y, dz = pv.z - pu.z;
        float x[32];
        // pair mode reads row (i, j) like the other pair kernels (the state is symmetric); directed mode the true row
        attn_edge_input(A, wEE, A.e + (PAIR ? r_out : r_in) * 64, dx * dx + dy * dy + dz * dz, gscale, gshift, mrow, half, x);
        // ---- scores ----
        const BRow qi = brow(A.q, 8, L.v, half), ki = brow(A.k, 8, L.v, half);
        const BRow qj = brow(A.q, 8, u, half), kj = brow(A.k, 8, u, half);
        float S1[8], S2[8];
        attn_scores(wL0, x, qi, ki, qj, kj, half, A.eflag[r_in], PAIR ? A.eflag[r_out] : 0, PAIR, S1, S2);
        float R[8];
        if (PAIR) {
            sx[slot] = make_float4(S2[0], S2[1], S2[2], S2[3]);
            sx[256 + slot] = make_float4(S2[4], S2[5], S2[6], S2[7]);
            __syncthreads();
            const float4 a = sx[rslot], b = sx[256 + rslot];
            R[0] = a.x; R[1] = a.y; R[2] = a.z; R[3] = a.w; R[4] = b.x; R[5] = b.y; R[6] = b.z; R[7] = b.w;
        }
        // ---- running softmax of this target: up to two new sources ----
        float sc[8], p1[8], p2[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float mn = st.m[b];
            if (ok) mn = fmaxf(mn, S1[b]);
            if (PAIR && rok) mn = fmaxf(mn, R[b]);
            sc[b] = fast_exp(st.m[b] - mn);
            p1[b] = ok ? fast_exp(S1[b] - mn) : 0.f;
            p2[b] = (PAIR && rok) ? fast_exp(R[b] - mn) : 0.f;
            st.l[b] = fmaf(st.l[b], sc[b], p1[b] + p2[b]);
            st.m[b] = mn;
        }
        // ---- messages: T1 = tanh(lin_edge1 x) once; own direction v_j * T1, partner's direction v_i * T1 ----
        const BRow vj = brow(A.v, 8, u, half), vi = brow(A.v, 8, L.v, half);
        float vjn[16], vin[16];
        bload16(vj, 0, vjn);
        if (PAIR) bload16(vi, 0, vin);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float vv[16], vo[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) { vv[s] = vjn[s]; vo[s] = PAIR ? vin[s] : 0.f; }
            if (b < 7) {
                bload16(vj, b + 1, vjn);
                if (PAIR) bload16(vi, b + 1, vin);
            }
            const unsigned cur = oL1 + (unsigned)(b * 8) * 1024;
            f32x16 acc = mfma_block_p<8>(wp, ws, cur, b < 7 ? cur + 8 * 1024 : oL1, x, zero16());
            float T[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) T[s] = tanh_f(acc[s]);
            float um[16];
            if (PAIR) {
                float4* ub = ux + (b & 1) * 4 * 256;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    ub[q * 256 + slot] = make_float4(T[q * 4 + 0] * vo[q * 4 + 0], T[q * 4 + 1] * vo[q * 4 + 1],
                                                     T[q * 4 + 2] * vo[q * 4 + 2], T[q * 4 + 3] * vo[q * 4 + 3]);
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = ub[q * 256 + rslot];
                    um[q * 4 + 0] = w.x; um[q * 4 + 1] = w.y; um[q * 4 + 2] = w.z; um[q * 4 + 3] = w.w;
                }
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float a = fmaf(st.acc[b * 16 + s], sc[b], p1[b] * (T[s] * vv[s]));
                if (PAIR) a = fmaf(p2[b], um[s], a);
                st.acc[b * 16 + s] = a;
            }
        }
    }
    // ---- partial of this item: unnormalised sums + (max, sum) per head ----
    if (L.valid) {
        store_nat<8>(A.hhat + ((size_t)L.v * A.pd.amax_parts + part) * 256, half, st.acc);
        float4* sp = reinterpret_cast<float4*>(A.astat + ((size_t)L.v * A.pd.amax_parts + part) * 32 + half * 16);
        sp[0] = make_float4(st.m[0], st.m[1], st.m[2], st.m[3]);
        sp[1] = make_float4(st.m[4], st.m[5], st.m[6], st.m[7]);
        sp[2] = make_float4(st.l[0], st.l[1], st.l[2], st.l[3]);
        sp[3] = make_float4(st.l[4], st.l[5], st.l[6], st.l[7]);
    }
}

// merge the attention partials of a node (k_node_post*): hhat = sum_p acc_p e^{m_p - M} / (sum_p l_p e^{m_p - M} + 1e-16),
// fixed order.  hh: this half-lane's 128 message features (block b = head 2b + half).
__device__ __forceinline__ void attn_merge(const KArgs& A, int v, int half, int p0, int pstep, float (&hh)[128], float (&M)[8],
                                           float (&Ls)[8]) {
    const int parts = A.pd.anode_parts[v];
    const float* sbase = A.astat + (size_t)v * A.pd.amax_parts * 32 + half * 16;
    const float* hbase = A.hhat + (size_t)v * A.pd.amax_parts * 256;
#pragma unroll
    for (int b = 0; b < 8; ++b) { M[b] = ATT_NEG; Ls[b] = 0.f; }
    for (int q = p0; q < parts; q += pstep) {
        float m[8];
        const float4* sp = reinterpret_cast<const float4*>(sbase + (size_t)q * 32);
        const float4 a = sp[0], c = sp[1];
        m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = c.x; m[5] = c.y; m[6] = c.z; m[7] = c.w;
#pragma unroll
        for (int b = 0; b < 8; ++b) M[b] = fmaxf(M[b], m[b]);
    }
#pragma unroll
    for (int s = 0; s < 128; ++s) hh[s] = 0.f;
    for (int q = p0; q < parts; q += pstep) {
        const float4* sp = reinterpret_cast<const float4*>(sbase + (size_t)q * 32);
        const float4 a = sp[0], c = sp[1], la = sp[2], lc = sp[3];
        const float m[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        const float l[8] = {la.x, la.y, la.z, la.w, lc.x, lc.y, lc.z, lc.w};
        float w[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) { w[b] = fast_exp(m[b] - M[b]); Ls[b] = fmaf(l[b], w[b], Ls[b]); }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float t[16];
            load16(hbase + (size_t)q * 256 + b * 32 + half * 16, t);
#pragma unroll
            for (int s = 0; s < 16; ++s) hh[b * 16 + s] = fmaf(t[s], w[b], hh[b * 16 + s]);
        }
    }
}
__device__ __forceinline__ void attn_normalise(float (&hh)[128], const float (&Ls)[8]) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const float inv = 1.f / (Ls[b] + 1e-16f);              // layers.py:178 (PyG softmax: e / (sum + 1e-16))
#pragma unroll
        for (int s = 0; s < 16; ++s) hh[b * 16 + s] *= inv;
    }
}

}  // namespace jd
