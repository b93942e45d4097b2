// Pair kernels: the symmetric fast path of a DGT block.
//
// With symmetric caller inputs the edge hidden state is exactly symmetric (e[a,c] == e[c,a] bit for
// bit: every per-edge operation is either symmetric in (a,c) or a per-row function).  Everything that
// depends on the edge state alone — edge_emb/LN/lin_edge0 + tanh in the attention scores, the edge
// FFN, the readout, and the W_e e + W_d G part of equi_update.input_lin — is therefore computed ONCE
// per unordered pair {i, j}; only the genuinely directed pieces (q_c.k_a products, W_row h_a +
// W_col h_c, coord_mlp) are evaluated for both directions.  Saves 50 % of the MFMA work of the
// scores kernel and 23 % of the update kernel.
//
// Enumeration: lane = atom i of a 32-atom strip, iteration d = 1 .. floor(n/2) pairs it with
// j = (i + d) mod n  (for even n the offset d = n/2 is taken by i < n/2 only), so every unordered
// pair is visited exactly once and all lanes of a molecule do the same number of iterations.
// Results that belong to the partner atom are written per edge row (scores S, positions dposE) and
// reduced later by that atom's own lane — still no atomics, still deterministic.
#pragma once
#include "dgt_kernels_block.h"

namespace jd {

#ifdef JODO_PHASE_TIMING
#define PT_INIT unsigned long long pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_last = __builtin_readcyclecounter();
#define PT(ph) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_readcyclecounter(); pt_acc[ph] += n_ - pt_last; pt_last = n_; } while (0)
#define PT_FLUSH do { if (A.dbgt && lane == 0 && (blockIdx.x % 61) == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&A.dbgt[i_], pt_acc[i_]); atomicAdd(&A.dbgt[15], 1ull); } } while (0)
#else
#define PT_INIT
#define PT(ph)
#define PT_FLUSH
#endif

struct PairLane {
    int j, u;          // partner index inside the molecule, packed node id
    size_t rij, rji;   // edge rows (i -> j as a=i,c=j) and (a=j,c=i)
    bool ok;
};

__device__ __forceinline__ PairLane pair_of(const LaneNode& L, int d) {
    PairLane P;
    const bool ok = L.valid && (2 * d < L.n || (2 * d == L.n && 2 * L.i < L.n));
    int j = ok ? L.i + d : (L.n > 1 ? (L.i + 1) : L.i);
    if (j >= L.n) j -= L.n;
    if (!L.valid) j = 0;
    P.ok = ok; P.j = j; P.u = L.noff + j;
    P.rij = (size_t)L.eoff + (size_t)L.i * L.n + j;
    P.rji = (size_t)L.eoff + (size_t)j * L.n + L.i;
    return P;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG_WAVES * 64, 2) void k_edge_scores_sym(KArgs A) {
    if (A.flags[FLAG_ASYM]) return;
    __shared__ float4 wl[(32 + 64) * 64];                       // edge_emb (2 x 16 quads) | lin_edge0 (8 x 8 quads)
    stage_weights<32>(wl, reinterpret_cast<const float4*>(A.W + A.wb[JB_EE_W]));
    stage_weights<64>(wl + 32 * 64, reinterpret_cast<const float4*>(A.W + A.wb[JB_LE0_W]));
    __syncthreads();
    const int lane = threadIdx.x & 63, jl = lane & 31, half = lane >> 5;
    const int it = blockIdx.x * WG_WAVES + (threadIdx.x >> 6);
    if (it >= A.pd.n_sitems) return;
    const int strip = A.pd.sitem_strip[it], t0 = A.pd.sitem_t0[it], t1 = A.pd.sitem_t1[it];
    const LaneNode L = lane_node(A, strip, jl);
    const float* mrow = mod_row(A, L.b) + A.mod_base;
    const float gscale = mrow[6 * 256 + 6 * 64 + 2 * 256 + 0], gshift = mrow[6 * 256 + 6 * 64 + 2 * 256 + 1];
    const float4 pv = reinterpret_cast<const float4*>(A.pos_out)[L.v];
    const float4* wEE = wl + lane;
    const float4* wL0 = wl + 32 * 64 + lane;
    for (int t = t0; t < t1; ++t) {
        const PairLane P = pair_of(L, t + 1);
        const float* es1 = launder(mrow + 6 * 256);
        const float* ec1 = es1 + 64;
        const float* cst = launder(A.W);
        const float* tab = cst + A.wb[JB_GBF];
        const float* bEE = cst + A.wb[JB_EE_B];
        TRow qi = trow(A.q, 8, L.v, half), ki = trow(A.k, 8, L.v, half);
        qi.p = launder(qi.p); ki.p = launder(ki.p);
        const TRow qj = trow(A.q, 8, P.u, half), kj = trow(A.k, 8, P.u, half);
        const float4 pu = reinterpret_cast<const float4*>(A.pos_out)[P.u];
        const float dx = pv.x - pu.x, dy = pv.y - pu.y, dz = pv.z - pu.z;
        float x[32];
        {
            float G[32], e[32];
            gbf64(dx * dx + dy * dy + dz * dz, gscale, gshift, tab, half, G);
            load_nat<2>(A.e + P.rij * 64, half, e);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float bb[16];
                load16(bEE + b * 32 + half * 16, bb);
                f32x16 acc = mfma_block_lds<8>(wEE + (b * 16) * 64, G, zero16());
                acc = mfma_block_lds<8>(wEE + (b * 16 + 8) * 64, e, acc);
#pragma unroll
                for (int s = 0; s < 16; ++s) x[b * 16 + s] = acc[s] + bb[s];
            }
        }
        layer_norm<32>(x);
        modulate<2>(x, es1, ec1, half);
        if (P.ok) {
            store_nat<2>(A.et + P.rij * 64, half, x);
            store_nat<2>(A.et + P.rji * 64, half, x);
        }
        // tanh(lin_edge0) once; direction 1 = edge (j -> i): q_i . k_j ; direction 2 = edge (i -> j): q_j . k_i
        float m1[7], m2[7];
        float qin[16], kin[16], qjn[16], kjn[16];
        load16T(qi, 0, qin); load16T(ki, 0, kin);
        load16T(qj, 0, qjn); load16T(kj, 0, kjn);
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            float a1[16], a2[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) { a1[s] = qin[s] * kjn[s]; a2[s] = qjn[s] * kin[s]; }
            load16T(qi, b + 1, qin); load16T(ki, b + 1, kin);
            load16T(qj, b + 1, qjn); load16T(kj, b + 1, kjn);
            f32x16 acc = mfma_block_lds<8>(wL0 + (b * 8) * 64, x, zero16());
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
            f32x16 acc = mfma_block_lds<8>(wL0 + (7 * 8) * 64, x, zero16());
#pragma unroll
            for (int g = 0; g < 14; ++g) {
                const float tt = tanh_f(acc[g]);
                tl1[g] = tt * qin[g] * kjn[g];
                tl2[g] = tt * qjn[g] * kin[g];
            }
        }
        const int f1 = A.eflag[P.rji], f2 = A.eflag[P.rij];
        float Sg1[14], Sg2[14];
#pragma unroll
        for (int g = 0; g < 14; ++g) {
            const float o1 = ((g & 1) == half) ? m1[g >> 1] : 0.f;
            const float o2 = ((g & 1) == half) ? m2[g >> 1] : 0.f;
            Sg1[g] = pair_sum(o1 + tl1[g]) * 0.25f;
            Sg2[g] = pair_sum(o2 + tl2[g]) * 0.25f;
        }
        float S1[8], S2[8];                                // slot b of this half = head 2b + half
        S1[0] = half == 0 ? ((f1 & 1) ? 1.f : -1e10f) : ((f1 & 2) ? 1.f : -1e10f);
        S2[0] = half == 0 ? ((f2 & 1) ? 1.f : -1e10f) : ((f2 & 2) ? 1.f : -1e10f);
#pragma unroll
        for (int b = 1; b < 8; ++b) {
            S1[b] = half == 0 ? Sg1[2 * (b - 1)] : Sg1[2 * (b - 1) + 1];
            S2[b] = half == 0 ? Sg2[2 * (b - 1)] : Sg2[2 * (b - 1) + 1];
        }
        if (P.ok) {
            float4* sp = reinterpret_cast<float4*>(A.S + P.rji * 16 + half * 8);      // edge (j -> i)
            sp[0] = make_float4(S1[0], S1[1], S1[2], S1[3]);
            sp[1] = make_float4(S1[4], S1[5], S1[6], S1[7]);
            float4* sq = reinterpret_cast<float4*>(A.S + P.rij * 16 + half * 8);      // edge (i -> j)
            sq[0] = make_float4(S2[0], S2[1], S2[2], S2[3]);
            sq[1] = make_float4(S2[4], S2[5], S2[6], S2[7]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(64, 1) void k_edge_update_sym(KArgs A) {
    if (A.flags[FLAG_ASYM]) return;
    const int lane = threadIdx.x & 63, jl = lane & 31, half = lane >> 5;
    const int it = blockIdx.x;
    const int strip = A.pd.pitem_strip[it], t0 = A.pd.pitem_t0[it], t1 = A.pd.pitem_t1[it];
    const LaneNode L = lane_node(A, strip, jl);
    const float* mrow = mod_row(A, L.b) + A.mod_base;
    const float* eg1 = mrow + 6 * 256 + 2 * 64;
    const float* qsh = mrow + 6 * 256 + 6 * 64;
    const float gscale = mrow[6 * 256 + 6 * 64 + 2 * 256 + 0], gshift = mrow[6 * 256 + 6 * 64 + 2 * 256 + 1];
    const float4 pv = reinterpret_cast<const float4*>(A.pos_out)[L.v];
    const float cscale = A.W[A.wb[JB_CSCALE]];
    const WSrc ws = make_wsrc(A.W, lane);
    const unsigned o3 = (unsigned)(A.wb[JB_FF3_W] * 4), o4 = (unsigned)(A.wb[JB_FF4_W] * 4);
    const unsigned oro = (unsigned)(A.wb[JB_ERO_W] * 4), oi = (unsigned)(A.wb[JB_INE_W] * 4), o0 = (unsigned)(A.wb[JB_C0_W] * 4);
    constexpr int KQ4 = R * 64 / 8;
    WPipe<8> wp;
    wpipe_prime(wp, ws, o3);
    __shared__ float4 pre1[32 * 64];                           // u of direction 1 (pre-LayerNorm) of this wave, [quad][lane]
    PT_INIT
    for (int t = t0; t < t1; ++t) {
        const PairLane P = pair_of(L, t + 1);
        const float* eg1_ = launder(eg1);
        const float* es2_ = eg1_ + 64, *ec2_ = es2_ + 64, *eg2_ = ec2_ + 64;
        const float* qsh_ = launder(qsh);
        const float* qsc_ = qsh_ + 256;
        const float* cst = launder(A.W);
        const float* n2bias_ = cst + A.wb[JB_N2E_B], *b3_ = cst + A.wb[JB_FF3_B], *b4_ = cst + A.wb[JB_FF4_B];
        const float* b0_ = cst + A.wb[JB_C0_B], *w2_ = cst + A.wb[JB_C2_W], *tab_ = cst + A.wb[JB_GBF];
        const float* bro_ = cst + A.wb[JB_ERO_B];
        TRow wrow_i = trow(A.wrow, 8, L.v, half), wcol_i = trow(A.wcol, 8, L.v, half);
        wrow_i.p = launder(wrow_i.p); wcol_i.p = launder(wcol_i.p);
        const TRow wrow_j = trow(A.wrow, 8, P.u, half), wcol_j = trow(A.wcol, 8, P.u, half);
        const float4 pu = reinterpret_cast<const float4*>(A.pos_out)[P.u];
        const float dx = pv.x - pu.x, dy = pv.y - pu.y, dz = pv.z - pu.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        float G[32];
        gbf64(d2, gscale, gshift, tab_, half, G);
        // ---- edge residual + LN2 + modulate (symmetric) ----
        float en[32];
        {
            float e[32], n2a[32], n2c[32];
            load_nat<2>(A.e + P.rij * 64, half, e);
            {
                const TRow ra = trow(A.n2e, 2, L.v, half), rc = trow(A.n2e, 2, P.u, half);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float ta[16], tc2[16];
                    load16T(ra, b, ta);
                    load16T(rc, b, tc2);
#pragma unroll
                    for (int s = 0; s < 16; ++s) { n2a[b * 16 + s] = ta[s]; n2c[b * 16 + s] = tc2[s]; }
                }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float g[16], bb[16];
                load16(eg1_ + b * 32 + half * 16, g);
                load16(n2bias_ + b * 32 + half * 16, bb);
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    en[b * 16 + s] = fmaf(g[s], n2a[b * 16 + s] + n2c[b * 16 + s] + bb[s], e[b * 16 + s]);
            }
        }
        layer_norm<32>(en);
        modulate<2>(en, es2_, ec2_, half);
        PT(0);
        // ---- edge FFN ----
        {
            f32x16 o[2] = {zero16(), zero16()};
            float ob4[32], og2[32];
#pragma unroll
            for (int c = 0; c < R; ++c) {
                float hid[32];
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    const unsigned wcur = o3 + (unsigned)(c * 2 + b2) * 8 * 1024;
                    const unsigned wnx = b2 == 0 ? wcur + 8 * 1024 : o4 + (unsigned)(c * 8) * 1024;
                    float bb[16];
                    load16(b3_ + (c * 2 + b2) * 32 + half * 16, bb);
                    f32x16 acc = mfma_block_p<8>(wp, ws, wcur, wnx, en, zero16());
#pragma unroll
                    for (int s = 0; s < 16; ++s) hid[b2 * 16 + s] = silu_f(acc[s] + bb[s]);
                }
                if (c == R - 1) {
                    load_nat<2>(b4_, half, ob4);
                    load_nat<2>(eg2_, half, og2);
                }
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) {
                    const unsigned wcur = o4 + (unsigned)(ob * KQ4 + c * 8) * 1024;
                    const unsigned wnx = ob == 0 ? o4 + (unsigned)(KQ4 + c * 8) * 1024
                                                 : (c + 1 < R ? o3 + (unsigned)((c + 1) * 2) * 8 * 1024 : oro);
                    o[ob] = mfma_block_p<8>(wp, ws, wcur, wnx, hid, o[ob]);
                }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    en[b * 16 + s] = fmaf(og2[b * 16 + s], o[b][s] + ob4[b * 16 + s], en[b * 16 + s]);
        }
        if (P.ok) {
            store_nat<2>(A.e + P.rij * 64, half, en);
            store_nat<2>(A.e + P.rji * 64, half, en);
        }
        PT(1);
        // ---- readout ----
        {
            float bb[16];
            load16(bro_ + half * 16, bb);
            f32x16 acc = mfma_block_p<8>(wp, ws, oro, oi, en, zero16());
            float rr[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) rr[s] = acc[s] + bb[s];
            if (P.ok && half == 0) {
                store16(A.ehid + P.rij * A.d.KEH + 64 + A.layer * 16, rr);
                store16(A.ehid + P.rji * A.d.KEH + 64 + A.layer * 16, rr);
            }
        }
        PT(2);
        // ---- symmetric part of input_lin: S = W_e e + W_d G, shared by both directions.  The per-node terms
        //      W_row h_a + W_col h_c of BOTH directions are gathered under these MFMAs; as soon as block b of S
        //      is complete, u(dir 0) = S + (W_row h_i + W_col h_j) replaces the gathered terms in registers and
        //      u(dir 1) = S + (W_row h_j + W_col h_i) is parked in this wave's LDS slab ----
        float uu0[128];
        {
            f32x16 U[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned wcur = oi + (unsigned)(b * 16) * 1024;
                const unsigned wnx = b < 7 ? oi + (unsigned)((b + 1) * 16) * 1024 : oi + 8u * 1024;
                float a1[16], a2[16];
                load16T(wrow_i, b, a1);
                load16T(wcol_j, b, a2);
                U[b] = mfma_block_p<8>(wp, ws, wcur, wnx, en, zero16());
#pragma unroll
                for (int s = 0; s < 16; ++s) uu0[b * 16 + s] = a1[s] + a2[s];
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned wcur = oi + (unsigned)(b * 16 + 8) * 1024;
                const unsigned wnx = b < 7 ? oi + (unsigned)((b + 1) * 16 + 8) * 1024 : o3;
                U[b] = mfma_block_p<8>(wp, ws, wcur, wnx, G, U[b]);
#pragma unroll
                for (int q = 0; q < 4; ++q)          // S parked for direction 1 (its per-node terms are added later)
                    pre1[(b * 4 + q) * 64 + lane] = make_float4(U[b][q * 4 + 0], U[b][q * 4 + 1], U[b][q * 4 + 2], U[b][q * 4 + 3]);
#pragma unroll
                for (int s = 0; s < 16; ++s) uu0[b * 16 + s] += U[b][s];
            }
        }
        PT(3);
        // ---- two directed evaluations of LN -> modulate -> coord_mlp ----
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            float uu[128];
            if (dir == 0) {
#pragma unroll
                for (int s = 0; s < 128; ++s) uu[s] = uu0[s];
            } else {
#pragma unroll
                for (int b = 0; b < 8; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 t = pre1[(b * 4 + q) * 64 + lane];
                        uu[b * 16 + q * 4 + 0] = t.x;
                        uu[b * 16 + q * 4 + 1] = t.y;
                        uu[b * 16 + q * 4 + 2] = t.z;
                        uu[b * 16 + q * 4 + 3] = t.w;
                    }
            }
            layer_norm<128>(uu);
            modulate<8>(uu, qsh_, qsc_, half);
            PT(4);
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
            // coord_mlp.0 runs on its own, deeper weight pipe (16 quads = 4096 MFMA cycles of cover): the
            // per-node gathers issued in this loop come from HBM/MALL (~2 us) and, because vmcnt retires in
            // order, would otherwise stall the weight stream once per block
            WPipe<16> wc;
            wpipe_prime(wc, ws, o0);
#pragma unroll 1
            for (int b = 0; b < 8; ++b) {
                const unsigned wcur = o0 + (unsigned)b * 32 * 1024;
                const unsigned wnx = b < 7 ? wcur + 32 * 1024 : o0;
                float bb[16], k0[16], k1[16], k2[16];
                load16(b0_ + b * 32 + half * 16, bb);
                load16(w2_ + b * 32 + half * 16, k0);
                load16(w2_ + 256 + b * 32 + half * 16, k1);
                load16(w2_ + 512 + b * 32 + half * 16, k2);
                float a1[16], a2[16];
                if (dir == 0) {                          // direction 1's W_row h_j + W_col h_i, hidden under this block
                    load16T(wrow_j, b, a1);
                    load16T(wcol_i, b, a2);
                }
                f32x16 acc = mfma_block_p<32>(wc, ws, wcur, wnx, uu, zero16());
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float ys = silu_f(acc[s] + bb[s]);
                    c0 = fmaf(ys, k0[s], c0);
                    c1 = fmaf(ys, k1[s], c1);
                    c2 = fmaf(ys, k2[s], c2);
                }
                if (dir == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 t = pre1[(b * 4 + q) * 64 + lane];
                        pre1[(b * 4 + q) * 64 + lane] = make_float4(t.x + (a1[q * 4 + 0] + a2[q * 4 + 0]), t.y + (a1[q * 4 + 1] + a2[q * 4 + 1]),
                                                                    t.z + (a1[q * 4 + 2] + a2[q * 4 + 2]), t.w + (a1[q * 4 + 3] + a2[q * 4 + 3]));
                    }
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
    PT_FLUSH;
}

}  // namespace jd
