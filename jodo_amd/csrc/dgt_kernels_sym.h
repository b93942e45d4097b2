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
constexpr int SYM_WAVES = 4;      // one wave per SIMD with the full register file: no spills, gathers pinned ahead

// LDS weights with the read of quad q + 1 pinned ahead of the MFMAs of quad q: with a single wave per SIMD
// nothing else hides the ds_read latency (hipcc emits ds_read -> s_waitcnt lgkmcnt(0) -> 4 MFMAs otherwise)
template <int KQ>
__device__ __forceinline__ f32x16 mfma_block_lds_p(const float4* wl, const float (&act)[KQ * 4], f32x16 acc) {
    float4 a = wl[0];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        float4 nx = a;
        if (q + 1 < KQ) nx = wl[(q + 1) * 64];
        pipeline_fence();
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, act[4 * q + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, act[4 * q + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, act[4 * q + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, act[4 * q + 3], acc, 0, 0, 0);
        a = nx;
    }
    return acc;
}

__global__ __launch_bounds__(SYM_WAVES * 64, 1) void k_edge_scores_sym(KArgs A) {
    if (A.flags[FLAG_ASYM]) return;
    __shared__ float4 wl[(32 + 64) * 64];                       // edge_emb (2 x 16 quads) | lin_edge0 (8 x 8 quads)
    stage_weights<32, SYM_WAVES>(wl, reinterpret_cast<const float4*>(A.W + A.wb[JB_EE_W]));
    stage_weights<64, SYM_WAVES>(wl + 32 * 64, reinterpret_cast<const float4*>(A.W + A.wb[JB_LE0_W]));
    __syncthreads();
    const int lane = threadIdx.x & 63, jl = lane & 31, half = lane >> 5;
    const int it = blockIdx.x * SYM_WAVES + (threadIdx.x >> 6);
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
        const BRow qi = brow(A.q, 8, L.v, half), ki = brow(A.k, 8, L.v, half);
        const BRow qj = brow(A.q, 8, P.u, half), kj = brow(A.k, 8, P.u, half);
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
                f32x16 acc = mfma_block_lds_p<8>(wEE + (b * 16) * 64, G, zero16());
                acc = mfma_block_lds_p<8>(wEE + (b * 16 + 8) * 64, e, acc);
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
        bload16(qi, 0, qin); bload16(ki, 0, kin);
        bload16(qj, 0, qjn); bload16(kj, 0, kjn);
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            float a1[16], a2[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) { a1[s] = qin[s] * kjn[s]; a2[s] = qjn[s] * kin[s]; }
            bload16(qi, b + 1, qin); bload16(ki, b + 1, kin);
            bload16(qj, b + 1, qjn); bload16(kj, b + 1, kjn);
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

// The pair variant of the update kernel is width-generic: wide::k_edge_update_sym<D, R> in dgt_kernels_wide.h
// (an earlier nf = 256-only version that parked direction 1 in an LDS slab measured 6 % slower and was removed).

}  // namespace jd
