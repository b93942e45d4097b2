// Pair enumeration shared by the symmetric fast path of a DGT block (k_edge_attn pair mode, dgt_kernels_attn.h;
// wide::k_edge_update_sym, dgt_kernels_wide.h) + LDS-resident-weight helpers.
//
// With symmetric caller inputs the edge hidden state is exactly symmetric (e[a,c] == e[c,a] bit for
// bit: every per-edge operation is either symmetric in (a,c) or a per-row function).  Everything that
// depends on the edge state alone — edge_emb/LN/lin_edge0/lin_edge1 + tanh in the attention phase, the edge
// FFN, the readout, and the W_e e + W_d G part of equi_update.input_lin — is therefore computed ONCE
// per unordered pair {i, j}; only the genuinely directed pieces (q_c.k_a products, v_a, W_row h_a +
// W_col h_c, coord_mlp) are evaluated for both directions.
//
// Enumeration: lane = atom i, iteration d = 1 .. floor(n/2) pairs it with
// j = (i + d) mod n  (for even n the offset d = n/2 is taken by i < n/2 only), so every unordered
// pair is visited exactly once and all lanes of a molecule do the same number of iterations.
// Results that belong to the partner atom are handed over through LDS (attention) or written per edge row
// (positions dposE) and reduced later by that atom's own lane — no atomics, deterministic.
#pragma once
#include "dgt_kernels_common.h"

namespace jd {

// ------------------------------------------------------------------------------------------------
// LDS-resident weights: at nf = 256 the attention kernel's K = 64 projections edge_emb (32 KiB) + lin_edge0 (64 KiB) fit
// in LDS.  A workgroup loads them once; every iteration then reads its A operands with conflict-free ds_read_b128
// instead of streaming them from L2.
constexpr int WG_WAVES = 4;

template <int NQ, int NW = WG_WAVES>   // cooperative copy of NQ quads (1 KiB each) global -> LDS by NW waves
__device__ __forceinline__ void stage_weights(float4* __restrict__ dst, const float4* __restrict__ src) {
    for (int i = threadIdx.x; i < NQ * 64; i += NW * 64) dst[i] = src[i];
}

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

}  // namespace jd
