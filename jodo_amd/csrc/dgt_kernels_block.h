// Edge-side kernels of one DGT block (EquivariantMixBlock.forward, models/mol_gnn.py:270-322), directed form:
//   k_edge_scores GBF, edge_emb, LN1 + modulate, lin_edge0, q*k*tanh head scores   (:284-297, layers.py:165-174)
//   k_softmax     per (target node, head) max / 1/sum over its sources   (layers.py:178)
//   k_edge_msgs   lin_edge1, tanh, * v * alpha, summed over the sources of each target (layers.py:182-184)
//   k_edge_update gated residual + LN2 + FFN on edges, readout, MultiCondEquiUpdate (:313-320, :71-94, :568)
// (node-side kernels: dgt_kernels_node.h; symmetric pair variants: dgt_kernels_sym.h)
// Lanes hold the *group* index (attention target c for scores/msgs, row a for the update) and the
// wave iterates over the reduced index, so softmax statistics, message sums and coordinate sums
// never cross lanes.  Edge rows are indexed r = eoff + a*n + c (a = row = source, c = column = target)
// everywhere; the attention phases walk a column (stride n rows), the update phase a row, so no
// symmetry of the edge state is assumed (asymmetric caller inputs behave as in the reference).
#pragma once
#include "dgt_kernels_common.h"

namespace jd {

// ------------------------------------------------------------------------------------------------
// LDS-resident weights: the two edge-attention kernels use small projections (K = 64) whose whole
// weight set fits in LDS (scores: edge_emb 32 KiB + lin_edge0 64 KiB; msgs: lin_edge1 64 KiB).  A
// workgroup (8 waves for the directed scores kernel, 4 for the message and the pair scores kernels: a
// workgroup holds its CU slots until its slowest wave is done, so smaller is better once staging is
// amortised) loads it once; every iteration then reads its A operands with conflict-free ds_read_b128
// instead of streaming them from L2 (which was the L1-bandwidth limiter: 256 B of weights per MFMA at K = 64).
constexpr int WG_WAVES = 8;

template <int NQ, int NW = WG_WAVES>   // cooperative copy of NQ quads (1 KiB each) global -> LDS by NW waves
__device__ __forceinline__ void stage_weights(float4* __restrict__ dst, const float4* __restrict__ src) {
    for (int i = threadIdx.x; i < NQ * 64; i += NW * 64) dst[i] = src[i];
}

template <int KQ>
__device__ __forceinline__ f32x16 mfma_block_lds(const float4* wl, const float (&act)[KQ * 4], f32x16 acc) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const float4 a = wl[q * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, act[4 * q + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, act[4 * q + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, act[4 * q + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, act[4 * q + 3], acc, 0, 0, 0);
    }
    return acc;
}

// scores: S stored as [row][half*8 + b] = head 2b+half  (head 0/1 = adjacency heads, 2.. learned)
__global__ __launch_bounds__(WG_WAVES * 64, 2) void k_edge_scores(KArgs A) {
    if (!A.flags[FLAG_ASYM]) return;                            // symmetric inputs: k_edge_scores_sym runs instead
    __shared__ float4 wl[(32 + 64) * 64];                       // edge_emb (2 x 16 quads) | lin_edge0 (8 x 8 quads)
    stage_weights<32>(wl, reinterpret_cast<const float4*>(A.W + A.wb[JB_EE_W]));
    stage_weights<64>(wl + 32 * 64, reinterpret_cast<const float4*>(A.W + A.wb[JB_LE0_W]));
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int it = blockIdx.x * WG_WAVES + (threadIdx.x >> 6);
    if (it >= A.pd.n_items) return;
    const int strip = A.pd.item_strip[it], t0 = A.pd.item_t0[it], t1 = A.pd.item_t1[it];
    const LaneNode L = lane_node(A, strip, j);
    const float* mrow = mod_row(A, L.b) + A.mod_base;
    const float gscale = mrow[6 * 256 + 6 * 64 + 2 * 256 + 0], gshift = mrow[6 * 256 + 6 * 64 + 2 * 256 + 1];
    const float4 pv = reinterpret_cast<const float4*>(A.pos_out)[L.v];
    const float4* wEE = wl + lane;
    const float4* wL0 = wl + 32 * 64 + lane;
    for (int t = t0; t < t1; ++t) {
        const bool ok = L.valid && t < L.n;
        const int tc = ok ? t : 0;
        const int u = L.noff + tc;
        const size_t r = (size_t)L.eoff + (size_t)tc * L.n + L.i;      // edge (source a = t) -> (target c = i)
        const float* es1 = launder(mrow + 6 * 256);                 // edge chunks: es1, ec1, ...
        const float* ec1 = es1 + 64;
        const float* cst = launder(A.W);
        const float* tab = cst + A.wb[JB_GBF];
        const float* bEE = cst + A.wb[JB_EE_B];
        TRow qrow = trow(A.q, 8, L.v, half), krow = trow(A.k, 8, u, half);
        qrow.p = launder(qrow.p);
        const float4 pu = reinterpret_cast<const float4*>(A.pos_out)[u];
        const float dx = pv.x - pu.x, dy = pv.y - pu.y, dz = pv.z - pu.z;
        float x[32];
        {
            float G[32], e[32];
            gbf64(dx * dx + dy * dy + dz * dz, gscale, gshift, tab, half, G);
            load_nat<2>(A.e + r * 64, half, e);
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
        if (ok) store_nat<2>(A.et + r * 64, half, x);
        // lin_edge0 -> tanh -> * q_target * k_source, reduced per head
        float mainsum[7];
        float qn[16], kn[16];
        load16T(qrow, 0, qn);
        load16T(krow, 0, kn);
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            float qq[16], kk[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) { qq[s] = qn[s]; kk[s] = kn[s]; }
            load16T(qrow, b + 1, qn);               // one block ahead
            load16T(krow, b + 1, kn);
            f32x16 acc = mfma_block_lds<8>(wL0 + (b * 8) * 64, x, zero16());
            float s_ = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) s_ = fmaf(tanh_f(acc[s]) * qq[s], kk[s], s_);
            mainsum[b] = s_;                              // head 2b+half, channels 0..15
            pipeline_fence();
        }
        float tail[14];
        {
            f32x16 acc = mfma_block_lds<8>(wL0 + (7 * 8) * 64, x, zero16());
#pragma unroll
            for (int g = 0; g < 14; ++g) tail[g] = tanh_f(acc[g]) * qn[g] * kn[g];   // head g, channel 16+half
        }
        const int fl = A.eflag[r];
        // S_g = main_g (half g&1, block g>>1) + tail_g(half 0) + tail_g(half 1), scaled by 1/sqrt(C)
        float Sg[14];
#pragma unroll
        for (int g = 0; g < 14; ++g) {
            const float own = ((g & 1) == half) ? mainsum[g >> 1] : 0.f;
            Sg[g] = pair_sum(own + tail[g]) * 0.25f;
        }
        float Sout[8];                                  // slot b of this half = head 2b + half
        Sout[0] = half == 0 ? ((fl & 1) ? 1.f : -1e10f) : ((fl & 2) ? 1.f : -1e10f);
#pragma unroll
        for (int b = 1; b < 8; ++b) Sout[b] = half == 0 ? Sg[2 * (b - 1)] : Sg[2 * (b - 1) + 1];
        if (ok) {
            float4* sp = reinterpret_cast<float4*>(A.S + r * 16 + half * 8);
            sp[0] = make_float4(Sout[0], Sout[1], Sout[2], Sout[3]);
            sp[1] = make_float4(Sout[4], Sout[5], Sout[6], Sout[7]);
        }
    }
}

// per (target node, stored head slot): max and 1/(sum exp + 1e-16) over its sources t != i
__global__ void k_softmax(KArgs A) {
    const int v = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int slot = threadIdx.x & 15;
    if (v >= A.pd.Nn) return;
    const int n = A.pd.node_n[v], i = A.pd.node_i[v];
    const float* Sr = A.S + ((size_t)A.pd.node_eoff[v] + i) * 16 + slot;   // rows (t, i), stride n
    const size_t st = (size_t)n * 16;
    // one pass (running maximum with rescaled sum): S is read once instead of twice
    float m = -INFINITY, sum = 0.f;
    for (int t = 0; t < n; ++t) {
        if (t == i) continue;
        const float v = Sr[t * st];
        if (v > m) { sum = sum * __expf(m - v) + 1.f; m = v; }
        else sum += __expf(v - m);
    }
    if (n <= 1) { m = 0.f; sum = 0.f; }
    A.stats[(size_t)v * 32 + slot] = m;
    A.stats[(size_t)v * 32 + 16 + slot] = n > 1 ? 1.f / (sum + 1e-16f) : 0.f;
}

// ------------------------------------------------------------------------------------------------
constexpr int MSG_WAVES = 4;      // two 4-wave workgroups per CU (2 x 64 KiB LDS): finer scheduling grain than one of 8

__global__ __launch_bounds__(MSG_WAVES * 64, 2) void k_edge_msgs(KArgs A) {
    __shared__ float4 wl[64 * 64];                              // lin_edge1 (8 x 8 quads)
    stage_weights<64, MSG_WAVES>(wl, reinterpret_cast<const float4*>(A.W + A.wb[JB_LE1_W]));
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int it = blockIdx.x * MSG_WAVES + (threadIdx.x >> 6);
    if (it >= A.pd.n_items) return;
    const int strip = A.pd.item_strip[it], t0 = A.pd.item_t0[it], t1 = A.pd.item_t1[it], part = A.pd.item_part[it];
    const LaneNode L = lane_node(A, strip, j);
    // softmax statistics of this lane's target are re-read (L1 hits) every iteration instead of living in 16
    // registers: keeps the kernel free of scratch spills at 2 waves per SIMD
    const float4* stp = launder(reinterpret_cast<const float4*>(A.stats + (size_t)L.v * 32 + half * 8));
    const float4* wL1 = wl + lane;
    float macc[128];
#pragma unroll
    for (int s = 0; s < 128; ++s) macc[s] = 0.f;
    for (int t = t0; t < t1; ++t) {
        const bool ok = L.valid && t < L.n && t != L.i;
        const int tc = (L.valid && t < L.n) ? t : 0;
        const int u = L.noff + tc;
        const size_t r = (size_t)L.eoff + (size_t)tc * L.n + L.i;      // edge (source a = t) -> (target c = i)
        float x[32];
        load_nat<2>(A.et + r * 64, half, x);
        float al[8];
        {
            const float4* sp = reinterpret_cast<const float4*>(A.S + r * 16 + half * 8);
            const float4 a = sp[0], b = sp[1];
            const float sv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            const float4 m0 = stp[0], m1 = stp[1], i0 = stp[4], i1 = stp[5];
            const float mx[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
            const float inv[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
            for (int b2 = 0; b2 < 8; ++b2) al[b2] = ok ? fast_exp(sv[b2] - mx[b2]) * inv[b2] : 0.f;
        }
        const BRow vrow = brow(A.v, 8, u, half);     // buffer loads: the fence keeps them ahead of the MFMAs
        float vnext[16];
        bload16(vrow, 0, vnext);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float vv[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) vv[s] = vnext[s];
            if (b < 7) bload16(vrow, b + 1, vnext);   // one block ahead
            pipeline_fence();
            f32x16 acc = mfma_block_lds<8>(wL1 + (b * 8) * 64, x, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) macc[b * 16 + s] = fmaf(tanh_f(acc[s]) * vv[s], al[b], macc[b * 16 + s]);
            pipeline_fence();                                          // keep hoisting (and registers) bounded
        }
    }
    store_nat<8>(A.hhat + ((size_t)L.v * A.pd.max_parts + part) * 256, half, macc);
}

// ------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(64, 1) void k_edge_update(KArgs A) {
    if (!A.flags[FLAG_ASYM]) return;                            // symmetric inputs: k_edge_update_sym runs instead
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int it = blockIdx.x;
    const int strip = A.pd.item_strip[it], t0 = A.pd.item_t0[it], t1 = A.pd.item_t1[it], part = A.pd.item_part[it];
    const LaneNode L = lane_node(A, strip, j);
    const float* mrow = mod_row(A, L.b) + A.mod_base;
    const float* eg1 = mrow + 6 * 256 + 2 * 64;               // edge chunks: .., eg1, es2, ec2, eg2
    const float* qsh = mrow + 6 * 256 + 6 * 64;              // equi_update.time_mlp: (shift, scale)
    const float gscale = mrow[6 * 256 + 6 * 64 + 2 * 256 + 0], gshift = mrow[6 * 256 + 6 * 64 + 2 * 256 + 1];
    const float4 pv = reinterpret_cast<const float4*>(A.pos_out)[L.v];
    const float cscale = A.W[A.wb[JB_CSCALE]];
    const WSrc ws = make_wsrc(A.W, lane);
    // byte offsets of the weight blocks inside the blob (wave-uniform)
    const unsigned o3 = (unsigned)(A.wb[JB_FF3_W] * 4), o4 = (unsigned)(A.wb[JB_FF4_W] * 4);
    const unsigned oro = (unsigned)(A.wb[JB_ERO_W] * 4), oi = (unsigned)(A.wb[JB_INE_W] * 4), o0 = (unsigned)(A.wb[JB_C0_W] * 4);
    constexpr int KQ4 = R * 64 / 8;
    WPipe<8> wp;
    wpipe_prime(wp, ws, o3);
    float dax = 0.f, day = 0.f, daz = 0.f;
    // Operand loads of every VALU epilogue are issued BEFORE the MFMA block they follow (the blocks
    // contain scheduling barriers), so they land while the matrix pipe is busy.
    for (int t = t0; t < t1; ++t) {
        const bool inr = L.valid && t < L.n;
        const bool ok = inr && t != L.i;
        const int tc = inr ? t : 0;
        const int u = L.noff + tc;
        const size_t r = (size_t)L.eoff + (size_t)L.i * L.n + tc;
        // keep loop-invariant vectors out of registers (see launder())
        const float* eg1_ = launder(eg1);
        const float* es2_ = eg1_ + 64, *ec2_ = es2_ + 64, *eg2_ = ec2_ + 64;
        const float* qsh_ = launder(qsh);
        const float* qsc_ = qsh_ + 256;
        const float* cst = launder(A.W);                       // biases / tables / raw coord_mlp.2
        const float* n2bias_ = cst + A.wb[JB_N2E_B], *b3_ = cst + A.wb[JB_FF3_B], *b4_ = cst + A.wb[JB_FF4_B];
        const float* b0_ = cst + A.wb[JB_C0_B], *w2_ = cst + A.wb[JB_C2_W], *tab_ = cst + A.wb[JB_GBF];
        const float* bro_ = cst + A.wb[JB_ERO_B];
        TRow wrow = trow(A.wrow, 8, L.v, half);
        wrow.p = launder(wrow.p);
        const TRow wcol = trow(A.wcol, 8, u, half);
        // ---- geometry + Gaussian basis (needed by the equivariant update; computed up front) ----
        const float4 pu = reinterpret_cast<const float4*>(A.pos_out)[u];
        const float dx = pv.x - pu.x, dy = pv.y - pu.y, dz = pv.z - pu.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        float G[32];
        gbf64(d2, gscale, gshift, tab_, half, G);
        // ---- edge residual + LN2 + modulate ----
        float en[32];
        {
            float e[32], n2a[32], n2c[32];
            load_nat<2>(A.e + r * 64, half, e);
            {
                const TRow ra = trow(A.n2e, 2, L.v, half), rc = trow(A.n2e, 2, u, half);
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
        // ---- edge FFN (hidden R*64, chunks of 64) ----
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
                if (c == R - 1) {                               // operands of the FFN epilogue
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
        if (inr) store_nat<2>(A.e_out + r * 64, half, en);
        // ---- readout edge_l(e) -> edge_hids[:, De + l*16 ...] (valid outputs live in half 0) ----
        {
            float bb[16];
            load16(bro_ + half * 16, bb);
            f32x16 acc = mfma_block_p<8>(wp, ws, oro, oi, en, zero16());
            float rr[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) rr[s] = acc[s] + bb[s];
            if (inr && half == 0) store16(A.ehid + r * A.d.KEH + 64 + A.layer * 16, rr);
        }
        // ---- equivariant update: u = W_e e + W_d G + (W_row h_a + b) + W_col h_c, 8 accumulators that
        //      start from the per-node terms ----
        f32x16 U[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned wcur = oi + (unsigned)(b * 16) * 1024;
            const unsigned wnx = b < 7 ? oi + (unsigned)((b + 1) * 16) * 1024 : oi + 8u * 1024;
            float a1[16], a2[16];
            load16T(wrow, b, a1);
            load16T(wcol, b, a2);
            U[b] = mfma_block_p<8>(wp, ws, wcur, wnx, en, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) U[b][s] += a1[s] + a2[s];
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned wcur = oi + (unsigned)(b * 16 + 8) * 1024;
            const unsigned wnx = b < 7 ? oi + (unsigned)((b + 1) * 16 + 8) * 1024 : o0;
            U[b] = mfma_block_p<8>(wp, ws, wcur, wnx, G, U[b]);
        }
        float uu[128];
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int s = 0; s < 16; ++s) uu[b * 16 + s] = U[b][s];
        layer_norm<128>(uu);
        modulate<8>(uu, qsh_, qsc_, half);
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll 1
        for (int b = 0; b < 8; ++b) {
            const unsigned wcur = o0 + (unsigned)b * 32 * 1024;
            const unsigned wnx = b < 7 ? wcur + 32 * 1024 : o3;
            float bb[16], k0[16], k1[16], k2[16];
            load16(b0_ + b * 32 + half * 16, bb);
            load16(w2_ + b * 32 + half * 16, k0);
            load16(w2_ + 256 + b * 32 + half * 16, k1);
            load16(w2_ + 512 + b * 32 + half * 16, k2);
            f32x16 acc = mfma_block_p<32>(wp, ws, wcur, wnx, uu, zero16());
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

}  // namespace jd
