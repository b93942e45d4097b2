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
// flash-style: every lane owns one target atom and keeps a running (max, sum) per head and D/2 message accumulators;
// k_node_post merges the partials of the items that shared a target (attn_merge below).
//
// Pair mode (symmetric edge inputs, i.e. sampling).  The edge state is exactly symmetric, so lane i evaluates the pair
// {i, j = (i + d) mod n} once (circulant walk d = 1 .. n/2, pair_of() in dgt_kernels_sym.h).  Target i's new source is
// j, computed in place.  Target j's new source is i: its scores and its unweighted message v_i * T1 are handed to j's
// lane through LDS — the sender writes its own slot, the receiver reads the slot of lane (i - d) mod n.  That needs both
// atoms of a pair in one workgroup, hence the plan's groups: whole molecules, up to 128 lanes, one 4-wave workgroup
// (dgt_plan.cpp).  One workgroup barrier for the scores and one per 32-feature message block (double buffered); all
// waves of an item run the same offsets.
//
// Directed mode (asymmetric caller inputs; molecules larger than a group): lane = target, the wave visits every
// source itself — no hand-over, no barriers, twice the matrix work per edge.
//
// template <D, WQK>: node width and q / k arrangement.  WQK = false (nf = 256 only): the tuned 8-block arrangement
// (csrc/dgt_pack.cpp qk_out_map), edge_emb + lin_edge0 (96 KiB) resident in LDS, lin_edge1 streamed from L2 through
// the software-pipelined ring; WQK = true: one 32-row block per head (qk_out_map_wide; the only option at nf = 384,
// SC = 27), all weights streamed.  40 KiB of LDS carry the hand-over buffers.
#pragma once
#include "dgt_kernels_sym.h"
#include "dgt_split.h"

// per-phase cycle sums of the attention kernel (debug builds with -DJODO_PHASE_TIMING_ATTN; tools/phase_timing.py attn)
#ifdef JODO_PHASE_TIMING_ATTN
#define APT_INIT unsigned long long pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_last = __builtin_readcyclecounter();
#define APT(ph) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_readcyclecounter(); pt_acc[ph] += n_ - pt_last; pt_last = n_; } while (0)
#define APT_FLUSH do { if (A.dbgt && (threadIdx.x & 63) == 0 && (blockIdx.x % 61) == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&A.dbgt[i_], pt_acc[i_]); atomicAdd(&A.dbgt[15], (unsigned long long)(t1 - t0)); } } while (0)
#else
#define APT_INIT
#define APT(ph)
#define APT_FLUSH
#endif

namespace jd {

constexpr int ATT_WAVES = 4;
constexpr int ATT_LANES = 128;
static_assert(ATT_LANES == PAIR_GROUP_LANES, "group size");
constexpr float ATT_NEG = -3.0e38f;                    // "no source yet": exp(ATT_NEG - m) == 0 for every real m

template <int D_, bool WQK_, int VAR_ = 0>
struct AttnT {
    static constexpr int D = D_, De = D_ / 4, NE = D_ / 128, HE = D_ / 8, ND = D_ / 32, HD = D_ / 2, C = D_ / 16;
    static constexpr bool WQK = WQK_;
    // weight residency / hand-over granularity (nf = 256, tuned arrangement only; the others stream everything):
    //   VAR 0: edge_emb + lin_edge0 in LDS (96 KiB), messages handed over block by block (double buffered, a barrier each)
    //   VAR 1: lin_edge0 in LDS (64 KiB), edge_emb streamed, messages handed over four blocks at a time (64 KiB)
    //   VAR 2: everything streamed, all message blocks handed over at once (128 KiB)
    //   VAR 3: edge_emb + the seven main blocks of lin_edge0 in LDS (88 KiB), its tail block streamed, four blocks per
    //          hand-over (64 KiB): exactly the 160 KiB of a CU
    //   VAR 4 / 5 (OPT-IN, JODO_OPT_SPLIT_BF16 = 2): the split-bf16 form — every projection operand as three bf16 terms (dgt_split.h), all
    //          three weight sets streamed through ONE LDS ring shared by the four waves (the cyclic tapes of dgt_pack.cpp).  The work of an
    //          item is dealt to TWO launches by heads: VAR 4 owns message blocks 0 .. 3 (head slots 0 .. 3: the adjacency heads and learned
    //          heads 0 .. 5, lin_edge0 blocks 0 .. 2 + its tail block), VAR 5 message blocks 4 .. 7 (slots 4 .. 7: learned heads 6 .. 13,
    //          lin_edge0 blocks 3 .. 6 + tail).  Each carries 64 message accumulators instead of 128 — the one-launch form (round 6, first
    //          attempt) needed 128 accumulators + the split operands, spilled 1.6 KB per lane and ran 3 x slower than the fp32 kernel.  The
    //          price: the edge input (4 of 20 chunk blocks) is evaluated by both.  Messages handed over block by block as in VAR 0.
    static constexpr bool SPLIT = !WQK_ && (VAR_ == 4 || VAR_ == 5);
    static constexpr bool LDS_EE = !WQK_ && (VAR_ == 0 || VAR_ == 3);
    static constexpr bool LDS_L0 = !WQK_ && VAR_ != 2 && !SPLIT;
    static constexpr bool LDS_TAIL = LDS_L0 && VAR_ != 3;          // tail block of lin_edge0 (tuned arrangement) resident too
    static constexpr int PHB = WQK_ ? 1 : ((VAR_ == 0 || SPLIT) ? 1 : (VAR_ == 2 ? D_ / 32 : 4));   // message blocks per hand-over phase
    static constexpr int NQB = WQK_ ? 14 : 8;                      // 32-row blocks of q / k / lin_edge0
    static constexpr int KQE = D_ / 32;                            // weight quads per output block for K = De
#ifndef JODO_X_ATT_PG384                                           // experiment builds (tools/gpu_attn384_ab.sh): -DJODO_X_ATT_...
#define JODO_X_ATT_PG384 4
#endif
#ifndef JODO_X_ATT_PREF384
#define JODO_X_ATT_PREF384 0
#endif
#ifndef JODO_X_ATT_LDSS384
#define JODO_X_ATT_LDSS384 1
#endif
    static constexpr int PG = (D_ % 256 == 0) ? 8 : (D_ == 384 ? JODO_X_ATT_PG384 : 4);   // quads in flight (must divide KQE)
    static constexpr bool PH = (D_ / 16 == 16) && !(D_ > 256);     // C = 16: a half-lane's registers belong to heads 2b + half only, so it
    static constexpr int NS = PH ? 8 : 16;                         // tracks 8 heads (slot k = head 2k + half) instead of all 16
    static constexpr bool PREF = (!(D_ > 256) || JODO_X_ATT_PREF384 != 0) && !SPLIT;   // request the next source's edge row one iteration ahead (D/8 registers; not in the split form: its operands' split images need them)
    static constexpr bool QK2 = false;   // q / k rows two blocks ahead in two register sets: measured SLOWER on MI355X (QM9 B = 2500: attention
                                         // 3.91 -> 4.00 ms/step, 455 -> 490 registers) — the block's wait is issue, not row latency; kept as a switch
    static constexpr bool LDSS = D_ > 256 && JODO_X_ATT_LDSS384 != 0;   // running softmax state in LDS (registers are short at nf = 384:
                                                                   // D/2 accumulators + D/8 inputs per lane; LDS is free, no resident weights)
    // the share of one launch: head slots [SL0, SL0 + NSL) (PH: slot k = head 2k + half = message block k), score blocks [SB0, SB1) of
    // lin_edge0 besides its tail block, learned heads [G0, G1) in the tail block; everything unless SPLIT
    static constexpr int SL0 = SPLIT && VAR_ == 5 ? 4 : 0, NSL = SPLIT ? 4 : NS;
    static constexpr int MB0 = SPLIT ? SL0 : 0, MB1 = SPLIT ? SL0 + 4 : D_ / 32;
    static constexpr int SB0 = SPLIT && VAR_ == 5 ? 3 : 0, SB1 = SPLIT ? (VAR_ == 5 ? 7 : 3) : (WQK_ ? 14 : 7);
    static constexpr int G0 = SPLIT && VAR_ == 5 ? 6 : 0, G1 = SPLIT && VAR_ == 4 ? 6 : 14;
    static constexpr int TAPE_CHUNKS = 4 + (SB1 - SB0) + 1 + 4;    // SPLIT: chunks of four K16 steps per pair offset (12 / 13)
    static constexpr int TAPE_FIRST = VAR_ == 5 ? 12 : 0;          // the second launch's tape follows the first's
    static_assert(!SPLIT || (PH && D_ == 256), "split form: nf 256, one head slot per message block");
    static constexpr float INV_SQRT_C = D_ == 256 ? 0.25f : (D_ == 384 ? 0.20412414523193150f : (D_ == 128 ? 0.35355339059327379f : 0.f));
    static constexpr int M_EDGE = 6 * D_, M_GBF = 6 * D_ + 6 * (D_ / 4) + 2 * D_;
    static_assert(!(LDS_EE || LDS_L0) || D_ == 256, "LDS-resident weights are sized for nf = 256");
    static_assert(C % 8 == 0, "message blocks are split at register 8 between two heads");
};

// Gaussian basis (CondGaussianLayer, layers.py:291-295, :328-334) for De = 32 * NB features
template <int NB>
__device__ __forceinline__ void gbf_n(float d2, float scale, float shift, const float* __restrict__ tab, int half, float (&g)[NB * 16]) {
    constexpr int De = NB * 32;
    const float x = fmaf(d2, scale + 1.f, shift);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float mu[16], is[16], cf[16];
        load16(tab + b * 32 + half * 16, mu);
        load16(tab + De + b * 32 + half * 16, is);
        load16(tab + 2 * De + b * 32 + half * 16, cf);
#pragma unroll
        for (int s = 0; s < 16; s += 2) {                 // pairs on the packed pipe; e^{-z^2/2} = 2^{-(log2 e / 2) z^2}
            f32x2 m, i2, c, z;
            m.x = mu[s]; m.y = mu[s + 1]; i2.x = is[s]; i2.y = is[s + 1]; c.x = cf[s]; c.y = cf[s + 1];
            z = (x - m) * i2;
            z = (z * -0.72134752044448170f) * z;
            f32x2 e;
            e.x = __builtin_amdgcn_exp2f(z.x); e.y = __builtin_amdgcn_exp2f(z.y);
            e = e * c;
            g[b * 16 + s] = e.x; g[b * 16 + s + 1] = e.y;
        }
    }
    if (half == 0) g[0] = x;
}

// weight source of one kernel instance: LDS image (tuned) or the streamed ring (wide)
template <typename X>
struct AttnW {
    const float4* wEE;      // LDS: edge_emb, lin_edge0 (this lane's float4 of quad 0)
    const float4* wL0;
    WSrc ws;
    unsigned oEE, oL0, oL1;
    WPipe<X::PG> wp;
    splitc::TapeC T;        // SPLIT: the shared weight tape and the chunk this wave consumes next
    int g;
};

// one K = De output block; `cur` / `nxt` are byte offsets for the streamed case, `wl` the LDS block for the resident case
// `after`: row gathers for the next block — behind the weight prefetch in the streamed case (mfma_block_p2 in dgt_device.h)
template <typename X, bool LDS, typename After = NoHook>
__device__ __forceinline__ f32x16 attn_block(AttnW<X>& w, const float4* wl, unsigned cur, unsigned nxt, const float (&act)[X::HE], f32x16 acc,
                                             After&& after = NoHook()) {
    if constexpr (LDS) { after(); return mfma_block_lds_p<X::KQE>(wl, act, acc); }
    else return mfma_block_p2<X::KQE>(w.wp, w.ws, cur, w.ws, nxt, act, acc, after);
}

// et of an edge row from its state e and squared length d2 (GBF -> edge_emb -> LN1 -> modulate)
template <typename X>
__device__ __forceinline__ void attn_edge_input(const KArgs& A, AttnW<X>& w, const float (&e)[X::HE], float d2, float gscale, float gshift,
                                                const float* mrow, int half, float (&x)[X::HE]) {
    const float* es1 = launder(mrow + X::M_EDGE);
    const float* ec1 = es1 + X::De;
    const float* cst = launder(A.W);
    const float* tab = cst + A.wb[JB_GBF];
    const float* bEE = cst + A.wb[JB_EE_B];
    float G[X::HE];
    gbf_n<X::NE>(d2, gscale, gshift, tab, half, G);
    if constexpr (X::SPLIT) {
        static_assert(X::HE == 32, "split form: nf 256");
        static_assert(X::NE == 2, "split form: nf 256");
        // one operand's split image at a time (tape order: the G halves of both output blocks, then the e halves): 128-bit register tuples
        // are what this kernel runs out of
        f32x16 acc0, acc1;
        {
            Split8 os[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) os[q] = split8(&G[8 * q]);
            acc0 = splitc::block4(w.T, w.g, os, zero16());
            acc1 = splitc::block4(w.T, w.g, os, zero16());
        }
        {
            Split8 os[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) os[q] = split8(&e[8 * q]);
            acc0 = splitc::block4(w.T, w.g, os, acc0);
            acc1 = splitc::block4(w.T, w.g, os, acc1);
        }
        {
            float bb[16];
            load16(bEE + half * 16, bb);
#pragma unroll
            for (int s = 0; s < 16; ++s) x[s] = acc0[s] + bb[s];
            load16(bEE + 32 + half * 16, bb);
#pragma unroll
            for (int s = 0; s < 16; ++s) x[16 + s] = acc1[s] + bb[s];
        }
        layer_norm<X::HE>(x);
        modulate<X::NE>(x, es1, ec1, half);
        return;
    }
#pragma unroll
    for (int b = 0; b < X::NE; ++b) {
        const unsigned cg = w.oEE + (unsigned)(b * 2 * X::KQE) * 1024, ce = cg + X::KQE * 1024;
        float bb[16];
        load16(bEE + b * 32 + half * 16, bb);
        f32x16 acc = attn_block<X, X::LDS_EE>(w, w.wEE + (b * 2 * X::KQE) * 64, cg, ce, G, zero16());
        acc = attn_block<X, X::LDS_EE>(w, w.wEE + ((b * 2 + 1) * X::KQE) * 64, ce, b + 1 < X::NE ? ce + X::KQE * 1024 : (X::LDS_L0 ? w.oL1 : w.oL0), e, acc);
#pragma unroll
        for (int s = 0; s < 16; ++s) x[b * 16 + s] = acc[s] + bb[s];
    }
    layer_norm<X::HE>(x);
    modulate<X::NE>(x, es1, ec1, half);
}

// scores of one pair for both directions from T0 = tanh(lin_edge0 x): S1 = edge (j -> i), S2 = edge (i -> j); all 16
// heads in every lane (heads 0, 1 = adjacency heads from the edge flags f1 / f2, 2.. = learned)
// first q / k rows of a pair (block 0): requested by the caller ahead of the edge-input projections where registers allow
// q / k rows of a pair, one register set per block in flight (X::QK2: two alternating sets, [0] even blocks, [1] odd)
struct QKRows { float qi[16], ki[16], qj[16], kj[16]; };
template <bool BOTH>
__device__ __forceinline__ void attn_rows(const BRow& qi, const BRow& ki, const BRow& qj, const BRow& kj, int b, QKRows& r) {
    bload16(qi, b, r.qi); bload16(kj, b, r.kj);
    if (BOTH) { bload16(qj, b, r.qj); bload16(ki, b, r.ki); }
}

template <typename X, bool BOTH>
__device__ __forceinline__ void attn_scores(AttnW<X>& w, const float (&x)[X::HE], const BRow& qi, const BRow& ki, const BRow& qj,
                                            const BRow& kj, int half, int f1, int f2, float (&S1)[16], float (&S2)[16],
                                            QKRows (&rows)[X::QK2 ? 2 : 1], const Split8* xs = nullptr) {
    // rows[b & 1] (QK2) / rows[0] holds block b on entry to iteration b: blocks 0 (and 1) were requested by the caller
    float m1[X::WQK ? 1 : 7], m2[X::WQK ? 1 : 7];       // blocks reduced per head / per head pair
    S1[0] = (f1 & 1) ? 1.f : -1e10f; S1[1] = (f1 & 2) ? 1.f : -1e10f;           // extra heads, 0 -> -1e10 (layers.py:170-174)
    S2[0] = (f2 & 1) ? 1.f : -1e10f; S2[1] = (f2 & 2) ? 1.f : -1e10f;
#pragma unroll
    for (int b = X::SB0; b < X::SB1; ++b) {
        float a1[16], a2[16];
        QKRows& R = rows[X::QK2 ? (b & 1) : 0];
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
            const f32x2 p1 = pk2(R.qi[s], R.qi[s + 1]) * pk2(R.kj[s], R.kj[s + 1]);
            a1[s] = p1.x; a1[s + 1] = p1.y;
            if (BOTH) { const f32x2 p2 = pk2(R.qj[s], R.qj[s + 1]) * pk2(R.ki[s], R.ki[s + 1]); a2[s] = p2.x; a2[s + 1] = p2.y; }
            else { a2[s] = 0.f; a2[s + 1] = 0.f; }
        }
        auto next_rows = [&]() {                       // the set just consumed takes the block two (one) ahead
            constexpr int AHEAD = X::QK2 ? 2 : 1;
            if constexpr (X::SPLIT) attn_rows<BOTH>(qi, ki, qj, kj, b + 1 < X::SB1 ? b + 1 : X::NQB - 1, R);   // after its share: the tail block
            else if (b + AHEAD < X::NQB) attn_rows<BOTH>(qi, ki, qj, kj, b + AHEAD, R);
        };
        if constexpr (X::LDS_L0) pipeline_fence();
        const unsigned cur = w.oL0 + (unsigned)(b * X::KQE) * 1024;
        f32x16 acc;
        if constexpr (X::SPLIT) { next_rows(); pipeline_fence(); acc = splitc::block4(w.T, w.g, xs, zero16()); }
        else acc = attn_block<X, X::LDS_L0>(w, w.wL0 + (b * X::KQE) * 64, cur, b + 1 < X::NQB ? cur + X::KQE * 1024 : w.oL1, x, zero16(), next_rows);
        float tt[16];
        tanh16(acc, tt);
        f32x2 s1a = {0.f, 0.f}, s1b = s1a, s2a = s1a, s2b = s1a;   // partial sums on the packed pipe, two chains per direction
#pragma unroll                                              // (dependent packed ops back to back pay a wait state)
        for (int s = 0; s < 16; s += 4) {
            const f32x2 t2 = pk2(tt[s], tt[s + 1]), t3 = pk2(tt[s + 2], tt[s + 3]);
            s1a = __builtin_elementwise_fma(t2, pk2(a1[s], a1[s + 1]), s1a);
            s1b = __builtin_elementwise_fma(t3, pk2(a1[s + 2], a1[s + 3]), s1b);
            if (BOTH) {
                s2a = __builtin_elementwise_fma(t2, pk2(a2[s], a2[s + 1]), s2a);
                s2b = __builtin_elementwise_fma(t3, pk2(a2[s + 2], a2[s + 3]), s2b);
            }
        }
        const f32x2 s1v = s1a + s1b, s2v = s2a + s2b;
        const float s1 = s1v.x + s1v.y, s2 = s2v.x + s2v.y;
        if constexpr (X::WQK) {                        // head g = block g (padded rows are zero in q and k)
            S1[2 + b] = pair_sum(s1) * X::INV_SQRT_C;                          // / sqrt(out_channels = D / H), layers.py:167
            S2[2 + b] = BOTH ? pair_sum(s2) * X::INV_SQRT_C : 0.f;
        } else {
            m1[b] = s1; m2[b] = s2;
        }
        pipeline_fence();
    }
    if constexpr (!X::WQK) {                           // tuned: half h of block b = head 2b + h (channels 0..15); tail block:
        float tl1[14], tl2[14];                        // register g of half h = head g, channel 16 + h
        f32x16 acc;
        if constexpr (X::SPLIT) acc = splitc::block4(w.T, w.g, xs, zero16());
        else acc = attn_block<X, X::LDS_TAIL>(w, w.wL0 + (7 * X::KQE) * 64, w.oL0 + (unsigned)(7 * X::KQE) * 1024, w.oL1, x, zero16());
#pragma unroll
        for (int g = X::G0; g < X::G1; ++g) {
            const float tt = tanh_f(acc[g]);
            const QKRows& R = rows[X::QK2 ? ((X::NQB - 1) & 1) : 0];
            tl1[g] = tt * R.qi[g] * R.kj[g];
            tl2[g] = BOTH ? tt * R.qj[g] * R.ki[g] : 0.f;
        }
#pragma unroll
        for (int g = X::G0; g < X::G1; ++g) {
            const float o1 = ((g & 1) == half) ? m1[g >> 1] : 0.f;
            const float o2 = ((g & 1) == half) ? m2[g >> 1] : 0.f;
            S1[2 + g] = pair_sum(o1 + tl1[g]) * X::INV_SQRT_C;
            S2[2 + g] = BOTH ? pair_sum(o2 + tl2[g]) * X::INV_SQRT_C : 0.f;
        }
    }
}

// the two heads a 16-register group of message block b belongs to: registers 0-7 -> lo, 8-15 -> hi (C is a multiple of 8)
template <typename X>
__device__ __forceinline__ void attn_pick(const float (&v)[16], int b, int half, float& lo, float& hi) {
    lo = half ? v[(b * 32 + 16) / X::C] : v[(b * 32) / X::C];
    hi = half ? v[(b * 32 + 24) / X::C] : v[(b * 32 + 8) / X::C];
}

// the same from a per-thread LDS column (element h at col[h * 256])
template <typename X>
__device__ __forceinline__ void attn_pick_lds(const float* col, int b, int half, float& lo, float& hi) {
    lo = col[(half ? (b * 32 + 16) / X::C : (b * 32) / X::C) * 256];
    hi = col[(half ? (b * 32 + 24) / X::C : (b * 32 + 8) / X::C) * 256];
}

// One work item of the fused attention phase: group `grp` (128 lanes of whole molecules, ag_node), iterations [t0, t1) — pair
// offsets d = t + 1 (PAIR) or sources t (directed) —, partial index `part`.  wl: the workgroup's LDS-resident weights;
// sx / ux: the pair mode's hand-over buffers (untouched in directed mode); stt: the running softmax state where it lives in LDS.
template <typename X, bool PAIR>
__device__ __forceinline__ void attn_item(const KArgs& A, const float4* wl, float4* sx, float4* ux, float* stt, int grp, int t0, int t1, int part) {
    constexpr int D = X::D;
    constexpr bool PREF = X::PREF;
    const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
    const int ln = wave * 32 + (lane & 31);             // lane of the group
    float* const stc = stt + threadIdx.x;                // element (k, h) of this thread at stc[(k * 16 + h) * 256]
    const int vraw = A.pd.ag_node[grp * ATT_LANES + ln];
    LaneNode L;
    L.valid = vraw >= 0;
    L.v = L.valid ? vraw : 0;
    L.b = A.pd.node_b[L.v]; L.i = A.pd.node_i[L.v]; L.n = L.valid ? A.pd.node_n[L.v] : 0;
    L.noff = A.pd.node_noff[L.v]; L.eoff = A.pd.node_eoff[L.v];
    const int lbase = ln - L.i;                         // group lane of atom 0 of this lane's molecule (pair mode)
    const float* mrow = mod_row(A, L.b) + A.mod_base;
    const float gscale = mrow[X::M_GBF + 0], gshift = mrow[X::M_GBF + 1];
    const float4 pv = reinterpret_cast<const float4*>(A.pos_out)[L.v];
    AttnW<X> w;
    w.wEE = wl + lane;
    w.wL0 = wl + (X::LDS_EE ? 32 * 64 : 0) + lane;
    w.ws = make_wsrc(A.W, lane);
    w.oEE = (unsigned)(A.wb[JB_EE_W] * 4); w.oL0 = (unsigned)(A.wb[JB_LE0_W] * 4); w.oL1 = (unsigned)(A.wb[JB_LE1_W] * 4);
    // first block of the streamed ring of one iteration
    const unsigned ring0 = !X::LDS_EE ? w.oEE : ((X::LDS_L0 && !X::LDS_TAIL) ? w.oL0 + (unsigned)(7 * X::KQE) * 1024 : w.oL1);
    if constexpr (X::SPLIT) {
        // the cyclic tape of this launch's share (A.wsplit_attn: 12 chunks of four K16 steps for VAR 4, then 13 for VAR 5); `wl` is the ring
        w.T.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A.wsplit_attn) + (size_t)X::TAPE_FIRST * (splitc::CH_BYTES / 2), 0, 0x7fffffff, 0x00020000);
        w.T.period = X::TAPE_CHUNKS;
        w.T.ntot = X::TAPE_CHUNKS * (t1 - t0);
        w.T.ld_off = (unsigned)wave * 3072u + (unsigned)lane * 16u;
        w.T.rd_off = (unsigned)lane * 16u;
        w.T.ring = reinterpret_cast<char*>(const_cast<float4*>(wl));
        w.g = 0;
        if (t0 < t1) splitc::start(w.T);
    } else {
        wpipe_prime(w.wp, w.ws, ring0);
    }
    float sm[X::LDSS ? 1 : X::NS], sl[X::LDSS ? 1 : X::NS];   // running max / sum of this target (X::NS head slots)
    float macc[X::HD];
#pragma unroll
    for (int h = X::SL0; h < X::SL0 + X::NSL; ++h) {
        if constexpr (X::LDSS) { stc[h * 256] = ATT_NEG; stc[(16 + h) * 256] = 0.f; }
        else { sm[h] = ATT_NEG; sl[h] = 0.f; }
    }
#pragma unroll
    for (int s = X::MB0 * 16; s < X::MB1 * 16; ++s) macc[s] = 0.f;
    const int slot = half * ATT_LANES + ln;             // this lane's slot in the hand-over buffers
    // the source of iteration t (pair mode: partner (i + t + 1) mod n and the lane that hands its result over)
    struct Src { bool ok, rok; int u, rln; size_t r_in, r_out; };
    auto source = [&](int t) {
        Src S;
        S.rok = false; S.rln = ln; S.r_out = 0;
        if (PAIR) {
            const PairLane P = pair_of(L, t + 1);
            S.ok = P.ok; S.u = P.u; S.r_in = P.rji; S.r_out = P.rij;
            const int d = t + 1;
            int rr = L.i - d;
            if (rr < 0) rr += L.n;
            S.rok = L.valid && L.n > 1 && (2 * d < L.n || (2 * d == L.n && 2 * rr < L.n));   // did lane (i - d) evaluate {i - d, i}?
            S.rln = S.rok ? lbase + rr : ln;
        } else {
            const bool inr = L.valid && t < L.n;
            S.ok = inr && t != L.i;
            const int tc = inr ? t : 0;
            S.u = L.noff + tc;
            S.r_in = (size_t)L.eoff + (size_t)tc * L.n + L.i;          // edge (source a = t) -> (target c = i)
        }
        return S;
    };
    // Per-source inputs that come straight from HBM (the edge row is read once per block by exactly one lane) are
    // requested one iteration ahead: the row of iteration t + 1 is in flight while iteration t computes.
    // pair mode reads row (i, j) like the other pair kernels (the state is symmetric); directed mode the true row
    Src cur = source(t0 < t1 ? t0 : 0);
    float e[X::HE];
    float4 pu = make_float4(0.f, 0.f, 0.f, 0.f);
    int f1 = 0, f2 = 0;
    auto request = [&]() {
        load_nat<X::NE>(A.e + (PAIR ? cur.r_out : cur.r_in) * X::De, half, e);
        pu = reinterpret_cast<const float4*>(A.pos_out)[cur.u];
        f1 = A.eflag[cur.r_in];
        if (PAIR) f2 = A.eflag[cur.r_out];
    };
    if constexpr (PREF) request();
    APT_INIT
    APT(0);
    for (int t = t0; t < t1; ++t) {
        if constexpr (!PREF) {                         // registers are short: request at the point of use
            cur = source(t);
            request();
        }
        const bool ok = cur.ok, rok = cur.rok;
        const int u = cur.u, rln = cur.rln;
        const float dx = pv.x - pu.x, dy = pv.y - pu.y, dz = pv.z - pu.z;
        const int fl1 = f1, fl2 = f2;
        float x[X::HE];
        // the first q / k rows of the pair travel behind the edge-input projections (8k cycles) where registers allow
        const BRow qi = brow(A.q, X::NQB, L.v, half), ki = brow(A.k, X::NQB, L.v, half);
        const BRow qj = brow(A.q, X::NQB, u, half), kj = brow(A.k, X::NQB, u, half);
        QKRows rows[X::QK2 ? 2 : 1];
        if constexpr (PREF) attn_rows<PAIR>(qi, ki, qj, kj, X::SB0, rows[0]);
        attn_edge_input<X>(A, w, e, dx * dx + dy * dy + dz * dz, gscale, gshift, mrow, half, x);
        Split8 xs[X::SPLIT ? 4 : 1];                    // SPLIT: et as split operands, for the 16 projection blocks of scores and messages
        if constexpr (X::SPLIT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) xs[q] = split8(&x[8 * q]);
        }
        APT(1);
        if constexpr (PREF) {
            cur = source(t + 1 < t1 ? t + 1 : t);      // next source: its row, position and flags are requested now
            request();
        } else {
            attn_rows<PAIR>(qi, ki, qj, kj, X::SB0, rows[0]);
        }
        if constexpr (X::QK2) attn_rows<PAIR>(qi, ki, qj, kj, 1, rows[1]);   // second set: requested behind the edge-input projections
        // ---- scores ----
        float Sa[X::NS], R[X::LDSS ? 1 : X::NS];           // scores of the own source / of the handed-over source per head slot
        {
            float S1[16], S2[16];
            attn_scores<X, PAIR>(w, x, qi, ki, qj, kj, half, fl1, fl2, S1, S2, rows, xs);
            float Sb[X::NS];
#pragma unroll
            for (int k = X::SL0; k < X::SL0 + X::NSL; ++k) {
                Sa[k] = X::PH ? (half ? S1[2 * k + 1] : S1[2 * k]) : S1[k];
                Sb[k] = X::PH ? (half ? S2[2 * k + 1] : S2[2 * k]) : S2[k];
            }
            if (PAIR) {                                 // PH: a half hands its 8 slots to the same half of the partner; otherwise half h
                if constexpr (X::SPLIT) {               // the launch's four slots
                    sx[slot] = make_float4(Sb[X::SL0], Sb[X::SL0 + 1], Sb[X::SL0 + 2], Sb[X::SL0 + 3]);
                } else if constexpr (X::PH) {           // hands over heads 8h .. 8h + 7 and the receiver reads both halves
                    sx[slot] = make_float4(Sb[0], Sb[1], Sb[2], Sb[3]);
                    sx[256 + slot] = make_float4(Sb[4], Sb[5], Sb[6], Sb[7]);
                } else {
                    sx[slot] = half ? make_float4(Sb[8], Sb[9], Sb[10], Sb[11]) : make_float4(Sb[0], Sb[1], Sb[2], Sb[3]);
                    sx[256 + slot] = half ? make_float4(Sb[12], Sb[13], Sb[14], Sb[15]) : make_float4(Sb[4], Sb[5], Sb[6], Sb[7]);
                }
            }
        }
        APT(2);
        // first v rows: requested ahead of the hand-over barrier and the softmax update (where registers allow)
        const BRow vj = brow(A.v, X::ND, u, half), vi = brow(A.v, X::ND, L.v, half);
        float vjn[16], vin[16];
        if constexpr (PREF) {
            bload16(vj, X::MB0, vjn);
            if (PAIR) bload16(vi, X::MB0, vin);
        }
        if (PAIR) {
            __syncthreads();
            if constexpr (X::SPLIT) {
                const float4 a = sx[half * ATT_LANES + rln];
                R[X::SL0] = a.x; R[X::SL0 + 1] = a.y; R[X::SL0 + 2] = a.z; R[X::SL0 + 3] = a.w;
            } else if constexpr (X::PH) {
                const float4 a = sx[half * ATT_LANES + rln], b = sx[256 + half * ATT_LANES + rln];
                R[0] = a.x; R[1] = a.y; R[2] = a.z; R[3] = a.w; R[4] = b.x; R[5] = b.y; R[6] = b.z; R[7] = b.w;
            } else if constexpr (!X::LDSS) {
                const float4 a = sx[rln], b = sx[256 + rln], c = sx[ATT_LANES + rln], e = sx[256 + ATT_LANES + rln];
                R[0] = a.x; R[1] = a.y; R[2] = a.z; R[3] = a.w; R[4] = b.x; R[5] = b.y; R[6] = b.z; R[7] = b.w;
                R[8] = c.x; R[9] = c.y; R[10] = c.z; R[11] = c.w; R[12] = e.x; R[13] = e.y; R[14] = e.z; R[15] = e.w;
            }
        }
        APT(3);
        // ---- running softmax of this target: up to two new sources ----
        float sc[X::LDSS ? 1 : X::NS], p1[X::LDSS ? 1 : X::NS], p2[X::LDSS ? 1 : X::NS];
#pragma unroll
        for (int h = X::SL0; h < X::SL0 + X::NSL; ++h) {
            const float m0 = X::LDSS ? stc[h * 256] : sm[h];
            // head h of the partner's hand-over: quad (h & 7) / 4 of half-slot h / 8, component h & 3
            const float rh = !PAIR ? 0.f : (X::LDSS ? reinterpret_cast<const float*>(sx + ((h & 7) / 4) * 256 + (h / 8) * ATT_LANES + rln)[h & 3] : R[h]);
            float mn = m0;
            if (ok) mn = fmaxf(mn, Sa[h]);
            if (PAIR && rok) mn = fmaxf(mn, rh);
            const float c_ = fast_exp(m0 - mn);
            const float a_ = ok ? fast_exp(Sa[h] - mn) : 0.f;
            const float b_ = (PAIR && rok) ? fast_exp(rh - mn) : 0.f;
            if constexpr (X::LDSS) {
                stc[h * 256] = mn;
                stc[(16 + h) * 256] = fmaf(stc[(16 + h) * 256], c_, a_ + b_);
                stc[(32 + h) * 256] = c_; stc[(48 + h) * 256] = a_; stc[(64 + h) * 256] = b_;
            } else {
                sl[h] = fmaf(sl[h], c_, a_ + b_);
                sm[h] = mn; sc[h] = c_; p1[h] = a_; p2[h] = b_;
            }
        }
        APT(4);
        // ---- messages: T1 = tanh(lin_edge1 x) once; own direction v_j * T1, partner's direction v_i * T1 ----
        const int rslot = half * ATT_LANES + rln;
        if constexpr (!PREF) {
            bload16(vj, X::MB0, vjn);
            if (PAIR) bload16(vi, X::MB0, vin);
        }
#pragma unroll
        for (int b = X::MB0; b < X::MB1; ++b) {
            float vv[16], vo[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) { vv[s] = vjn[s]; vo[s] = PAIR ? vin[s] : 0.f; }
            // the next block's v rows are requested BEHIND the weight prefetch of the next block (loads return in order: a
            // gather ahead of the prefetch would have to land before the next block's MFMAs may start, behind it only
            // before their epilogue)
            auto next_rows = [&]() {
                if (b + 1 < X::MB1) {
                    bload16(vj, b + 1, vjn);
                    if (PAIR) bload16(vi, b + 1, vin);
                }
            };
            if (PAIR && X::PHB > 1 && b > 0 && b % X::PHB == 0) __syncthreads();      // the previous phase has been read everywhere
            const unsigned cur = w.oL1 + (unsigned)(b * X::KQE) * 1024;
            f32x16 acc;
            if constexpr (X::SPLIT) { next_rows(); pipeline_fence(); acc = splitc::block4(w.T, w.g, xs, zero16()); }
            else acc = mfma_block_p2<X::KQE>(w.wp, w.ws, cur, w.ws, b + 1 < X::ND ? cur + X::KQE * 1024 : ring0, x, zero16(), next_rows);
            float T[16];
            tanh16(acc, T);
            float sc_lo, sc_hi, p1_lo, p1_hi, p2_lo = 0.f, p2_hi = 0.f;
            if constexpr (X::LDSS) {
                attn_pick_lds<X>(stc + 32 * 256, b, half, sc_lo, sc_hi);
                attn_pick_lds<X>(stc + 48 * 256, b, half, p1_lo, p1_hi);
                if (PAIR) attn_pick_lds<X>(stc + 64 * 256, b, half, p2_lo, p2_hi);
            } else if constexpr (X::PH) {               // block b = head 2b + half = slot b
                sc_lo = sc_hi = sc[b]; p1_lo = p1_hi = p1[b];
                if (PAIR) p2_lo = p2_hi = p2[b];
            } else {
                attn_pick<X>(sc, b, half, sc_lo, sc_hi);
                attn_pick<X>(p1, b, half, p1_lo, p1_hi);
                if (PAIR) attn_pick<X>(p2, b, half, p2_lo, p2_hi);
            }
            if (PAIR) {                                 // this atom's unweighted message for the partner's accumulator
                float4* ub = ux + (X::PHB == 1 ? (b & 1) : (b % X::PHB)) * 4 * 256;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 u0 = pk2(T[q * 4 + 0], T[q * 4 + 1]) * pk2(vo[q * 4 + 0], vo[q * 4 + 1]);
                    const f32x2 u1 = pk2(T[q * 4 + 2], T[q * 4 + 3]) * pk2(vo[q * 4 + 2], vo[q * 4 + 3]);
                    ub[q * 256 + slot] = make_float4(u0.x, u0.y, u1.x, u1.y);
                }
            }
#pragma unroll
            for (int s = 0; s < 16; s += 2) {           // own source (element pairs on the packed pipe; a pair never straddles register 8)
                const f32x2 m = pk2(T[s], T[s + 1]) * pk2(vv[s], vv[s + 1]) * (s < 8 ? p1_lo : p1_hi);
                const f32x2 a = __builtin_elementwise_fma(pk2(macc[b * 16 + s], macc[b * 16 + s + 1]), (f32x2)(s < 8 ? sc_lo : sc_hi), m);
                macc[b * 16 + s] = a.x; macc[b * 16 + s + 1] = a.y;
            }
            if (PAIR && (b + 1) % X::PHB == 0) {        // end of a hand-over phase: the partners' messages of its blocks
                __syncthreads();
#pragma unroll
                for (int bb = b + 1 - X::PHB; bb <= b; ++bb) {
                    const float4* ub = ux + (X::PHB == 1 ? (bb & 1) : (bb % X::PHB)) * 4 * 256;
                    float q_lo = p2_lo, q_hi = p2_hi;
                    if constexpr (X::PHB > 1) {
                        if constexpr (X::PH) { q_lo = q_hi = p2[bb]; }
                        else if constexpr (X::LDSS) attn_pick_lds<X>(stc + 64 * 256, bb, half, q_lo, q_hi);
                        else attn_pick<X>(p2, bb, half, q_lo, q_hi);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 r4 = ub[q * 256 + rslot];
                        const f32x2 w2 = (f32x2)(q < 2 ? q_lo : q_hi);
                        const f32x2 m0 = __builtin_elementwise_fma(w2, pk2(r4.x, r4.y), pk2(macc[bb * 16 + q * 4 + 0], macc[bb * 16 + q * 4 + 1]));
                        const f32x2 m1 = __builtin_elementwise_fma(w2, pk2(r4.z, r4.w), pk2(macc[bb * 16 + q * 4 + 2], macc[bb * 16 + q * 4 + 3]));
                        macc[bb * 16 + q * 4 + 0] = m0.x; macc[bb * 16 + q * 4 + 1] = m0.y;
                        macc[bb * 16 + q * 4 + 2] = m1.x; macc[bb * 16 + q * 4 + 3] = m1.y;
                    }
                }
            }
        }
    }
    APT(5);
    // ---- partial of this item: unnormalised sums + (max, sum) per head ----
    if (L.valid) {
        float* hrow = A.hhat + ((size_t)L.v * A.pd.amax_parts + part) * D;
        if constexpr (X::SPLIT) {                       // this launch's four message blocks and head slots of the partial
#pragma unroll
            for (int b = X::MB0; b < X::MB1; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    reinterpret_cast<float4*>(hrow + b * 32 + half * 16)[q] = make_float4(macc[b * 16 + q * 4], macc[b * 16 + q * 4 + 1], macc[b * 16 + q * 4 + 2], macc[b * 16 + q * 4 + 3]);
        } else {
            store_nat<X::ND>(hrow, half, macc);
        }
        float* sp = A.astat + ((size_t)L.v * A.pd.amax_parts + part) * 32;       // [16 maxima | 16 sums], head-indexed
        if constexpr (X::PH) {
#pragma unroll
            for (int k = X::SL0; k < X::SL0 + X::NSL; ++k) { sp[2 * k + half] = sm[k]; sp[16 + 2 * k + half] = sl[k]; }
        } else {
            float fin[16];                              // half 0 stores the maxima, half 1 the sums (both halves hold both)
#pragma unroll
            for (int h = 0; h < 16; ++h) {
                if constexpr (X::LDSS) fin[h] = stc[((half ? 16 : 0) + h) * 256];
                else fin[h] = half ? sl[h] : sm[h];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                reinterpret_cast<float4*>(sp)[(half ? 4 : 0) + q] = make_float4(fin[q * 4 + 0], fin[q * 4 + 1], fin[q * 4 + 2], fin[q * 4 + 3]);
        }
    }
    APT(6);
#ifdef JODO_PHASE_TIMING_ATTN
    if constexpr (X::SPLIT) pt_acc[7] = w.T.bar;        // (inside the phases above: the ring's chunk boundaries, commit + barrier)
#endif
    APT_FLUSH;
}

// PAIR = true: the pair launch (ai_* items), exits when the inputs are asymmetric.  Its items are pair-mode items of groups of
// whole molecules and — under the plan's persistent schedule — the directed-mode items of molecules larger than a group
// (ai_dir: every lane visits all its sources, no hand-over), so that symmetric inputs need one launch whatever the sizes.
// PAIR = false: the directed launch (ad_* items), runs when the inputs are asymmetric (and for ad_big items: molecules larger than
// a group when the pair launch does not carry them, i.e. fixed-chunk plans)
#ifndef JODO_X_ATT_SPLIT_OCC
#define JODO_X_ATT_SPLIT_OCC(VAR) 1
#endif
template <int D, bool WQK, bool PAIR, int VAR = 0>
__global__ __launch_bounds__(ATT_WAVES * 64, JODO_X_ATT_SPLIT_OCC(VAR)) void k_edge_attn(KArgs A) {
    using X = AttnT<D, WQK, VAR>;
    const bool asym = A.flags[FLAG_ASYM] != 0;
    if (PAIR ? asym : !(asym || A.pd.ad_big[blockIdx.x])) return;
    // pair mode under the plan's wrap-around schedule: this workgroup is slot blockIdx.x and works through its items (the
    // resident weights below are staged once); otherwise one item per workgroup
    const bool pers = PAIR && A.pd.a_persist != 0;
    const int it0 = pers ? A.pd.aw_off[blockIdx.x] : (int)blockIdx.x, it1 = pers ? A.pd.aw_off[blockIdx.x + 1] : (int)blockIdx.x + 1;
    __shared__ float4 wl[(X::LDS_EE ? 32 * 64 : 0) + (X::LDS_L0 ? (X::LDS_TAIL ? 64 : 56) * 64 : 0) + (X::LDS_EE && !X::LDS_TAIL ? 0 : 1)];   // edge_emb (2 x 16 quads) | lin_edge0 (8 x 8 quads)
    __shared__ float4 ringc[X::SPLIT ? splitc::RING_SLOTS * splitc::CH_BYTES / 16 : 1];   // SPLIT: the shared weight ring (36 KiB)
    __shared__ float4 sx[PAIR ? 2 * 256 : 1];            // scores handed to the partner: [quad][half * 128 + lane], heads 8h .. 8h + 7
    __shared__ float4 ux[PAIR ? (X::PHB == 1 ? 2 : X::PHB) * 4 * 256 : 1];   // unweighted messages: one block double buffered, or a phase of PHB blocks
    __shared__ float stt[X::LDSS ? 5 * 16 * 256 : 1];    // per thread: running max, sum | rescale, p(own source), p(handed-over source)
    if constexpr (X::LDS_EE) stage_weights<32, ATT_WAVES>(wl, reinterpret_cast<const float4*>(A.W + A.wb[JB_EE_W]));
    if constexpr (X::LDS_L0) stage_weights<(X::LDS_TAIL ? 64 : 56), ATT_WAVES>(wl + (X::LDS_EE ? 32 * 64 : 0), reinterpret_cast<const float4*>(A.W + A.wb[JB_LE0_W]));
    if constexpr (X::LDS_EE || X::LDS_L0) __syncthreads();
    for (int it = it0; it < it1; ++it) {
        if (it > it0) __syncthreads();                  // the hand-over buffers of the previous item have been read everywhere
        if constexpr (PAIR) {
            const int grp = A.pd.ai_group[it], t0 = A.pd.ai_t0[it], t1 = A.pd.ai_t1[it], part = A.pd.ai_part[it];
            const float4* wl_ = X::SPLIT ? ringc : wl;
            if constexpr (X::SPLIT) {                   // (the launcher takes this variant only for plans without directed-mode items in the pair
                attn_item<X, true>(A, wl_, sx, ux, stt, grp, t0, t1, part);      // launch: a second item body doubles the kernel and spills)
            } else {
            if (A.pd.ai_dir[it]) attn_item<X, false>(A, wl_, sx, ux, stt, grp, t0, t1, part);
            else attn_item<X, true>(A, wl_, sx, ux, stt, grp, t0, t1, part);
            }
        } else {
            attn_item<X, false>(A, X::SPLIT ? ringc : wl, sx, ux, stt, A.pd.ad_group[it], A.pd.ad_t0[it], A.pd.ad_t1[it], A.pd.ad_part[it]);
        }
    }
}

// merge the attention partials of a node (k_node_post*): hhat = sum_p acc_p e^{m_p - M} / (sum_p l_p e^{m_p - M} + 1e-16),
// fixed order.  hh: this half-lane's D / 2 message features.  (layers.py:178: PyG softmax = e / (sum + 1e-16))
template <int D>
__device__ __forceinline__ void attn_merge(const KArgs& A, int v, int half, float (&hh)[D / 2]) {
    using X = AttnT<D, true>;
    const int parts = A.pd.anode_parts[v];
    const float* sbase = A.astat + (size_t)v * A.pd.amax_parts * 32;
    const float* hbase = A.hhat + (size_t)v * A.pd.amax_parts * D;
    float M[16], Ls[16];
#pragma unroll
    for (int h = 0; h < 16; ++h) { M[h] = ATT_NEG; Ls[h] = 0.f; }
    for (int q = 0; q < parts; ++q) {
        float m[16];
        load16(sbase + (size_t)q * 32, m);
#pragma unroll
        for (int h = 0; h < 16; ++h) M[h] = fmaxf(M[h], m[h]);
    }
#pragma unroll
    for (int s = 0; s < D / 2; ++s) hh[s] = 0.f;
    for (int q = 0; q < parts; ++q) {
        float m[16], l[16], wq[16];
        load16(sbase + (size_t)q * 32, m);
        load16(sbase + (size_t)q * 32 + 16, l);
#pragma unroll
        for (int h = 0; h < 16; ++h) { wq[h] = fast_exp(m[h] - M[h]); Ls[h] = fmaf(l[h], wq[h], Ls[h]); }
#pragma unroll
        for (int b = 0; b < X::ND; ++b) {
            float t[16], lo, hi;
            load16(hbase + (size_t)q * D + b * 32 + half * 16, t);
            attn_pick<X>(wq, b, half, lo, hi);
#pragma unroll
            for (int s = 0; s < 16; ++s) hh[b * 16 + s] = fmaf(t[s], s < 8 ? lo : hi, hh[b * 16 + s]);
        }
    }
    float inv[16];
#pragma unroll
    for (int h = 0; h < 16; ++h) inv[h] = 1.f / (Ls[h] + 1e-16f);
#pragma unroll
    for (int b = 0; b < X::ND; ++b) {
        float lo, hi;
        attn_pick<X>(inv, b, half, lo, hi);
#pragma unroll
        for (int s = 0; s < 16; ++s) hh[b * 16 + s] *= s < 8 ? lo : hi;
    }
}

}  // namespace jd
