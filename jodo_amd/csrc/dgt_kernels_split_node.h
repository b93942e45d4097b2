// OPT-IN split-bf16 form of k_node_post (JODO_OPT_SPLIT_BF16; tuned nf = 256 kernel set; never the default, never the headline).
//
// Same work per strip of 32 atoms, same formulas, same stores as k_node_post<R> (dgt_kernels_node.h; the reference: the node side of
// EquivariantMixBlock.forward, models/mol_gnn.py:296-311, and the per-node halves of equi_update.input_lin / node2edge_lin that the
// strip model moved from the edges to the nodes): attention merge -> node2edge_lin -> gated residual + LayerNorm + modulate -> node FFN
// -> rotated W_row / W_col images -> readout -> (with >= 1024 strips) the next block's q / k / v.  Every projection runs in the
// three-term split form (dgt_split.h): 1 216 K16 steps = 7 296 bf16 MFMAs x 32 cycles = 233 k matrix cycles per strip at mlp_ratio 2
// against 9 728 fp32 MFMAs x 64 = 623 k.  Four strips per workgroup share one weight tape through an LDS ring (dgt_kernels_split.h
// Tape2; the node kernels stream 3.7 MB of split weights per strip).  One wave per SIMD: the node kernels carry h's split image (192
// registers) and the FFN accumulators (128).
//
// MEASURED (MI355X, round 6; profiles/r06_split_node_*.txt): the node class goes 640 -> 603 us per block at QM9 B = 2500 (k_node_post 326 +
// k_node_mix 324 -> two split launches 229 + 217, k_node_ab 131 and the Gram tiles 35 as launches of their own) and 507 -> 397 at GEOM
// B = 512 — far from the 2.7 x of the matrix cycles.  What the SQ counters say: the waves sit parked 44 % of their cycles and the matrix
// pipe is busy 36 %; a wave issues one MFMA per 56 cycles instead of 33.  Tried, each measured: fragments read a step ahead (kept), a
// second stage set (two periods of prefetch distance: 4 %), eight-step chunks (half the barriers: 2 %), two launches so that a
// launch's tape fits the XCD's L2 (kept: 6 %), two alternating accumulators (none), two workgroups per CU for the second launch
// (276 B of scratch per lane: not run).  The pair update's split kernel got its factor from two waves per SIMD (vector work of one under
// the MFMAs of the other); this kernel's 440 live registers do not allow that, and a single in-order wave serialises its MFMA chain with
// everything else it issues.  Opt-in like the pair kernel; the default path is the exact-fp32 k_node_post.
#pragma once
#include "dgt_kernels_node.h"
#include "dgt_kernels_split.h"

namespace jd {
namespace split {

// 8 values back out of their split image: hi + mid + lo is the value itself, exactly.  The packed words pass through an opaque asm first:
// otherwise the compiler recognises float(hi) / float(mid) as values it already computed while SPLITTING (the residuals x - float(hi))
// and keeps all of them alive across the FFN instead of re-deriving them from the packed image — 900 B of scratch per lane (seen in the ISA).
__device__ __forceinline__ void unsplit8(const Split8& p, float* x) {
    u32x4 h = __builtin_bit_cast(u32x4, p.h), m = __builtin_bit_cast(u32x4, p.m), l = __builtin_bit_cast(u32x4, p.l);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        unsigned hw = h[w], mw = m[w], lw = l[w];
        asm volatile("" : "+v"(hw), "+v"(mw), "+v"(lw));
        x[2 * w] = (__uint_as_float(hw << 16) + __uint_as_float(mw << 16)) + __uint_as_float(lw << 16);
        x[2 * w + 1] = (__uint_as_float(hw & 0xffff0000u) + __uint_as_float(mw & 0xffff0000u)) + __uint_as_float(lw & 0xffff0000u);
    }
}
template <int NR>                                    // NR / 8 Split8 -> NR fp32 registers
__device__ __forceinline__ void unsplit_regs(const Split8* in, float (&x)[NR]) {
#pragma unroll
    for (int g = 0; g < NR / 8; ++g) unsplit8(in[g], &x[8 * g]);
}

// PART: 1 = attention merge .. node FFN (h' to memory), 2 = h' from memory .. rotated W_row / W_col, readout, next q / k / v.
// Why two launches: the node tape of a block is 3.65 MB of split weights (mlp_ratio 2), the L2 of an XCD 4 MiB.  The workgroups of a
// launch drift apart (the merge's partial counts differ, stores stall differently), so their common working set is the WHOLE tape; in one
// launch it did not stay in L2 next to the launch's 379 MB of output stores and the stream came over the fabric at 2.6 TB/s — the waves
// sat parked 44 % of their cycles whatever the chunk size or prefetch distance (SQ counters, tools/gpu_split_pmc.sh), 498 us per launch
// at QM9 B = 2500, no faster than the fp32 kernels.  Each half's tape (1.6 / 2.0 MB) fits.
template <int R, int PART>
__global__ __launch_bounds__(SPLIT_WAVES * 64, 1) void k_node_post_split(KArgs A) {
    static_assert(PART == 1 || PART == 2, "two launches");
    if (A.flags[FLAG_ASYM] || !A.flags[FLAG_UNIFORM_T]) return;      // pinned paths only (the launcher checks; a violated pin is reported by k_finalize_nodes)
    constexpr int STEPS_NOQKV = 2 * 16 + R * 4 * (2 * 16 + 8 * 4) + 16 * 16 + 2 * 16, STEPS_QKV = 24 * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    int strip = (int)blockIdx.x * SPLIT_WAVES + wave;
    const bool live = strip < A.pd.n_strips;         // an idle wave walks strip 0 without stores: the ring needs all four waves
    if (!live) strip = 0;
    const LaneNode L = lane_node(A, strip, j);
    const float* mr = mod_row(A, L.b) + A.mod_base;
    const float* ng2 = mr + 5 * 256;
    __shared__ u32x4 ring[N_SLOTS * N_CH_BYTES / 16];
    Tape2 T;
    T.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A.wsplit_node), 0, 0x7fffffff, 0x00020000);
    constexpr int STEPS_P1 = 2 * 16 + R * 4 * (2 * 16 + 8 * 4);                  // node2edge + FFN
    static_assert(STEPS_P1 % N_CHS == 0, "the second launch starts on a chunk boundary");
    T.ntot = PART == 1 ? STEPS_P1 / N_CHS : (STEPS_NOQKV + (A.fuse_next ? STEPS_QKV : 0)) / N_CHS;
    T.ld_off = (unsigned)wave * (unsigned)(N_CH_BYTES / SPLIT_WAVES) + (unsigned)lane * 16u;
    T.rd_off = (unsigned)lane * 16u;
    T.ring = reinterpret_cast<char*>(ring);
    int g = PART == 1 ? 0 : STEPS_P1 / N_CHS;
    tape2_start(T, g);
    float hx[128];                                    // first the messages, then (in place) the FFN input
    Split8 hs[16];
    if constexpr (PART == 1) {
    node_load_hh(A, L, half, hx);
    split_regs<128>(hx, hs);
    // node2edge_lin applied per node (bias added on the edge side)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const f32x16 acc = tape2_block<16, 0>(T, g, hs, zero16());
        float r[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = acc[s];
        if (live) store16T(A.n2e, 2, L.v, half, b, r);
    }
    node_residual_ln(A, L, half, mr, hx);
    split_regs<128>(hx, hs);                          // from here to the end of the FFN h lives as its split image only (registers)
    {
        f32x16 o[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) o[b] = zero16();
        const float* b1 = A.W + A.wb[JB_FF1_B];
#pragma unroll 1
        for (int c = 0; c < R * 4; ++c) {
            Split8 hh[4];
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                float bb[16], hid[16];
                load16(b1 + (c * 2 + b2) * 32 + half * 16, bb);
                const f32x16 acc = tape2_block<16, 0>(T, g, hs, zero16());
                silu_bias16(acc, bb, hid);
                hh[2 * b2] = split8(&hid[0]);
                hh[2 * b2 + 1] = split8(&hid[8]);
            }
            static_for<8>([&](auto obc) { constexpr int ob = decltype(obc)::value; o[ob] = tape2_block<4, 4 * (ob & 1), false>(T, g, hh, o[ob]); });
        }
        // h' = x + ng2 (FFN(x) + b2), block by block: the FFN input comes back out of its split image (hi + mid + lo is the value, bit for
        // bit), h' goes to memory and straight back into the split image — the 128 fp32 registers of h are not held beside it
        // (opaque pointers + fence: left visible, the 256 bias / gate values below are hoisted above the FFN loop and spilled across it)
        pipeline_fence();
        const float* b2 = launder(A.W) + A.wb[JB_FF2_B];
        const float* ng2_ = launder(ng2);
        float* hrow = A.h + (size_t)L.v * 256;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float bb[16], gg[16], x16[16];
            load16(b2 + b * 32 + half * 16, bb);
            load16(ng2_ + b * 32 + half * 16, gg);
            unsplit8(hs[2 * b], &x16[0]);
            unsplit8(hs[2 * b + 1], &x16[8]);
#pragma unroll
            for (int s = 0; s < 16; ++s) x16[s] = fmaf(gg[s], o[b][s] + bb[s], x16[s]);
            if (live) store16(hrow + b * 32 + half * 16, x16);
        }
    }
    } else {
    load_nat<8>(A.h + (size_t)L.v * 256, half, hx);   // h' of the first launch
    split_regs<128>(hx, hs);
    // per-node halves of equi_update.input_lin in the rotated basis: Q P (W_row h + b), Q P W_col h
    {
        const float* bin = A.W + A.wb[JB_INQ_B];
#pragma unroll 1
        for (int b = 0; b < 8; ++b) {
            float bb[16], r[16];
            load16(bin + b * 32 + half * 16, bb);
            f32x16 acc = tape2_block<16, 0>(T, g, hs, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) r[s] = acc[s] + bb[s];
            if (live) store16T(A.wrow, 8, L.v, half, b, r);
            acc = tape2_block<16, 0>(T, g, hs, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) r[s] = acc[s];
            if (live) store16T(A.wcol, 8, L.v, half, b, r);
        }
    }
    // readout node_l(h) -> atom_hids[:, D + l * 64 ...]
    {
        const float* bias = A.W + A.wb[JB_NRO_B];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float bb[16], r[16];
            load16(bias + b * 32 + half * 16, bb);
            const f32x16 acc = tape2_block<16, 0>(T, g, hs, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) r[s] = acc[s] + bb[s];
            if (live) store16(A.ahid + (size_t)L.v * A.d.KNH + 256 + A.layer * 64 + b * 32 + half * 16, r);
        }
    }
    if (A.fuse_next) {                                // next block's LN1 + modulate + q / k / v, h' still in registers (as its split image)
        unsplit_regs<128>(hs, hx);
        node_next_ln(A, L, half, hx);
        split_regs<128>(hx, hs);
#pragma unroll 1
        for (int gq = 0; gq < 24; ++gq) {
            const int piece = gq >> 3, b = gq & 7;
            const float* bias = A.W + A.wbn[2 * piece + 1];
            float* outp = piece == 0 ? A.q : (piece == 1 ? A.k : A.v);
            float bb[16], r[16];
            load16(bias + b * 32 + half * 16, bb);
            const f32x16 acc = tape2_block<16, 0>(T, g, hs, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) r[s] = acc[s] + bb[s];
            if (live) store16T(outp, 8, L.v, half, b, r);
        }
    }
    }
}

// OPT-IN split-bf16 form of k_node_ab under the rotated statistics (wide::node_ab_body, rot branch): the per-node rows of the coord_mlp.0
// hoist, A' = F (Q P R_a) -> A.ua and B' = F (Q P C_c) -> A.ub, F = W0 diag(1 + sc) Q^T of this forward (k_fold_coord wrote its split image
// A.ffold_s beside the fp32 one: the same matrix as three terms).  One (strip, piece) item per wave as in the fp32 kernel; the four items
// of a workgroup share F's 128-step tape (384 KiB per block) through the ring — the fp32 kernel reads its 256 KiB per item from L2,
// 2 x n_strips times (5.6 TB/s at QM9 B = 2500: that stream is what bounds it).
// Measured (MI355X, QM9 B = 2500, rocprofv3): fp32 k_node_ab 131 us per launch -> 90 us in this form.  Equal: the row's split image in
// registers (192, one wave per SIMD).  Slower: both pieces of a strip per wave through the same fragments (two independent accumulator
// chains, half the LDS reads) — 128 us with the operands converted in the loop (one wave per SIMD: nothing hides 96 conversions per
// step), 120 us with both rows' split images in registers (384 + accumulators: 212 B of scratch).
// Faster alone, slower in the step: a K-major walk (k_fold_coord writing F's image as [step / 4][out block][step % 4]; 64 features of the row
// converted once and applied to all eight output blocks, eight accumulators, 248 registers, two workgroups per CU) runs 69 us — and the
// step around it 12.50 -> 12.72 ms on the same box, three alternations: the pair update 392 -> 409 us and the attention kernel 444 -> 461
// with it (the denser launch costs the launches around it ~ 4 %; power, by every sign).  Not kept.
__global__ __launch_bounds__(SPLIT_WAVES * 64, 2) void k_node_ab_split(KArgs A) {
    if (A.flags[FLAG_ASYM] || !A.flags[FLAG_UNIFORM_T]) return;      // pinned paths only (the launcher checks)
    constexpr int D = 256, ND = D / 32;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    int item = A.ab0 + (int)blockIdx.x * SPLIT_WAVES + wave;
    const bool live = item < A.ab1;                  // an idle wave walks the first item without stores: the ring needs all four waves
    if (!live) item = A.ab0;
    const int strip = item >> 1, piece = item & 1;
    const LaneNode L = lane_node(A, strip, j);
    __shared__ u32x4 ring[N_SLOTS * N_CH_BYTES / 16];
    Tape2 T;
    T.rs = __builtin_amdgcn_make_buffer_rsrc(A.ffold_s + (size_t)A.layer * D * D * 3, 0, 0x7fffffff, 0x00020000);
    T.ntot = ND * (D / 16) / N_CHS;
    T.ld_off = (unsigned)wave * (unsigned)(N_CH_BYTES / SPLIT_WAVES) + (unsigned)lane * 16u;
    T.rd_off = (unsigned)lane * 16u;
    T.ring = reinterpret_cast<char*>(ring);
    int g = 0;
    tape2_start(T, g);
    // The row stays in fp32 (128 registers) and every K16 step's operand is split where it is used, once per output block: its split
    // image would be 192 registers — one wave per SIMD, and a single in-order wave gets an MFMA out every 56 cycles (k_node_post_split).
    // With 256 registers two workgroups share a CU and one wave's conversions run under the other's MFMAs.
    float x[D / 2];
    {
        const TRow src = trow(piece == 0 ? A.wrow : A.wcol, ND, L.v, half);
#pragma unroll
        for (int b = 0; b < ND; ++b) {
            float t[16];
            load16T(src, b, t);
#pragma unroll
            for (int s = 0; s < 16; ++s) x[b * 16 + s] = t[s];
        }
    }
    float* dst = piece == 0 ? A.ua : A.ub;
#pragma unroll 1
    for (int b = 0; b < ND; ++b) {
#pragma unroll
        for (int s = 0; s < D / 2; ++s) asm volatile("" : "+v"(x[s]));      // (keeps the conversions inside the loop: hoisted they are the 192 registers again)
        f32x16 acc = zero16();
#pragma unroll
        for (int c = 0; c < D / 16 / N_CHS; ++c) {   // a chunk of the tape = eight steps = 64 features of the row, converted four steps at a time
            static_assert(N_CHS == 8, "two half-chunks");
            Split8 xs[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xs[q] = split8(&x[(c * 8 + q) * 8]);
            acc = tape2_block<4, 0>(T, g, xs, acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) xs[q] = split8(&x[(c * 8 + 4 + q) * 8]);
            acc = tape2_block<4, 4>(T, g, xs, acc);
        }
        float r[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = acc[s];
        if (live) store16T(dst, ND, L.v, half, b, r);
    }
}

}  // namespace split
}  // namespace jd
