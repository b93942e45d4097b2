// Split-bf16 ("bf16x3") form of the strip model's projections (gfx950 / CDNA4 only) — OPT-IN, never the default path.
//
// On gfx950 the f32-input MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate on the vector lanes: 64 cycles per SIMD for
// 4 096 flop, and vector work issued beside it does not overlap.  v_mfma_f32_32x32x16_bf16 does 32 768 flop in 32
// cycles on the matrix pipe proper, with vector issue underneath it.  A float is the exact sum of three bf16 numbers
// (x = hi + mid + lo, 8 + 8 + 8 significand bits), a bf16 x bf16 product is exact in fp32, so
//     w * x = wh*xh + wh*xm + wm*xh + wm*xm + wh*xl + wl*xh          (dropped: wm*xl, wl*xm, wl*xl  <=  3 * 2^-26 |w x|)
// accumulated in fp32 is fp32-equivalent arithmetic at 6 bf16 MFMAs per K = 16 step (192 cycles) against 8 f32 MFMAs (512 cycles).
//
// Lane maps (same 32 x 32 accumulator image as the f32 form, so chains still stay in registers):
//   A (weights):      lane l supplies W[row(l & 31)][k = 8 * (l >> 5) + j], j = 0..7      (one 16-byte load per split term)
//   B (activations):  lane l supplies X[k = 8 * (l >> 5) + j][item l & 31]                (8 consecutive activation registers)
// K = 16 step G of a projection therefore consumes activation registers 8 G .. 8 G + 7 of BOTH half-lanes — the same 16 input
// features the f32 k-steps 8 G .. 8 G + 7 consume, so the packer's in_map / out_map are unchanged (dgt_pack.cpp pack_proj_split).
// Packed layout: [out block][K16 step][term: hi, mid, lo][64 lanes][8 bf16]  (3 KiB per block and step).
#pragma once
#include "dgt_device.h"

namespace jd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Split8 { bf16x8 h, m, l; };           // 8 activation values as three bf16 terms (12 VGPRs)

// x[0..7] -> hi + mid + lo, round-to-nearest-even at every term (v_cvt_pk_bf16_f32; the residuals are exact in fp32)
__device__ __forceinline__ Split8 split8(const float* x) {
    Split8 s;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        const float r1 = x[j] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        s.h[j] = h; s.m[j] = m; s.l[j] = (__bf16)r2;
    }
    return s;
}

__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// one split term (0 hi, 1 mid, 2 lo) of K16 step `st` of the block that starts at byte offset `soff`
__device__ __forceinline__ u32x4 wload_s(const WSrc& w, unsigned soff, int st, int term) {
    return __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.voff, soff + (unsigned)(st * 3 + term) * 1024u, 0);
}

template <int PG>          // K16 steps per prefetch group: PG * 6 MFMAs (PG * 192 cycles) of cover per group and item tile
struct WPipeS {
    u32x4 q[PG][3];
};

template <int PG>
__device__ __forceinline__ void wpipe_prime_s(WPipeS<PG>& p, const WSrc& w, unsigned soff) {
#pragma unroll
    for (int i = 0; i < PG; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t) p.q[i][t] = wload_s(w, soff, i, t);
}

// the six products of one K16 step into one accumulator, small terms first
__device__ __forceinline__ f32x16 mfma_step_s(const bf16x8& wh, const bf16x8& wm, const bf16x8& wl, const Split8& x, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, x.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, x.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, x.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, x.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, x.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, x.h, acc, 0, 0, 0);
    return acc;
}

// The same six products dealt to TWO accumulators, alternating: neighbouring MFMAs are then independent.  On gfx950 an instruction issued
// between two MFMAs on the SAME accumulator costs about 43 cycles (the forwarding path of a dependent chain is lost; MI355X_MICROARCH.md,
// per-instruction constants), between MFMAs on different accumulators about 6 — and a K16 step has its ds_reads and address arithmetic to
// place somewhere.  Measured on k_node_post_split: 56 cycles per MFMA with one accumulator (three such gaps per step).  The caller adds the
// two halves once per output block.
__device__ __forceinline__ void mfma_step_s2(const bf16x8& wh, const bf16x8& wm, const bf16x8& wl, const Split8& x, f32x16& a, f32x16& b) {
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, x.l, a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, x.h, b, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, x.m, a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, x.m, b, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, x.h, a, 0, 0, 0);
    b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, x.h, b, 0, 0, 0);
}

// One output block of a split projection over NS K16 steps, weights streamed through the ring (the split-form counterpart of
// mfma_block_p2): TILES item tiles share every weight fragment.  act[t][g] = Split8 of tile t, step g.  NS % PG == 0.
template <int NS, int PG, int TILES, typename After = NoHook>
__device__ __forceinline__ void mfma_block_s(WPipeS<PG>& p, const WSrc& w, unsigned cur_off, const WSrc& wn, unsigned next_off,
                                             const Split8 (&act)[TILES][NS], f32x16 (&acc)[TILES], After&& after = NoHook()) {
    static_assert(NS % PG == 0, "block length must be a multiple of the prefetch group");
#pragma unroll
    for (int g = 0; g < NS / PG; ++g) {
        u32x4 cur[PG][3];
#pragma unroll
        for (int i = 0; i < PG; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t) cur[i][t] = p.q[i][t];
        if (g + 1 < NS / PG) {
#pragma unroll
            for (int i = 0; i < PG; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t) p.q[i][t] = wload_s(w, cur_off, (g + 1) * PG + i, t);
        } else {
#pragma unroll
            for (int i = 0; i < PG; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t) p.q[i][t] = wload_s(wn, next_off, i, t);
            after();
        }
        pipeline_fence();
#pragma unroll
        for (int i = 0; i < PG; ++i) {
            const bf16x8 wh = as_bf16x8(cur[i][0]), wm = as_bf16x8(cur[i][1]), wl = as_bf16x8(cur[i][2]);
#pragma unroll
            for (int t = 0; t < TILES; ++t) acc[t] = mfma_step_s(wh, wm, wl, act[t][g * PG + i], acc[t]);
        }
        pipeline_fence();
    }
}

// ---- cyclic weight tape through an LDS ring (the attention kernel's split form: a launch walks the same 48 / 52 K16 steps for every pair offset) ----
// Same protocol as the pair update's ring (dgt_kernels_split.h): chunk g lives in slot g % 3; boundary(g) commits chunk g + 1 from the
// stage registers, barriers, requests chunk g + 2.  Chunk g of the walk is chunk g % period of the tape.
namespace splitc {
constexpr int CH_STEPS = 4, CH_BYTES = CH_STEPS * 3072, RING_SLOTS = 3, WAVES = 4;
struct TapeC {
    __amdgpu_buffer_rsrc_t rs;
    int period, ntot;                                // chunks per pass of the tape, chunks of this walk
    unsigned ld_off, rd_off;
    char* ring;
    u32x4 stage[3];
#ifdef JODO_PHASE_TIMING_ATTN
    unsigned long long bar = 0;                      // cycles between entering a chunk boundary and leaving its barrier
#endif
};
__device__ __forceinline__ void request(TapeC& T, int g) {
    if (g >= T.ntot) return;
    const unsigned base = (unsigned)(g % T.period) * CH_BYTES;
#pragma unroll
    for (int i = 0; i < 3; ++i) T.stage[i] = __builtin_amdgcn_raw_buffer_load_b128(T.rs, T.ld_off + (unsigned)i * 1024u, base, 0);
}
__device__ __forceinline__ void commit(TapeC& T, int g) {
    if (g >= T.ntot) return;
    char* dst = T.ring + (g % RING_SLOTS) * CH_BYTES + T.ld_off;
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(dst + i * 1024) = T.stage[i];
}
__device__ __forceinline__ void start(TapeC& T) {
    request(T, 0);
    commit(T, 0);
    request(T, 1);
    pipeline_fence();
}
// one chunk-aligned block of CH_STEPS steps (every K = De projection block of the attention kernel at nf 256): acc += W_chunk * act
__device__ __forceinline__ f32x16 block4(TapeC& T, int& g, const Split8* act, f32x16 acc) {
#ifdef JODO_PHASE_TIMING_ATTN
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long c0 = __builtin_readcyclecounter();
#endif
    commit(T, g + 1);
    __syncthreads();
#ifdef JODO_PHASE_TIMING_ATTN
    __builtin_amdgcn_sched_barrier(0);
    T.bar += __builtin_readcyclecounter() - c0;
#endif
    request(T, g + 2);
    pipeline_fence();
    const char* base = T.ring + (g % RING_SLOTS) * CH_BYTES + T.rd_off;
#pragma unroll
    for (int w = 0; w < CH_STEPS; ++w) {
        const bf16x8 wh = as_bf16x8(*reinterpret_cast<const u32x4*>(base + w * 3072));
        const bf16x8 wm = as_bf16x8(*reinterpret_cast<const u32x4*>(base + w * 3072 + 1024));
        const bf16x8 wl = as_bf16x8(*reinterpret_cast<const u32x4*>(base + w * 3072 + 2048));
        acc = mfma_step_s(wh, wm, wl, act[w], acc);
    }
    ++g;
    return acc;
}
}  // namespace splitc

}  // namespace jd
