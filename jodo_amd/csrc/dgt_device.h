// Device building blocks for the DGT kernels (gfx950 / CDNA4 only).
//
// Execution model used by every kernel in this directory ("strip" model):
//   * one wavefront (64 lanes) owns 32 items (edges or nodes); item j = lane & 31;
//   * a feature vector of an item is split over the two half-lanes h = lane >> 5 of that item:
//     register R of half h holds feature (R / 16) * 32 + h * 16 + (R % 16)  ("natural-half" slots);
//   * dense projections run on v_mfma_f32_32x32x2_f32 in the transposed orientation
//       D[out_feature, item] += W[out_feature, k] * X[k, item]
//     with the pre-packed weights as A operand (one 16-byte load per lane feeds 4 MFMAs, see
//     csrc/dgt_pack.cpp) and the activation registers as B operand.  The accumulator of an output
//     block *is* the next projection's B operand: chains of projections stay in registers.
//   * per-item reductions over features (LayerNorm, head scores) are in-lane sums plus one
//     exchange with lane ^ 32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

namespace jd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// ---- scalar math (hardware transcendental units; abs error ~1e-7) -------------------------------
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// tanh(x) = 1 - 2 / (1 + e^{2x}): 3 full-rate + 2 transcendental instructions; saturates correctly
// (e^{2x} -> inf gives 1, -> 0 gives -1); absolute error ~1e-7 like the (1 - t) / (1 + t) form it replaces
__device__ __forceinline__ float tanh_f(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);     // e^{2x}
    return fmaf(-2.f, fast_rcp(1.f + e), 1.f);
}
// two at a time on the packed fp32 pipe: v_pk_mul, 2 x v_exp, v_pk_add, 2 x v_rcp, v_pk_fma (3 full-rate instructions per
// PAIR instead of per value; hipcc packs only the final fma by itself)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 tanh_f2(f32x2 x) {
    const f32x2 t = x * 2.8853900817779268f;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
    e = e + 1.f;
    f32x2 r;
    r.x = fast_rcp(e.x); r.y = fast_rcp(e.y);
    return __builtin_elementwise_fma(r, (f32x2)(-2.f), (f32x2)(1.f));
}
// in place on an accumulator block (16 values)
__device__ __forceinline__ void tanh16(const f32x16& acc, float (&T)[16]) {
#pragma unroll
    for (int s = 0; s < 16; s += 2) {
        f32x2 v; v.x = acc[s]; v.y = acc[s + 1];
        v = tanh_f2(v);
        T[s] = v.x; T[s + 1] = v.y;
    }
}
__device__ __forceinline__ f32x2 silu_f2(f32x2 x) {             // x * rcp(1 + e^{-x}), pairs on the packed pipe
    const f32x2 t = x * -1.4426950408889634f;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
    e = e + 1.f;
    f32x2 r;
    r.x = fast_rcp(e.x); r.y = fast_rcp(e.y);
    return x * r;
}
__device__ __forceinline__ f32x2 pk2(float a, float b) { f32x2 v; v.x = a; v.y = b; return v; }
// hid = SiLU(acc + bias) for an accumulator block
__device__ __forceinline__ void silu_bias16(const f32x16& acc, const float (&bb)[16], float* hid) {
#pragma unroll
    for (int s = 0; s < 16; s += 2) {
        const f32x2 v = silu_f2(pk2(acc[s], acc[s + 1]) + pk2(bb[s], bb[s + 1]));
        hid[s] = v.x; hid[s + 1] = v.y;
    }
}
__device__ __forceinline__ float silu_f(float x) { return x * fast_rcp(1.f + fast_exp(-x)); }

// Make a pointer opaque to the optimiser at this program point.  Used at the top of per-edge loops:
// without it LICM hoists every loop-invariant vector (modulation rows, biases, GBF tables: hundreds
// of registers) out of the loop and the kernel spills; re-reading them from L1 each iteration is
// far cheaper than the scratch traffic.
template <typename T>
__device__ __forceinline__ T* launder(T* p) {
    // an opaque zero added to the pointer: the address space (global) stays visible to the compiler,
    // so these remain global_load (not flat_load) instructions
    int z = 0;
    asm volatile("" : "+v"(z));
    return p + z;
}

// Full compiler + scheduler fence.  sched_barrier alone does not order the (readonly, unchained)
// buffer-load intrinsics at instruction-selection time; the "memory" clobber does.
__device__ __forceinline__ void pipeline_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// sum of a per-half partial over the two half-lanes of an item
// gfx950: v_permlane32_swap exchanges the upper half of one register with the lower half of another in the
// VALU (x' = [x.lo, y.lo], y' = [x.hi, y.hi]); with x = y = v the two results are v.lo and v.hi broadcast to
// both halves.  Replaces a ds_bpermute (LDS round trip) per reduction.
__device__ __forceinline__ float pair_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ---- register <-> memory in natural-half slot order -----------------------------------------------
// row points at feature 0 of this lane's item; NB = number of 32-feature blocks.
template <int NB>
__device__ __forceinline__ void load_nat(const float* __restrict__ row, int half, float (&r)[NB * 16]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float4* p = reinterpret_cast<const float4*>(row + b * 32 + half * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = p[q];
            r[b * 16 + q * 4 + 0] = v.x;
            r[b * 16 + q * 4 + 1] = v.y;
            r[b * 16 + q * 4 + 2] = v.z;
            r[b * 16 + q * 4 + 3] = v.w;
        }
    }
}

template <int NB>
__device__ __forceinline__ void store_nat(float* __restrict__ row, int half, const float (&r)[NB * 16]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float4* p = reinterpret_cast<float4*>(row + b * 32 + half * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            p[q] = make_float4(r[b * 16 + q * 4 + 0], r[b * 16 + q * 4 + 1], r[b * 16 + q * 4 + 2],
                               r[b * 16 + q * 4 + 3]);
    }
}

// 16 consecutive floats (one block-half) -> registers
__device__ __forceinline__ void load16(const float* __restrict__ p16, float (&r)[16]) {
    const float4* p = reinterpret_cast<const float4*>(p16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = p[q];
        r[q * 4 + 0] = v.x; r[q * 4 + 1] = v.y; r[q * 4 + 2] = v.z; r[q * 4 + 3] = v.w;
    }
}
__device__ __forceinline__ void store16(float* __restrict__ p16, const float (&r)[16]) {
    float4* p = reinterpret_cast<float4*>(p16);
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = make_float4(r[q * 4 + 0], r[q * 4 + 1], r[q * 4 + 2], r[q * 4 + 3]);
}

// ---- node arrays read by the edge kernels: strip-transposed layout ---------------------------------
// q, k, v, W_row h, W_col h, node2edge(hhat) are stored as [strip][block][quad][lane][4 floats], lane =
// (node & 31) + 32 * half — i.e. exactly the register image of a 32-node strip.  A strip's own rows are
// then read/written with fully coalesced 1-KiB instructions, and partner rows (j = i + d inside the same
// molecule) are a *shifted* contiguous read touching 8-16 cache lines instead of 32 scattered ones
// (row-major rows cost ~250 issue cycles per 16-byte gather instruction and starved the weight stream).
struct TRow {
    const float4* p;     // array base + strip * NB * 256 + lane'
};
__device__ __forceinline__ TRow trow(const float* arr, int NB, int node, int half) {
    TRow r;
    r.p = reinterpret_cast<const float4*>(arr) + (size_t)(node >> 5) * NB * 256 + (node & 31) + 32 * half;
    return r;
}
__device__ __forceinline__ void load16T(const TRow& r, int b, float (&x)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = r.p[(b * 4 + q) * 64];
        x[q * 4 + 0] = v.x; x[q * 4 + 1] = v.y; x[q * 4 + 2] = v.z; x[q * 4 + 3] = v.w;
    }
}
__device__ __forceinline__ void store16T(float* arr, int NB, int node, int half, int b, const float (&x)[16]) {
    float4* p = reinterpret_cast<float4*>(arr) + (size_t)(node >> 5) * NB * 256 + (node & 31) + 32 * half;
#pragma unroll
    for (int q = 0; q < 4; ++q) p[(b * 4 + q) * 64] = make_float4(x[q * 4 + 0], x[q * 4 + 1], x[q * 4 + 2], x[q * 4 + 3]);
}

// Buffer-descriptor variant of TRow.  Plain global loads from the (noalias, read-only) node arrays are not
// held in place by pipeline_fence(): hipcc sinks each one down to its first use and emits load ->
// s_waitcnt vmcnt(0) -> add, one exposed L2 round trip per 16 bytes (seen in the ISA of the pair-update
// kernel: 32 serialised round trips per direction).  Buffer-load intrinsics ARE ordered by the fence, so a
// group of rows can be requested up front and consumed after one wait.
struct BRow {
    __amdgpu_buffer_rsrc_t rs;   // descriptor of the whole array (wave-uniform)
    unsigned voff;               // byte offset of this lane's float4 of (block 0, quad 0)
};
__device__ __forceinline__ BRow brow(const float* arr, int NB, int node, int half) {
    BRow r;
    r.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(arr), 0, 0x7fffffff, 0x00020000);
    r.voff = (unsigned)((((size_t)(node >> 5) * NB * 256) + (node & 31) + 32 * half) * 16);
    return r;
}
__device__ __forceinline__ void bload16(const BRow& r, int b, float (&x)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned int __attribute__((ext_vector_type(4))) v =
            __builtin_amdgcn_raw_buffer_load_b128(r.rs, r.voff, (unsigned)((b * 4 + q) * 1024), 0);
        x[q * 4 + 0] = __uint_as_float(v.x); x[q * 4 + 1] = __uint_as_float(v.y);
        x[q * 4 + 2] = __uint_as_float(v.z); x[q * 4 + 3] = __uint_as_float(v.w);
    }
}

// ---- one output block of a projection -------------------------------------------------------------
// w: this lane's float4 of quad 0 of the block (= block base + lane); KQ quads of 4 k-steps;
// act: KQ*4 activation registers.  acc += W_block * act.
template <int KQ>
__device__ __forceinline__ f32x16 mfma_block(const float4* __restrict__ w, const float (&act)[KQ * 4], f32x16 acc) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const float4 a = w[q * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, act[4 * q + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, act[4 * q + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, act[4 * q + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, act[4 * q + 3], acc, 0, 0, 0);
    }
    return acc;
}

// ---- software-pipelined variant ---------------------------------------------------------------------
// The weight stream is read straight from L2 (it is shared by every wave of the launch and far
// larger than LDS); one 16-byte load per lane feeds 4 MFMAs = 256 matrix-pipe cycles, while an L2
// hit costs a few hundred.  Left to itself hipcc issues load -> s_waitcnt vmcnt(0) -> 4 MFMAs, i.e.
// fully serialised, and materialises a 64-bit VGPR address per load (hundreds of registers, spilled).
// Here:
//   * weights are addressed through ONE buffer descriptor (SGPRs) + a per-lane 32-bit byte offset
//     (lane * 16) + a wave-uniform scalar byte offset per quad (SALU arithmetic, no VGPR addresses);
//   * WPipe keeps one group of PG quads in flight: while the MFMAs of group g run, the loads of
//     group g+1 (or of the first group of the NEXT block, `next`) are already issued;
//   * sched_barrier(0) pins that order.  Requires KQ % PG == 0.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct WSrc {
    __amdgpu_buffer_rsrc_t rs;   // descriptor of the whole packed-weight blob (wave-uniform)
    unsigned voff;               // lane * 16
};

__device__ __forceinline__ WSrc make_wsrc(const float* blob, int lane) {
    WSrc w;
    w.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(blob), 0, 0x7fffffff, 0x00020000);
    w.voff = (unsigned)lane * 16u;
    return w;
}

// quad q (1 KiB each) of the block that starts at byte offset `soff` of the blob
__device__ __forceinline__ float4 wload(const WSrc& w, unsigned soff, int q) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.voff, soff + (unsigned)q * 1024u, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int PG>          // quads per prefetch group: PG * 4 MFMAs (PG * 256 cycles) of cover per group
struct WPipe {
    float4 q[PG];
};

template <int PG>
__device__ __forceinline__ void wpipe_prime(WPipe<PG>& p, const WSrc& w, unsigned soff) {
#pragma unroll
    for (int i = 0; i < PG; ++i) p.q[i] = wload(w, soff, i);
}

// cur: byte offset of this block; next: byte offset of the block that will be consumed after it
template <int KQ, int PG>
__device__ __forceinline__ f32x16 mfma_block_p(WPipe<PG>& p, const WSrc& w, unsigned cur_off, unsigned next_off,
                                               const float (&act)[KQ * 4], f32x16 acc) {
    static_assert(KQ % PG == 0, "block length must be a multiple of the prefetch group");
#pragma unroll
    for (int g = 0; g < KQ / PG; ++g) {
        float4 cur[PG];
#pragma unroll
        for (int i = 0; i < PG; ++i) cur[i] = p.q[i];
        if (g + 1 < KQ / PG) {
#pragma unroll
            for (int i = 0; i < PG; ++i) p.q[i] = wload(w, cur_off, (g + 1) * PG + i);
        } else {
#pragma unroll
            for (int i = 0; i < PG; ++i) p.q[i] = wload(w, next_off, i);
        }
        pipeline_fence();
#pragma unroll
        for (int i = 0; i < PG; ++i) {
            const int k = (g * PG + i) * 4;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].x, act[k + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].y, act[k + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].z, act[k + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].w, act[k + 3], acc, 0, 0, 0);
        }
        pipeline_fence();
    }
    return acc;
}

// Same with the NEXT block living in another buffer (wn): the prefetch of its first group goes through that descriptor.
// `after` is called right after that last prefetch has been issued: the place for loads that are consumed much later
// (per-node row gathers).  Loads return in order, so a slow gather issued BEFORE a weight prefetch delays the wait for
// those weights; issued after it, it delays nothing until its own consumer.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <int KQ, int PG, typename After = NoHook>
__device__ __forceinline__ f32x16 mfma_block_p2(WPipe<PG>& p, const WSrc& w, unsigned cur_off, const WSrc& wn, unsigned next_off,
                                                const float (&act)[KQ * 4], f32x16 acc, After&& after = NoHook()) {
    static_assert(KQ % PG == 0, "block length must be a multiple of the prefetch group");
#pragma unroll
    for (int g = 0; g < KQ / PG; ++g) {
        float4 cur[PG];
#pragma unroll
        for (int i = 0; i < PG; ++i) cur[i] = p.q[i];
        if (g + 1 < KQ / PG) {
#pragma unroll
            for (int i = 0; i < PG; ++i) p.q[i] = wload(w, cur_off, (g + 1) * PG + i);
        } else {
#pragma unroll
            for (int i = 0; i < PG; ++i) p.q[i] = wload(wn, next_off, i);
            after();
        }
        pipeline_fence();
#pragma unroll
        for (int i = 0; i < PG; ++i) {
            const int k = (g * PG + i) * 4;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].x, act[k + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].y, act[k + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].z, act[k + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].w, act[k + 3], acc, 0, 0, 0);
        }
        pipeline_fence();
    }
    return acc;
}

// General form: the block has KQ quads of which the ring holds the first min(PG, KQ); the rest is prefetched in groups of up to PG
// (KQ need not be a multiple of PG), and the last group prefetches NXT (<= PG) quads of the NEXT block (other buffer wn) — what the
// next consumer expects to find in the ring.  Used by the block-upper-triangular projection of the rotated statistics, whose
// blocks shrink by four quads each.  act points at KQ * 4 activation registers.
template <int KQ, int NXT, int PG, typename After = NoHook>
__device__ __forceinline__ f32x16 mfma_block_g(WPipe<PG>& p, const WSrc& w, unsigned cur_off, const WSrc& wn, unsigned next_off,
                                               const float* act, f32x16 acc, After&& after = NoHook()) {
    static_assert(NXT <= PG && KQ > 0, "prefetch group");
    constexpr int NG = (KQ + PG - 1) / PG;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int ng = KQ - g * PG < PG ? KQ - g * PG : PG;
        float4 cur[PG];
#pragma unroll
        for (int i = 0; i < PG; ++i) if (i < ng) cur[i] = p.q[i];
        if (g + 1 < NG) {
            const int nn = KQ - (g + 1) * PG < PG ? KQ - (g + 1) * PG : PG;
#pragma unroll
            for (int i = 0; i < PG; ++i) if (i < nn) p.q[i] = wload(w, cur_off, (g + 1) * PG + i);
        } else {
#pragma unroll
            for (int i = 0; i < NXT; ++i) p.q[i] = wload(wn, next_off, i);
            after();
        }
        pipeline_fence();
#pragma unroll
        for (int i = 0; i < PG; ++i) {
            if (i < ng) {
                const int k = (g * PG + i) * 4;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].x, act[k + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].y, act[k + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].z, act[k + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].w, act[k + 3], acc, 0, 0, 0);
            }
        }
        pipeline_fence();
    }
    return acc;
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// accumulator (+ bias in slot order) -> registers [16]
__device__ __forceinline__ void acc_bias(const f32x16& acc, const float* __restrict__ bias16, float (&r)[16]) {
    float b[16];
    load16(bias16, b);
#pragma unroll
    for (int s = 0; s < 16; ++s) r[s] = acc[s] + b[s];
}

// ---- LayerNorm (no affine, eps 1e-6, biased variance) over NR*2 features of an item ----------------
// (element pairs on the packed fp32 pipe: v_pk_add / v_pk_fma / v_pk_mul do two features per instruction.  Back-to-back
// DEPENDENT packed ops pay a wait state each, so the two reductions run over four independent accumulator pairs)
template <int NR>
__device__ __forceinline__ void layer_norm(float (&x)[NR]) {
    static_assert(NR % 8 == 0, "feature registers come in groups of eight");
    f32x2 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = pk2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NR; i += 8)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = a[k] + pk2(x[i + 2 * k], x[i + 2 * k + 1]);
    const f32x2 st = (a[0] + a[1]) + (a[2] + a[3]);
    const float mean = pair_sum(st.x + st.y) * (1.f / (2 * NR));
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = pk2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NR; i += 8)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x2 v = pk2(x[i + 2 * k], x[i + 2 * k + 1]) - mean;
            a[k] = __builtin_elementwise_fma(v, v, a[k]);
            x[i + 2 * k] = v.x; x[i + 2 * k + 1] = v.y;
        }
    const f32x2 qt = (a[0] + a[1]) + (a[2] + a[3]);
    const float rstd = __builtin_amdgcn_rsqf(pair_sum(qt.x + qt.y) * (1.f / (2 * NR)) + 1e-6f);
#pragma unroll
    for (int i = 0; i < NR; i += 2) { const f32x2 v = pk2(x[i], x[i + 1]) * rstd; x[i] = v.x; x[i + 1] = v.y; }
}

// x = x * (1 + scale) + shift with scale/shift vectors in natural order (NB blocks)
template <int NB>
__device__ __forceinline__ void modulate(float (&x)[NB * 16], const float* __restrict__ shift,
                                         const float* __restrict__ scale, int half) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float sh[16], sc[16];
        load16(shift + b * 32 + half * 16, sh);
        load16(scale + b * 32 + half * 16, sc);
#pragma unroll
        for (int s = 0; s < 16; ++s) x[b * 16 + s] = fmaf(x[b * 16 + s], 1.f + sc[s], sh[s]);
        // (kept scalar on purpose: written with packed pairs hipcc sinks every 16-byte load of the modulation row down to its
        // use and waits for it there — k_node_pre 0.116 -> 0.146 ms per launch at QM9 B = 1250 — while this form keeps a dozen
        // loads in flight)
    }
}

}  // namespace jd
