// Device building blocks for the DGT kernels (gfx950 / CDNA4 only).
//
// Execution model used by every kernel in this directory ("strip" model):
//   * one wavefront (64 lanes) owns 32 items (edges or nodes); item j = lane & 31;
//   * a feature vector of an item is split over the two half-lanes h = lane >> 5 of that item:
//     register R of half h holds feature (R / 16) * 32 + h * 16 + (R % 16)  ("natural-half" slots);
//   * dense projections run on v_mfma_f32_32x32x2_f32 in the transposed orientation
//       D[out_feature, item] += W[out_feature, k] * X[k, item]
//     with the pre-packed weights as A operand (one 16-byte load per lane feeds 4 MFMAs, see
//     jodo_amd/packing.py) and the activation registers as B operand.  The accumulator of an output
//     block *is* the next projection's B operand: chains of projections stay in registers.
//   * per-item reductions over features (LayerNorm, head scores) are in-lane sums plus one
//     exchange with lane ^ 32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

__device__ __forceinline__ float tanh_f(float x) {
    const float t = fast_exp(-2.f * fabsf(x));            // in (0, 1]
    const float r = (1.f - t) * fast_rcp(1.f + t);
    return copysignf(r, x);
}
__device__ __forceinline__ float silu_f(float x) { return x * fast_rcp(1.f + fast_exp(-x)); }

// sum of a per-half partial over the two half-lanes of an item
__device__ __forceinline__ float pair_sum(float v) { return v + __shfl_xor(v, 32); }

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

// accumulator (+ bias in slot order) -> registers [16]
__device__ __forceinline__ void acc_bias(const f32x16& acc, const float* __restrict__ bias16, float (&r)[16]) {
    float b[16];
    load16(bias16, b);
#pragma unroll
    for (int s = 0; s < 16; ++s) r[s] = acc[s] + b[s];
}

// ---- LayerNorm (no affine, eps 1e-6, biased variance) over NR*2 features of an item ----------------
template <int NR>
__device__ __forceinline__ void layer_norm(float (&x)[NR]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) s += x[i];
    const float mean = pair_sum(s) * (1.f / (2 * NR));
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        x[i] -= mean;
        v = fmaf(x[i], x[i], v);
    }
    const float rstd = __builtin_amdgcn_rsqf(pair_sum(v) * (1.f / (2 * NR)) + 1e-6f);
#pragma unroll
    for (int i = 0; i < NR; ++i) x[i] *= rstd;
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
    }
}

// ---- Gaussian basis of a squared distance (CondGaussianLayer) ---------------------------------------
// tab: [3][64] = mu, 1/sigma, 1/(sqrt(2*3.14159)*sigma); entry 0 of each row unused (feature 0 = x').
// Half h produces features b*32 + h*16 + s, b = 0,1.
__device__ __forceinline__ void gbf64(float d2, float scale, float shift, const float* __restrict__ tab,
                                      int half, float (&g)[32]) {
    const float x = fmaf(d2, scale + 1.f, shift);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        float mu[16], is[16], cf[16];
        load16(tab + b * 32 + half * 16, mu);
        load16(tab + 64 + b * 32 + half * 16, is);
        load16(tab + 128 + b * 32 + half * 16, cf);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float z = (x - mu[s]) * is[s];
            g[b * 16 + s] = fast_exp(-0.5f * z * z) * cf[s];
        }
    }
    if (half == 0) g[0] = x;
}

}  // namespace jd
