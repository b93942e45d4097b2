// Building-block self-test kernel: y = LN(W2 * silu(W1 * x + b1) + b2) over rows of x, using exactly
// the strip model / packed-weight chain the production kernels use.  Exposed through the C ABI as
// jodo_debug_mlp so tests can validate the MFMA lane maps and the packer on real hardware.
#include "dgt_device.h"
#include "../../include/jodo_hip.h"
#include "jodo_hip_internal.h"

using namespace jd;

__global__ __launch_bounds__(64) void k_debug_mlp(const float* __restrict__ x, int rows,
                                                  const float4* __restrict__ w1, const float* __restrict__ b1,
                                                  const float4* __restrict__ w2, const float* __restrict__ b2,
                                                  float* __restrict__ y) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int row = blockIdx.x * 32 + j;
    const int rowc = row < rows ? row : rows - 1;
    float a0[32];
    load_nat<2>(x + (size_t)rowc * 64, half, a0);
    float a1[64];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        f32x16 acc = mfma_block<8>(w1 + (size_t)b * 8 * 64 + lane, a0, zero16());
        float r[16];
        acc_bias(acc, b1 + b * 32 + half * 16, r);
#pragma unroll
        for (int s = 0; s < 16; ++s) a1[b * 16 + s] = silu_f(r[s]);
    }
    float a2[32];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        f32x16 acc = mfma_block<16>(w2 + (size_t)b * 16 * 64 + lane, a1, zero16());
        float r[16];
        acc_bias(acc, b2 + b * 32 + half * 16, r);
#pragma unroll
        for (int s = 0; s < 16; ++s) a2[b * 16 + s] = r[s];
    }
    layer_norm<32>(a2);
#pragma unroll
    for (int s = 0; s < 32; ++s) a2[s] = tanh_f(a2[s]);
    if (row < rows) store_nat<2>(y + (size_t)row * 64, half, a2);
}

extern "C" int jodo_debug_mlp(const float* x, int rows, const float* w1, const float* b1, const float* w2,
                              const float* b2, float* y, void* stream) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_debug_mlp, dim3((rows + 31) / 32), dim3(64), 0, (hipStream_t)stream, x, rows,
                       (const float4*)w1, b1, (const float4*)w2, b2, y);
    return jodo_check_launch("k_debug_mlp");
}

// fp32 matrix-pipe microbenchmark: every wave runs `iters` rounds of 8 independent v_mfma_f32_32x32x2_f32 chains
// (or one dependent chain, which is what a projection block is) on register operands — no memory traffic.  Gives
// the ceiling the roofline fractions are measured against on the actual box (spec: 157.3 TFLOP/s).
template <int CHAINS>
__global__ __launch_bounds__(64, 1) void k_mfma_peak(int iters, float* __restrict__ sink) {
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = zero16();
    float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8 / CHAINS; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0];
    if (s == 123.456f) sink[0] = s;             // keeps the chains alive
}

// 1024 x waves_per_simd waves; returns TFLOP/s of `chains` in {1, 8} (dependent chain / 8 independent chains).
// Synchronises: a measurement helper for bench.py / DESIGN.md, not part of the data path.
extern "C" int jodo_debug_mfma_peak(int iters, int chains, int waves_per_simd, float* sink_dev, float* tflops_out) {
    if (iters <= 0 || !sink_dev || !tflops_out || (chains != 1 && chains != 8)) return jodo_set_error(JODO_ERR_ARG, "mfma_peak: bad argument");
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return jodo_set_error(JODO_ERR_LAUNCH, "mfma_peak: events");
    if (waves_per_simd < 1 || waves_per_simd > 8) return jodo_set_error(JODO_ERR_ARG, "mfma_peak: waves_per_simd");
    const int waves = 1024 * waves_per_simd;
    for (int rep = 0; rep < 2; ++rep) {          // first repetition warms up clocks / code
        (void)hipEventRecord(e0, 0);
        if (chains == 1) hipLaunchKernelGGL(k_mfma_peak<1>, dim3(waves), dim3(64), 0, 0, iters, sink_dev);
        else hipLaunchKernelGGL(k_mfma_peak<8>, dim3(waves), dim3(64), 0, 0, iters, sink_dev);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
    }
    int rc = jodo_check_launch("k_mfma_peak");
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc != JODO_OK) return rc;
    const double flops = (double)waves * iters * 8.0 * 4096.0;
    *tflops_out = (float)(flops / (ms * 1e-3) / 1e12);
    return JODO_OK;
}

// Does VALU work issued between the MFMAs of a dependent chain hide under them?  NV independent v_fma per MFMA
// (and optionally one transcendental) in the same wave; returns the MFMA-only TFLOP/s.
template <int NV, int NT>
__global__ __launch_bounds__(64, 1) void k_mfma_valu(int iters, float* __restrict__ sink) {
    f32x16 acc = zero16();
    float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) v[(r * NV + k) & 15] = fmaf(v[(r * NV + k) & 15], 1.0001f, 0.5f);
#pragma unroll
            for (int k = 0; k < NT; ++k) v[(r + k) & 15] = __builtin_amdgcn_exp2f(v[(r + k) & 15] * 0.001f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = acc[0];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 123.456f) sink[0] = s;
}

extern "C" int jodo_debug_mfma_valu(int iters, int nv, int nt, int waves_per_simd, float* sink_dev, float* tflops_out) {
    if (iters <= 0 || !sink_dev || !tflops_out) return jodo_set_error(JODO_ERR_ARG, "mfma_valu: bad argument");
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return jodo_set_error(JODO_ERR_LAUNCH, "mfma_valu: events");
    if (waves_per_simd < 1 || waves_per_simd > 8) return jodo_set_error(JODO_ERR_ARG, "mfma_valu: waves_per_simd");
    const int waves = 1024 * waves_per_simd;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        if (nv == 0 && nt == 0) hipLaunchKernelGGL((k_mfma_valu<0, 0>), dim3(waves), dim3(64), 0, 0, iters, sink_dev);
        else if (nv == 4 && nt == 0) hipLaunchKernelGGL((k_mfma_valu<4, 0>), dim3(waves), dim3(64), 0, 0, iters, sink_dev);
        else if (nv == 8 && nt == 0) hipLaunchKernelGGL((k_mfma_valu<8, 0>), dim3(waves), dim3(64), 0, 0, iters, sink_dev);
        else if (nv == 12 && nt == 0) hipLaunchKernelGGL((k_mfma_valu<12, 0>), dim3(waves), dim3(64), 0, 0, iters, sink_dev);
        else if (nv == 16 && nt == 0) hipLaunchKernelGGL((k_mfma_valu<16, 0>), dim3(waves), dim3(64), 0, 0, iters, sink_dev);
        else if (nv == 4 && nt == 1) hipLaunchKernelGGL((k_mfma_valu<4, 1>), dim3(waves), dim3(64), 0, 0, iters, sink_dev);
        else if (nv == 4 && nt == 2) hipLaunchKernelGGL((k_mfma_valu<4, 2>), dim3(waves), dim3(64), 0, 0, iters, sink_dev);
        else { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return jodo_set_error(JODO_ERR_ARG, "mfma_valu: unsupported (nv, nt)"); }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
    }
    int rc = jodo_check_launch("k_mfma_valu");
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc != JODO_OK) return rc;
    *tflops_out = (float)((double)waves * iters * 8.0 * 4096.0 / (ms * 1e-3) / 1e12);
    return JODO_OK;
}
