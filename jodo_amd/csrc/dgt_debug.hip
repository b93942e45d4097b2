// Building-block self-test kernel: y = LN(W2 * silu(W1 * x + b1) + b2) over rows of x, using exactly
// the strip model / packed-weight chain the production kernels use.  Exposed through the C ABI as
// jodo_debug_mlp so tests can validate the MFMA lane maps and the packer on real hardware.
#include "dgt_device.h"
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
