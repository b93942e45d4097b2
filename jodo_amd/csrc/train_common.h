// Shared by the training kernels (train_ops.h) and the training GEMM's epilogue (train_gemm.hip): dropout masks and SiLU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace jt {

// ---- dropout masks: Philox4x32-10 keyed by the call's seed, counter = (element / 4, site); the backward regenerates them ----
__host__ __device__ __forceinline__ void philox4(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4]) {
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
struct Drop { float p; unsigned long long seed; unsigned site; };
// multiplier of element idx: 0 (dropped) or 1 / (1 - p); p = 0 -> 1 (F.dropout / nn.Dropout semantics)
__host__ __device__ __forceinline__ float drop_mul(const Drop& d, unsigned long long idx) {
    if (d.p <= 0.f) return 1.f;
    unsigned u[4];
    philox4((unsigned)(idx >> 2), (unsigned)(idx >> 34), d.site, 0x4a4f444fu, (unsigned)d.seed, (unsigned)(d.seed >> 32), u);
    const float uni = (float)(u[idx & 3] >> 8) * (1.0f / 16777216.0f);
    return uni < d.p ? 0.f : 1.f / (1.f - d.p);
}

__host__ __device__ __forceinline__ float silu_f(float x) { return x / (1.f + expf(-x)); }
__host__ __device__ __forceinline__ float silu_grad(float x) {
    const float s = 1.f / (1.f + expf(-x));
    return s * (1.f + x * (1.f - s));
}

}  // namespace jt
