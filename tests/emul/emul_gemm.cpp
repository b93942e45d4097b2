// TEST INFRASTRUCTURE ONLY: plain-loop stand-in for jt::gemm (jodo_amd/csrc/train_gemm.hip) in the host emulation build
// (tests/emul/hip/hip_runtime.h explains the build).  Same signature, same semantics; accumulates in double.
#include <cstdlib>
#include "train_gemm.h"
thread_local emu_idx threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
namespace jt {
// Mirrors the ROUNDING STRUCTURE of the device kernel so that the CPU suite predicts its accuracy: fp32 fused multiply-adds, four
// accumulators taken in turn by the k-pairs, split-K partial sums over 512-wide chunks (rounded up to the 32-wide tile) added in order.
void gemm(hipStream_t, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
          const float* bias, int acc, float* ws, size_t ws_floats) {
    // debugging aid: products summed in double; value = bit mask of the products it applies to (1 forward, 2 input gradient, 4 weight gradient)
    static const int exact_mask = getenv("JODO_EMUL_GEMM_DOUBLE") ? atoi(getenv("JODO_EMUL_GEMM_DOUBLE")) : 0;
    const int kind = tA ? 4 : (tB ? 1 : 2);
    static const int only_n = getenv("JODO_EMUL_ONLY_N") ? atoi(getenv("JODO_EMUL_ONLY_N")) : 0, only_k = getenv("JODO_EMUL_ONLY_K") ? atoi(getenv("JODO_EMUL_ONLY_K")) : 0;
    if ((exact_mask & kind) && (!only_n || N == only_n) && (!only_k || K == only_k)) {
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double t = 0.0;
                for (int k = 0; k < K; ++k) t += (double)(tA ? A[(long)k * lda + m] : A[(long)m * lda + k]) * (double)(tB ? B[(long)n * ldb + k] : B[(long)k * ldb + n]);
                if (bias) t += bias[n];
                float* o = C + (long)m * ldc + n;
                *o = acc ? (float)(*o + t) : (float)t;
            }
        return;
    }
    const int gx = (N + 63) / 64, gy = (M + 63) / 64;
    int nsplit = 1;
    if (ws && K >= 512 && (tA || (K >= 2048 && (long)gx * gy < 512))) {
        nsplit = (K + 511) / 512;
        const long cap = (long)(ws_floats / ((size_t)M * N));
        if (nsplit > cap) nsplit = (int)cap;
        if (nsplit > 256) nsplit = 256;
        if (nsplit < 1) nsplit = 1;
    }
    int kchunk = (K + nsplit - 1) / nsplit;
    kchunk = (kchunk + 31) / 32 * 32;                   // the device kernel's K tile
    if (kchunk < 32) kchunk = 32;
    nsplit = K > 0 ? (K + kchunk - 1) / kchunk : 1;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float total = 0.f;
            for (int z = 0; z < nsplit; ++z) {
                static const int mode = getenv("JODO_EMUL_MODE") ? atoi(getenv("JODO_EMUL_MODE")) : 0;
                float c4[4] = {0.f, 0.f, 0.f, 0.f};
                const int k1 = K < (z + 1) * kchunk ? K : (z + 1) * kchunk;
                if (mode == 0) {
                    for (int k = z * kchunk; k < k1; ++k) {
                        const float a = tA ? A[(long)k * lda + m] : A[(long)m * lda + k];
                        const float b = tB ? B[(long)n * ldb + k] : B[(long)k * ldb + n];
                        float& c = c4[((k - z * kchunk) >> 1) & 3];
                        c = fmaf(a, b, c);
                    }
                    total += (c4[0] + c4[1]) + (c4[2] + c4[3]);
                } else {
                    float run[4] = {0.f, 0.f, 0.f, 0.f};
                    int tile = 0;
                    for (int k0 = z * kchunk; k0 < k1; k0 += 16, ++tile) {
                        float t4[4] = {0.f, 0.f, 0.f, 0.f};
                        for (int k = k0; k < k0 + 16 && k < k1; ++k) {
                            const float a = tA ? A[(long)k * lda + m] : A[(long)m * lda + k];
                            const float b = tB ? B[(long)n * ldb + k] : B[(long)k * ldb + n];
                            float& c = t4[((k - k0) >> 1) & 3];
                            c = fmaf(a, b, c);
                        }
                        run[tile & 3] += (t4[0] + t4[1]) + (t4[2] + t4[3]);
                    }
                    total += (run[0] + run[1]) + (run[2] + run[3]);
                }
            }
            if (bias) total += bias[n];
            float* o = C + (long)m * ldc + n;
            *o = acc ? *o + total : total;
        }
}
}
