// TEST INFRASTRUCTURE ONLY: plain-loop stand-in for jt::gemm (jodo_amd/csrc/train_gemm.hip) in the host emulation build
// (tests/emul/hip/hip_runtime.h explains the build).  Same signature, same semantics; accumulates in double.
#include <cstdlib>
#include "train_gemm.h"
thread_local emu_idx threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
namespace jt {
// Mirrors the ROUNDING STRUCTURE of the device kernel so that the CPU suite predicts its accuracy: fp32 fused multiply-adds, two
// accumulator chains taken in turn by the k-pairs, the split-K slices of train_gemm.h gemm_plan() added the way k_splitk_sum adds them:
// eight interleaved partial sums (slices z = g mod 8, ascending), then those in the order g = 0 .. 7.
void gemm(hipStream_t, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
          const float* bias, int acc, float* ws, size_t ws_floats, const GemmEpi* epi) {
    // debugging aid: products summed in double; value = bit mask of the products it applies to (1 forward, 2 input gradient, 4 weight gradient)
    static const int exact_mask = getenv("JODO_EMUL_GEMM_DOUBLE") ? atoi(getenv("JODO_EMUL_GEMM_DOUBLE")) : 0;
    const int kind = tA ? 4 : (tB ? 1 : 2);
    static const int only_n = getenv("JODO_EMUL_ONLY_N") ? atoi(getenv("JODO_EMUL_ONLY_N")) : 0, only_k = getenv("JODO_EMUL_ONLY_K") ? atoi(getenv("JODO_EMUL_ONLY_K")) : 0;
    if ((exact_mask & kind) && (!only_n || N == only_n) && (!only_k || K == only_k)) {
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double t = 0.0;
                for (int k = 0; k < K; ++k) t += (double)(tA ? A[(long)k * lda + m] : A[(long)m * lda + k]) * (double)(tB ? B[(long)n * ldb + k] : B[(long)k * ldb + n]);
                if (epi && epi->dbias && tA && n == 0) { double bsum = 0.0; for (int k = 0; k < K; ++k) bsum += (double)A[(long)k * lda + m]; epi->dbias[m] += (float)bsum; }
                if (bias) t += bias[n];
                if (epi && epi->act) { gemm_epilogue(*epi, (float)t, C, (long)m * ldc + n, (long)m * N + n); continue; }
                float* o = C + (long)m * ldc + n;
                *o = acc ? (float)(*o + t) : (float)t;
            }
        return;
    }
    const GemmPlan pl = gemm_plan(tA, M, N, K, ws != nullptr, ws_floats);
    const int nsplit = pl.nsplit, kchunk = pl.kchunk;
    if (epi && epi->dbias && tA) {                              // bias gradient: per slice a double sum, slices added in order
        for (int m = 0; m < M; ++m) {
            float sg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bs = 0.f;
            for (int z = 0; z < nsplit; ++z) {
                double t = 0.0;
                const int k1 = K < (z + 1) * kchunk ? K : (z + 1) * kchunk;
                for (int k = z * kchunk; k < k1; ++k) t += (double)A[(long)k * lda + m];
                if (nsplit > 1) sg[z & 7] += (float)t; else bs = (float)t;
            }
            if (nsplit > 1) { bs = sg[0]; for (int g = 1; g < 8; ++g) bs += sg[g]; }
            epi->dbias[m] += bs;
        }
    }
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float sg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, total = 0.f;
            for (int z = 0; z < nsplit; ++z) {
                float c2[2] = {0.f, 0.f};                     // the device kernel's two chains: even / odd k-pairs of the slice
                const int k1 = K < (z + 1) * kchunk ? K : (z + 1) * kchunk;
                for (int k = z * kchunk; k < k1; ++k) {
                    const float a = tA ? A[(long)k * lda + m] : A[(long)m * lda + k];
                    const float b = tB ? B[(long)n * ldb + k] : B[(long)k * ldb + n];
                    float& c = c2[((k - z * kchunk) >> 1) & 1];
                    c = fmaf(a, b, c);
                }
                if (nsplit > 1) sg[z & 7] += c2[0] + c2[1];
                else total = c2[0] + c2[1];
            }
            if (nsplit > 1) { total = sg[0]; for (int g = 1; g < 8; ++g) total += sg[g]; }
            if (bias) total += bias[n];
            if (epi && epi->act) { gemm_epilogue(*epi, total, C, (long)m * ldc + n, (long)m * N + n); continue; }
            float* o = C + (long)m * ldc + n;
            *o = acc ? *o + total : total;
        }
}
}

// The fused forward chains (jodo_amd/csrc/train_fused.hip) are matrix-instruction kernels: not available to the host emulation, which
// runs — and thereby checks — the op-by-op sequence they replace.
#include "train_fused.h"
namespace jt {
FusedPackLayout fused_pack_layout(const FusedDims&) { return FusedPackLayout{}; }
// the grouped weight-gradient launch: job by job (the device form runs the same plans in one launch, bit-identical to its own
// job-by-job form)
void gemm_dw_group(hipStream_t s, const GemmJob* jobs, int n, float* ws, size_t, size_t plan_floats) {
    for (int i = 0; i < n; ++i) {
        const GemmJob& q = jobs[i];
        GemmEpi e; e.act = 0; e.out2 = nullptr; e.drop.p = 0.f; e.drop.seed = 0; e.drop.site = 0; e.dbias = q.dbias;
        gemm(s, 1, 0, q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.C, q.ldc, nullptr, 1, ws, plan_floats, &e);
    }
}
bool fused_available(const FusedDims&) { return false; }
void fused_pack_block(hipStream_t, const FusedDims&, const FusedBlockParams&, float*) {}
void fused_chain_a(hipStream_t, const FusedDims&, const FusedTopo&, const FusedBlockParams&, const float*, const float*, const float*, const float*,
                   const float*, float*, float*, float*, float*, float*, float*, float*) {}
void fused_chain_b(hipStream_t, const FusedDims&, const FusedTopo&, const FusedBlockParams&, const float*, const float*, const float*, const float*, Drop,
                   Drop, float*, float*, float*, float*, float*, float*, float*, float*, int, int) {}
void fused_chain_c(hipStream_t, const FusedDims&, const FusedTopo&, const FusedBlockParams&, const float*, const float*, const float*, const float*,
                   const float*, const float*, float*, float*, float*, float*, float*, float*) {}
}
namespace jt {
void fused_pack_block_bwd(hipStream_t, const FusedDims&, const FusedBlockParams&, float*) {}
void fused_bwd_c(hipStream_t, const FusedDims&, const FusedTopo&, const FusedBlockParams&, const float*, const float*, float*, const float*, const float*,
                 const float*, const float*, float*, float*, float*, float*, float*) {}
void fused_bwd_b(hipStream_t, const FusedDims&, const FusedTopo&, const FusedBlockParams&, const float*, const float*, const float*, const float*, const float*,
                 const float*, const float*, Drop, Drop, float*, float*, float*, float*, float*) {}
void fused_bwd_a(hipStream_t, const FusedDims&, const FusedTopo&, const FusedBlockParams&, const float*, const float*, const float*, const float*, const float*,
                 const float*, float*, float*, float*, float*) {}
void fused_node_ln_mod(hipStream_t, long, int, const float*, const float*, const int*, const float*, int, int, int, int, float*, float*, float*) {}
void fused_node_ln_mod_bwd(hipStream_t, long, int, const float*, const float*, const float*, const int*, const float*, int, int, float*, int) {}
bool fused_attention_available(int, int, int) { return false; }
void fused_attn_fwd(hipStream_t, const AttnTopo&, int, int, int, int, float, const float*, const float*, const float*, const float*, const float*, const float*,
                    const float*, float*, float*) {}
void fused_attn_bwd(hipStream_t, const AttnTopo&, int, int, int, int, float, const float*, const float*, const float*, const float*, const float*, const float*,
                    const float*, float*, float*, float*, float*, float*, float*) {}
void fused_gbf_bwd(hipStream_t, long, int, const float*, const int*, const float*, const float*, const float*, const float*, int, int, float*, float*, int,
                   float*, float*) {}
}
