// Strip form of the training path's row products:  C[M, N] (+)= A[M, K] op(B)[K, N] (+ bias) with M = every edge
// (or node) row of the batch and N, K <= a few hundred — the forward X W^T and the input gradient dY W of every projection.
//
// The LDS-tiled kernel of train_gemm.hip reaches 0.45 of the fp32 MFMA peak on these shapes (matrix pipe busy 35 - 45 % of the time,
// waves waiting on barriers and tile loads the rest, whatever the tile depth and occupancy: tools/gpu_gemm_pmc.sh).  This is the
// inference path's formulation instead (dgt_device.h, dgt_kernels_pre.h k_rowgemm): one wave owns 32 rows (row in lane & 31, the two
// half-waves hold the two k-slots of v_mfma_f32_32x32x2_f32), the weights are the A operand and stream from L2 through the software
// ring of dgt_device.h in a packed [block][k / 8][lane][4] image; no barriers (a workgroup is four independent waves, LDS only
// transposes each wave's own activation chunk).  Training weights change every step, so k_pack_w builds the packed image of the (slice of the) weight first, into the
// caller's scratch — a launch of a few microseconds in front of products that take tens.
//
// Rounding: one fp32 accumulator per output walks k in steps of two (k, k + 16 of every 32-feature block pair up), against two
// interleaved chains in the tiled kernel; both are within the float64 bounds of tests/test_train_gpu.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include "../../jodo_amd/csrc/dgt_device.h"

namespace jt {

using jd::f32x16;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// packed[(ob * (Kp / 8) + kq) * 256 + lane * 4 + e] = W(n, k):  n = ob 32 + 16 ((i >> 2) & 1) + (i & 3) + 4 (i >> 3), i = lane & 31 (the
// accumulator register image of dgt_pack.cpp's natural out map), k = 32 (kq / 4) + 16 (lane >> 5) + 4 (kq % 4) + e (natural in map);
// W(n, k) = tB ? B[n ldb + k] : B[k ldb + n]; zero outside N x K.
__global__ void k_pack_w(const float* __restrict__ B, int ldb, int tB, int N, int K, int NB, int Kp, float* __restrict__ out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)NB * (Kp / 8) * 64;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const long blk = t >> 6;
    const int kq = (int)(blk % (Kp / 8)), ob = (int)(blk / (Kp / 8));
    const int i = lane & 31, kh = lane >> 5;
    const int n = ob * 32 + 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3);
    const int k0 = 32 * (kq >> 2) + 16 * kh + 4 * (kq & 3);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = k0 + e;
        v[e] = (n < N && k < K) ? (tB ? B[(long)n * ldb + k] : B[(long)k * ldb + n]) : 0.f;
    }
    reinterpret_cast<float4*>(out)[t] = make_float4(v[0], v[1], v[2], v[3]);
}

template <int KQ, int PG>
__device__ __forceinline__ f32x16 block_x(jd::WPipe<PG>& p, const jd::WSrc& w, unsigned cur_off, unsigned next_off, const float (&act)[KQ * 4], f32x16 acc, int no_w, int no_mfma) {
#pragma unroll
    for (int g = 0; g < KQ / PG; ++g) {
        float4 cur[PG];
#pragma unroll
        for (int i = 0; i < PG; ++i) cur[i] = p.q[i];
        if (!no_w) {
            if (g + 1 < KQ / PG) {
#pragma unroll
                for (int i = 0; i < PG; ++i) p.q[i] = jd::wload(w, cur_off, (g + 1) * PG + i);
            } else {
#pragma unroll
                for (int i = 0; i < PG; ++i) p.q[i] = jd::wload(w, next_off, i);
            }
        }
        jd::pipeline_fence();
        if (!no_mfma) {
#pragma unroll
            for (int i = 0; i < PG; ++i) {
                const int k = (g * PG + i) * 4;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].x, act[k + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].y, act[k + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].z, act[k + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[i].w, act[k + 3], acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PG; ++i) acc[i] += cur[i].x * act[i];
        }
        jd::pipeline_fence();
    }
    return acc;
}

struct StripArgs {
    const float* X; long ldx;
    float* Y; long ldy;
    const float* Wp; const float* bias;
    int rows, K, N, NB, nob, acc, groups, items;
    int no_w, no_store, no_mfma, copies; long copy_floats;
};

// A workgroup is four independent waves (no barriers; one dispatch instead of four).  The activation chunk of a strip (32 rows x 64
// features) is read with coalesced 16-byte loads — four rows of 256 bytes per instruction — and turned into the row-per-lane register
// image through the wave's own 8.5 KiB of LDS: reading each lane's row straight from global memory (the layout the MFMA wants) makes
// every load instruction touch 32 cache lines and the strip's 32 KiB working set fight the other waves for the 32 KiB L1
// (first version: 76 % of wave time waiting, matrix pipe 52 % busy).
#define STRIP_LD 68
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_strip_gemm(StripArgs G) {
    __shared__ __attribute__((aligned(16))) float S[4][32][STRIP_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, half = lane >> 5;
    const int stride = gridDim.x * 4;
    int id = blockIdx.x * 4 + wave;                                   // item = (strip, output group); waves walk the items persistently:
    if (id >= G.items) return;                                        // launch, ring priming and the first activation load are paid once
    const int kq = G.K / 8, nch = G.K / 64;
    const jd::WSrc ws = jd::make_wsrc(G.Wp + (size_t)__builtin_amdgcn_readfirstlane(id % G.copies) * G.copy_floats, lane);
    auto woff = [&](int ob, int c) { return (unsigned)(((size_t)ob * kq + (size_t)c * 8) * 1024); };
    // quad i of a chunk: lane l holds floats 4 (l & 15) .. + 3 of row 4 i + (l >> 4)
    auto src = [&](int item, int i) {
        int r = (item / G.groups) * 32 + 4 * i + (lane >> 4);
        r = r < G.rows ? r : G.rows - 1;
        return G.X + (size_t)r * G.ldx + 4 * (lane & 15);
    };
    jd::WPipe<8> wp;
    jd::wpipe_prime(wp, ws, woff((id % G.groups) * G.nob, 0));
    f32x4 g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = *reinterpret_cast<const f32x4*>(src(id, i));
    float (*Sw)[STRIP_LD] = S[wave];
    for (; id < G.items; id += stride) {
        const int r0 = (id / G.groups) * 32, row = r0 + j;
        const int ob0 = (id % G.groups) * G.nob;
        const int nv = G.NB - ob0 < G.nob ? G.NB - ob0 : G.nob;      // output blocks of this item (1..4)
        const int nid = id + stride;
        const int nob0 = nid < G.items ? (nid % G.groups) * G.nob : ob0;
        f32x16 acc[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] = jd::zero16();
        for (int c = 0; c < nch; ++c) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(&Sw[4 * i + (lane >> 4)][4 * (lane & 15)]) = g[i];
            if (c + 1 < nch) {                                       // next chunk (or the next item's first): in flight behind these MFMAs
#pragma unroll
                for (int i = 0; i < 8; ++i) g[i] = *reinterpret_cast<const f32x4*>(src(id, i) + (c + 1) * 64);
            } else if (nid < G.items) {
#pragma unroll
                for (int i = 0; i < 8; ++i) g[i] = *reinterpret_cast<const f32x4*>(src(nid, i));
            }
            float x[32];
#pragma unroll
            for (int bq = 0; bq < 8; ++bq) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&Sw[j][(bq >> 2) * 32 + half * 16 + (bq & 3) * 4]);
                x[bq * 4 + 0] = v[0]; x[bq * 4 + 1] = v[1]; x[bq * 4 + 2] = v[2]; x[bq * 4 + 3] = v[3];
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o < nv) {
                    const bool last_o = o + 1 >= nv;
                    const unsigned nxt = !last_o ? woff(ob0 + o + 1, c) : (c + 1 < nch ? woff(ob0, c + 1) : woff(nob0, 0));
                    acc[o] = block_x<8>(wp, ws, woff(ob0 + o, c), nxt, x, acc[o], G.no_w, G.no_mfma);
                }
            }
        }
        if (row >= G.rows) continue;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o < nv) {
                const int col0 = (ob0 + o) * 32 + half * 16;
                float r[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) r[s] = acc[o][s];
                if (G.bias) {
                    float b[16];
                    jd::load16(G.bias + col0, b);
#pragma unroll
                    for (int s = 0; s < 16; ++s) r[s] += b[s];
                }
                float* yp = G.Y + (size_t)row * G.ldy + col0;
                if (G.no_store) {
                    if (r[0] == 123.456f) yp[0] = r[1];
                } else {
                    if (G.acc) {
                        float old[16];
                        jd::load16(yp, old);
#pragma unroll
                        for (int s = 0; s < 16; ++s) r[s] = old[s] + r[s];
                    }
                    jd::store16(yp, r);
                }
            }
        }
    }
}


}  // namespace jt

extern "C" int x_strip(int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias, float* ws,
                       int nob, int no_w, int no_store, int no_mfma, int wgs_cap, int copies, void* stream) {
    using namespace jt;
    hipStream_t s = (hipStream_t)stream;
    const int NB = N / 32;
    const long total = (long)NB * (K / 8) * 64;
    for (int c = 0; c < copies; ++c) hipLaunchKernelGGL(k_pack_w, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, B, ldb, tB, N, K, NB, K, ws + (size_t)c * N * K);
    StripArgs G;
    G.X = A; G.ldx = lda; G.Y = C; G.ldy = ldc; G.Wp = ws; G.bias = bias;
    G.rows = M; G.K = K; G.N = N; G.NB = NB; G.acc = 0; G.no_w = no_w; G.no_store = no_store; G.no_mfma = no_mfma; G.copies = copies; G.copy_floats = (long)N * K;
    const int strips = (M + 31) / 32;
    G.nob = nob < NB ? nob : NB;
    G.groups = (NB + G.nob - 1) / G.nob;
    G.items = strips * G.groups;
    const int wgs = (G.items + 3) / 4;
    hipLaunchKernelGGL(k_strip_gemm, dim3(wgs < wgs_cap ? wgs : wgs_cap), dim3(256), 0, s, G);
    return (int)hipGetLastError();
}
