// Gate experiments for the split-bf16 (three-term) MFMA form of the projections (round-5 review, "Next round" item 1, step A).
// Nothing here is on the product path: measurement helpers behind jodo_debug_* entry points (tools/split_gate.py,
// tests/test_split_gate.py).
//   (i)   k_mfma_bf16_valu: does vector work issued between dependent v_mfma_f32_32x32x16_bf16 hide under them?
//   (ii)  k_chain<256, MODE, 1>: one K = 256 -> 256 projection in the strip model, exact-fp32 MFMA chain vs the split form; the error
//         against float64 is taken by the caller.
//   (iii) k_chain<128, MODE, TILES> with iters > 1: the K = 128 -> 256 shape of the pair update's largest projection, weights streamed
//         from L2 through the ring exactly as in production, timed with HIP events.
#include "dgt_split.h"
#include "../../include/jodo_hip.h"
#include "jodo_hip_internal.h"

using namespace jd;

template <int NV, int CH>
__global__ __launch_bounds__(64, 1) void k_mfma_bf16_valu(int iters, float* __restrict__ sink) {
    f32x16 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = zero16();
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.0f + 0.01f * (threadIdx.x + j)); b[j] = (__bf16)0.5f; }
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            acc[r % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[r % CH], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) v[(r * NV + k) & 15] = fmaf(v[(r * NV + k) & 15], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c][0];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 123.456f) sink[0] = s;
}

template <int NV, int CH>
static void launch_bf16_valu(int waves, int iters, float* sink) {
    hipLaunchKernelGGL((k_mfma_bf16_valu<NV, CH>), dim3(waves), dim3(64), 0, 0, iters, sink);
}

extern "C" int jodo_debug_mfma_bf16_valu(int iters, int nv, int chains, int waves_per_simd, float* sink_dev, float* tflops_out) {
    if (iters <= 0 || !sink_dev || !tflops_out) return jodo_set_error(JODO_ERR_ARG, "mfma_bf16_valu: bad argument");
    if (waves_per_simd < 1 || waves_per_simd > 8) return jodo_set_error(JODO_ERR_ARG, "mfma_bf16_valu: waves_per_simd");
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return jodo_set_error(JODO_ERR_LAUNCH, "mfma_bf16_valu: events");
    const int waves = 1024 * waves_per_simd;
    bool ok = true;
    for (int rep = 0; rep < 2 && ok; ++rep) {
        (void)hipEventRecord(e0, 0);
        const int key = nv * 10 + chains;
        switch (key) {
            case 1: launch_bf16_valu<0, 1>(waves, iters, sink_dev); break;
            case 21: launch_bf16_valu<2, 1>(waves, iters, sink_dev); break;
            case 41: launch_bf16_valu<4, 1>(waves, iters, sink_dev); break;
            case 61: launch_bf16_valu<6, 1>(waves, iters, sink_dev); break;
            case 81: launch_bf16_valu<8, 1>(waves, iters, sink_dev); break;
            case 161: launch_bf16_valu<16, 1>(waves, iters, sink_dev); break;
            case 2: launch_bf16_valu<0, 2>(waves, iters, sink_dev); break;
            case 42: launch_bf16_valu<4, 2>(waves, iters, sink_dev); break;
            case 82: launch_bf16_valu<8, 2>(waves, iters, sink_dev); break;
            case 162: launch_bf16_valu<16, 2>(waves, iters, sink_dev); break;
            default: ok = false;
        }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (!ok) return jodo_set_error(JODO_ERR_ARG, "mfma_bf16_valu: unsupported (nv, chains)");
    const int rc = jodo_check_launch("k_mfma_bf16_valu");
    if (rc != JODO_OK) return rc;
    *tflops_out = (float)((double)waves * iters * 8.0 * 32768.0 / (ms * 1e-3) / 1e12);
    return JODO_OK;
}

// MODE 0: exact fp32 (v_mfma_f32_32x32x2_f32, the product form); 1: split, one accumulator; 2: split, the five correction products
// in an accumulator of their own (added once per output block).
// x [rows, K] row-major; wf: the f32 packed projection (dgt_pack.cpp put_proj, natural maps); wsp: the split packing of the same
// matrix; y [rows, 256].  iters == 1: y = W x.  iters > 1 (timing): the projection is repeated on x + 1e-3 * (running block sum),
// nothing but a 16-value digest per item is kept, y receives the digest.
template <int K, int MODE, int TILES>
__global__ __launch_bounds__(64, 1) void k_chain(const float* __restrict__ x, int rows, const float* __restrict__ wf,
                                                 const float* __restrict__ wsp, float* __restrict__ y, int iters) {
    constexpr int NB = 8;                       // 256 output features
    constexpr int KQ = K / 8;                   // f32 quads per block
    constexpr int NS = K / 16;                  // K16 steps per block
    constexpr int PG = 4;
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    float xr[TILES][K / 2];
    int row[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        row[t] = (blockIdx.x * TILES + t) * 32 + j;
        const int rc = row[t] < rows ? row[t] : rows - 1;
        load_nat<K / 32>(x + (size_t)rc * K, half, xr[t]);
    }
    float dig[TILES][16];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int s = 0; s < 16; ++s) dig[t][s] = 0.f;
    if constexpr (MODE == 0) {
        const WSrc ws = make_wsrc(wf, lane);
        WPipe<PG> wp;
        wpipe_prime(wp, ws, 0);
        for (int it = 0; it < iters; ++it) {
#pragma unroll 1
            for (int b = 0; b < NB; ++b) {
                const unsigned cur = (unsigned)b * KQ * 1024, nxt = b + 1 < NB ? cur + KQ * 1024 : 0;
                // (TILES > 1 would re-stream the block per tile in the f32 form: the ring holds one group)
                f32x16 acc = mfma_block_p<KQ, PG>(wp, ws, cur, nxt, xr[0], zero16());
                if (iters == 1) {
                    float r[16];
#pragma unroll
                    for (int s = 0; s < 16; ++s) r[s] = acc[s];
                    if (row[0] < rows) store16(y + (size_t)row[0] * 256 + b * 32 + half * 16, r);
                } else {
#pragma unroll
                    for (int s = 0; s < 16; ++s) dig[0][s] += acc[s];
                }
            }
            if (iters > 1) {
#pragma unroll
                for (int s = 0; s < K / 2; ++s) xr[0][s] = fmaf(dig[0][s & 15], 1e-3f, xr[0][s]);
            }
        }
    } else {
        const WSrc ws = make_wsrc(wsp, lane);
        WPipeS<PG> wp;
        wpipe_prime_s(wp, ws, 0);
        for (int it = 0; it < iters; ++it) {
            Split8 xs[TILES][NS];
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int g = 0; g < NS; ++g) xs[t][g] = split8(&xr[t][8 * g]);
#pragma unroll 1
            for (int b = 0; b < NB; ++b) {
                const unsigned cur = (unsigned)b * NS * 3072, nxt = b + 1 < NB ? cur + NS * 3072 : 0;
                f32x16 acc[TILES];
#pragma unroll
                for (int t = 0; t < TILES; ++t) acc[t] = zero16();
                if constexpr (MODE == 1) {
                    mfma_block_s<NS, PG, TILES>(wp, ws, cur, ws, nxt, xs, acc);
                } else {
                    // main chain hi * hi, corrections in their own accumulator (two independent chains per tile)
                    f32x16 cor[TILES];
#pragma unroll
                    for (int t = 0; t < TILES; ++t) cor[t] = zero16();
#pragma unroll
                    for (int g = 0; g < NS / PG; ++g) {
                        u32x4 c[PG][3];
#pragma unroll
                        for (int i = 0; i < PG; ++i)
#pragma unroll
                            for (int t = 0; t < 3; ++t) c[i][t] = wp.q[i][t];
#pragma unroll
                        for (int i = 0; i < PG; ++i)
#pragma unroll
                            for (int t = 0; t < 3; ++t)
                                wp.q[i][t] = g + 1 < NS / PG ? wload_s(ws, cur, (g + 1) * PG + i, t) : wload_s(ws, nxt, i, t);
                        pipeline_fence();
#pragma unroll
                        for (int i = 0; i < PG; ++i) {
                            const bf16x8 wh = as_bf16x8(c[i][0]), wm = as_bf16x8(c[i][1]), wl = as_bf16x8(c[i][2]);
#pragma unroll
                            for (int t = 0; t < TILES; ++t) {
                                const Split8& a = xs[t][g * PG + i];
                                cor[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, a.l, cor[t], 0, 0, 0);
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, a.h, acc[t], 0, 0, 0);
                                cor[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, a.h, cor[t], 0, 0, 0);
                                cor[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, a.m, cor[t], 0, 0, 0);
                                cor[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, a.m, cor[t], 0, 0, 0);
                                cor[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, a.h, cor[t], 0, 0, 0);
                            }
                        }
                        pipeline_fence();
                    }
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
#pragma unroll
                        for (int s = 0; s < 16; ++s) acc[t][s] += cor[t][s];
                }
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    if (iters == 1) {
                        float r[16];
#pragma unroll
                        for (int s = 0; s < 16; ++s) r[s] = acc[t][s];
                        if (row[t] < rows) store16(y + (size_t)row[t] * 256 + b * 32 + half * 16, r);
                    } else {
#pragma unroll
                        for (int s = 0; s < 16; ++s) dig[t][s] += acc[t][s];
                    }
                }
            }
            if (iters > 1) {
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int s = 0; s < K / 2; ++s) xr[t][s] = fmaf(dig[t][s & 15], 1e-3f, xr[t][s]);
            }
        }
    }
    if (iters > 1) {
#pragma unroll
        for (int t = 0; t < TILES; ++t)
            if (row[t] < rows) store16(y + (size_t)row[t] * 256 + half * 16, dig[t]);
    }
}

// mode 0 / 1 / 2 as above; K in {128, 256}; tiles in {1, 2} (2: split modes only).  ms_out != NULL: the launch is repeated and the
// second repetition timed with HIP events on `stream` (synchronises).  wf / wsp: device copies of jodo_debug_pack_split's outputs.
extern "C" int jodo_debug_chain(int mode, int K, int tiles, const float* x, int rows, const float* wf, const void* wsp, float* y,
                                int iters, float* ms_out, void* stream) {
    if (rows <= 0 || iters <= 0 || !x || !y || (mode == 0 ? !wf : !wsp)) return jodo_set_error(JODO_ERR_ARG, "debug_chain: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((rows + 32 * tiles - 1) / (32 * tiles)), blk(64);
    const float* ws = (const float*)wsp;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ms_out && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) return jodo_set_error(JODO_ERR_LAUNCH, "debug_chain: events");
    bool ok = true;
    for (int rep = 0; rep < (ms_out ? 2 : 1) && ok; ++rep) {
        if (ms_out) (void)hipEventRecord(e0, st);
        const int key = K * 100 + mode * 10 + tiles;
        switch (key) {
            case 25601: hipLaunchKernelGGL((k_chain<256, 0, 1>), grid, blk, 0, st, x, rows, wf, ws, y, iters); break;
            case 25611: hipLaunchKernelGGL((k_chain<256, 1, 1>), grid, blk, 0, st, x, rows, wf, ws, y, iters); break;
            case 25621: hipLaunchKernelGGL((k_chain<256, 2, 1>), grid, blk, 0, st, x, rows, wf, ws, y, iters); break;
            case 12801: hipLaunchKernelGGL((k_chain<128, 0, 1>), grid, blk, 0, st, x, rows, wf, ws, y, iters); break;
            case 12811: hipLaunchKernelGGL((k_chain<128, 1, 1>), grid, blk, 0, st, x, rows, wf, ws, y, iters); break;
            case 12812: hipLaunchKernelGGL((k_chain<128, 1, 2>), grid, blk, 0, st, x, rows, wf, ws, y, iters); break;
            case 12821: hipLaunchKernelGGL((k_chain<128, 2, 1>), grid, blk, 0, st, x, rows, wf, ws, y, iters); break;
            case 12822: hipLaunchKernelGGL((k_chain<128, 2, 2>), grid, blk, 0, st, x, rows, wf, ws, y, iters); break;
            default: ok = false;
        }
        if (ms_out) { (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1); }
    }
    if (ms_out) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *ms_out = ms;
    }
    if (!ok) return jodo_set_error(JODO_ERR_ARG, "debug_chain: unsupported (K, mode, tiles) = (%d, %d, %d)", K, mode, tiles);
    return jodo_check_launch("k_chain");
}
