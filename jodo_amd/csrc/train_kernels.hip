// First slice of the training step: backward of phase D of a DGT block — edge residual with the message
// gate, LayerNorm2 + modulate, edge FFN with its gate (EquivariantMixBlock.forward, /root/reference/models/mol_gnn.py:313-317;
// the autograd of loss.backward(), /root/reference/losses.py:286-385, restricted to those five lines):
//
//     x1 = e_in + eg1 * ehat                  ehat = node2edge_lin(h_attn_i + h_attn_j)
//     xn = LN(x1) (1 + ec2) + es2
//     y  = W4 SiLU(W3 xn + b3) + b4
//     e_out = xn + eg2 * y
//
// Given d e_out it produces d e_in, d ehat, dW3, db3, dW4, db4 and the gradients of the four modulation vectors per
// modulation row (training draws one noise level per molecule, losses.py:313, so they are per-molecule sums).
//
// Design: RECOMPUTE instead of saving activations — a wave owns 32 edge rows (lane = row, features split over
// the two half-lanes exactly like the forward kernels), reruns the forward chain in registers on the forward's packed
// weights, then walks it backwards:
//   * dX = W^T dY uses the same transposed-orientation MFMA blocks as the forward, on packed copies of W4^T and W3^T
//     (jodo_edge_ffn_pack);
//   * dW = sum_rows dY (x) X is a GEMM whose contraction index is the ROW: both operands go through a per-wave LDS tile in
//     row-major [row][feature] layout — there lane (m, k-slot) of an MFMA A / B operand reads element [row 2t + k-slot][feature
//     m], a conflict-free column walk — and the 32 x 32 output tiles stay in accumulators across all tiles of the wave
//     (persistent waves, grid = one wave per SIMD);  bias gradients are column sums of the same tiles;
//   * the wave's partial dW / db go to the workspace and a second kernel adds the partials in a fixed order; the per-row
//     modulation contributions go to the workspace too and are summed per modulation row by a segmented-sum kernel: no
//     atomics anywhere, results are bit-deterministic.
// Registers: 128 accumulators for one weight gradient + ~250 of working state, so the weight gradients are produced by two
// kinds of workgroups in the same launch (blockIdx.y = 0: dX + dW4 / db4 + modulation rows, 1: dW3 / db3).
#include <hip/hip_runtime.h>
#include "../../include/jodo_hip.h"
#include "dgt_device.h"
#include "jodo_hip_internal.h"

namespace {
using namespace jd;

constexpr int DE = 64, RR = 2, HID = DE * RR;          // slice: nf = 256 (De = 64), mlp_ratio = 2 (the QM9 configs)
constexpr int NE = DE / 32, NH = HID / 32;             // 32-feature blocks of an edge row / of the hidden layer
constexpr int SA = DE + 4, SB = HID + 4;               // padded LDS row strides (floats): conflict-free b128 writes

struct BwdArgs {
    int rows, U, n_waves;
    const float *e_in, *ehat, *mods, *dout;
    const int* row_mod;
    const float* packed;                               // packed projections (natural maps): W3 | W4 | W4^T | W3^T at byte offsets o3, o4, o4t, o3t
    unsigned o3, o4, o4t, o3t;
    const float *b3, *b4;                              // plain biases
    float *d_e_in, *d_ehat;
    float* dmod_rows;                                  // [rows][4][De]: eg1, es2, ec2, eg2 contributions
    float* partial;                                    // [n_waves][PART] per-wave weight-gradient partials
};
constexpr int PART = HID * DE + HID + DE * HID + DE;   // dW3 | db3 | dW4 | db4

__device__ __forceinline__ float dsilu(float x) {      // d/dx x sigmoid(x) = s (1 + x (1 - s))
    const float s = fast_rcp(1.f + fast_exp(-x));
    return s * fmaf(x, 1.f - s, 1.f);
}

// registers (natural-half layout) -> LDS tile [32 rows][stride], row = lane & 31
template <int NB, int STRIDE>
__device__ __forceinline__ void tile_store(float* tile, int j, int half, const float (&r)[NB * 16]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float t[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) t[s] = r[b * 16 + s];
        store16(tile + j * STRIDE + b * 32 + half * 16, t);
    }
}

// acc[mi][nj] += sum_rows A[row][mi*32 + m] * B[row][nj*32 + n]   (A tile stride SAs, B tile stride SBs)
template <int MB, int NB, int SAs, int SBs>
__device__ __forceinline__ void outer_accumulate(const float* ta, const float* tb, int j, int half, f32x16 (&acc)[MB * NB]) {
#pragma unroll 4
    for (int t = 0; t < 16; ++t) {
        const int row = 2 * t + half;
        float a[MB], b[NB];
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) a[mi] = ta[row * SAs + mi * 32 + j];
#pragma unroll
        for (int nj = 0; nj < NB; ++nj) b[nj] = tb[row * SBs + nj * 32 + j];
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int nj = 0; nj < NB; ++nj) acc[mi * NB + nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[nj], acc[mi * NB + nj], 0, 0, 0);
    }
}

// column sums of a tile: lane l adds feature f = l + 64 k of all 32 rows
template <int NF, int STRIDE>
__device__ __forceinline__ void column_sums(const float* tile, int lane, float (&sum)[(NF + 63) / 64]) {
#pragma unroll
    for (int k = 0; k < (NF + 63) / 64; ++k) {
        const int f = lane + 64 * k;
        if (f < NF) {
            float s = 0.f;
            for (int r = 0; r < 32; ++r) s += tile[r * STRIDE + f];
            sum[k] += s;
        }
    }
}

// accumulator tile (mi, nj) -> row-major matrix [M][N]:  register r of lane l = element (mi*32 + (r&3) + 8 (r>>2) + 4 (l>>5), nj*32 + (l&31))
template <int MB, int NB>
__device__ __forceinline__ void write_tiles(float* dst, int ldn, int lane, const f32x16 (&acc)[MB * NB]) {
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int nj = 0; nj < NB; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dst[(size_t)(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ldn + nj * 32 + (lane & 31)] = acc[mi * NB + nj][r];
}

template <int WHICH>
__device__ __forceinline__ void bwd_body(const BwdArgs& A, float* tA, float* tB) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int wave = blockIdx.x;
    constexpr int KQ3 = DE / 8, KQ4 = HID / 8;                 // quads per output block for K = De / K = r De
    f32x16 acc[NE * NH];                                       // WHICH 0: dW4 [De x rDe] tiles (mi < NE, nj < NH); 1: dW3 [rDe x De] (mi < NH, nj < NE)
#pragma unroll
    for (int i = 0; i < NE * NH; ++i) acc[i] = zero16();
    // weights stream from L2 through the software-pipelined ring of the forward kernels (dgt_device.h): one group of 8 quads in
    // flight, fences keep hipcc from hoisting every block's loads to the top (it did: 128 registers of weights, 860 B of scratch)
    const WSrc ws = make_wsrc(A.packed, lane);
    WPipe<8> wp;
    wpipe_prime(wp, ws, A.o3);
    float dbsum[WHICH == 0 ? 1 : 2] = {0.f};
    if (WHICH == 1) dbsum[WHICH == 1 ? 1 : 0] = 0.f;
    const int tiles = (A.rows + 31) / 32;
    for (int tile = wave; tile < tiles; tile += A.n_waves) {
        const int row = tile * 32 + j;
        const bool valid = row < A.rows;
        const size_t rc = (size_t)(valid ? row : A.rows - 1);
        const float* mr = A.mods + (size_t)A.row_mod[rc] * 6 * DE;          // es1 ec1 eg1 es2 ec2 eg2 (edge_time_mlp chunks, mol_gnn.py:289-290)
        const float *eg1 = mr + 2 * DE, *es2 = mr + 3 * DE, *ec2 = mr + 4 * DE, *eg2 = mr + 5 * DE;
        // ---- forward chain, recomputed ----
        float xh[NE * 16];
        load_nat<NE>(A.e_in + rc * DE, half, xh);
#pragma unroll
        for (int b = 0; b < NE; ++b) {
            float g[16], eh[16];
            load16(eg1 + b * 32 + half * 16, g);
            load16(A.ehat + rc * DE + b * 32 + half * 16, eh);
#pragma unroll
            for (int s = 0; s < 16; ++s) xh[b * 16 + s] = fmaf(g[s], eh[s], xh[b * 16 + s]);
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NE * 16; ++i) sum += xh[i];
        const float mean = pair_sum(sum) * (1.f / DE);
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < NE * 16; ++i) { xh[i] -= mean; var = fmaf(xh[i], xh[i], var); }
        const float rstd = __builtin_amdgcn_rsqf(pair_sum(var) * (1.f / DE) + 1e-6f);
        float xn[NE * 16];
#pragma unroll
        for (int b = 0; b < NE; ++b) {
            float sc[16], sh[16];
            load16(ec2 + b * 32 + half * 16, sc);
            load16(es2 + b * 32 + half * 16, sh);
#pragma unroll
            for (int s = 0; s < 16; ++s) { xh[b * 16 + s] *= rstd; xn[b * 16 + s] = fmaf(xh[b * 16 + s], 1.f + sc[s], sh[s]); }     // xh = x^ now
        }
        // x^ of block b again (it is not kept across the FFN: the working set next to 128 accumulators must fit 256 arch VGPRs)
        auto xhat_block = [&](int b, float (&o)[16]) {
            float g[16], eh[16], e0[16];
            load16(eg1 + b * 32 + half * 16, g);
            load16(A.ehat + rc * DE + b * 32 + half * 16, eh);
            load16(A.e_in + rc * DE + b * 32 + half * 16, e0);
#pragma unroll
            for (int s = 0; s < 16; ++s) o[s] = (fmaf(g[s], eh[s], e0[s]) - mean) * rstd;
        };
        float ds[NH * 16], hid[NH * 16];
#pragma unroll
        for (int b = 0; b < NH; ++b) {
            float bb[16];
            load16(A.b3 + b * 32 + half * 16, bb);
            const unsigned cur = A.o3 + (unsigned)b * KQ3 * 1024;
            const f32x16 p = mfma_block_p<KQ3>(wp, ws, cur, b + 1 < NH ? cur + KQ3 * 1024 : (WHICH == 0 ? A.o4 : A.o4t), xn, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) { const float v = p[s] + bb[s]; hid[b * 16 + s] = silu_f(v); ds[b * 16 + s] = dsilu(v); }
        }
        float gy[NE * 16];
        {
            float dout[NE * 16];
            load_nat<NE>(A.dout + rc * DE, half, dout);
#pragma unroll
            for (int b = 0; b < NE; ++b) {
                float g2[16];
                load16(eg2 + b * 32 + half * 16, g2);
                if constexpr (WHICH == 0) {                                       // y = W4 hid + b4 is only needed for d eg2
                    float bb[16], dg[16];
                    load16(A.b4 + b * 32 + half * 16, bb);
                    const unsigned cur = A.o4 + (unsigned)b * KQ4 * 1024;
                    const f32x16 y = mfma_block_p<KQ4>(wp, ws, cur, b + 1 < NE ? cur + KQ4 * 1024 : A.o4t, hid, zero16());
#pragma unroll
                    for (int s = 0; s < 16; ++s) dg[s] = valid ? dout[b * 16 + s] * (y[s] + bb[s]) : 0.f;
                    if (valid) store16(A.dmod_rows + ((size_t)row * 4 + 3) * DE + b * 32 + half * 16, dg);
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) gy[b * 16 + s] = valid ? g2[s] * dout[b * 16 + s] : 0.f;     // padding rows contribute nothing below
            }
        }
        if constexpr (WHICH == 0) {                                               // hid leaves the registers here (dW4's B operand tile)
            tile_store<NH, SB>(tB, j, half, hid);
            tile_store<NE, SA>(tA, j, half, gy);
        }
        // ---- backward through the FFN: d hid = W4^T gy, d pre3 = d hid * SiLU'(pre3) ----
#pragma unroll
        for (int b = 0; b < NH; ++b) {
            const unsigned cur = A.o4t + (unsigned)b * KQ3 * 1024;
            const f32x16 dh = mfma_block_p<KQ3>(wp, ws, cur, b + 1 < NH ? cur + KQ3 * 1024 : (WHICH == 0 ? A.o3t : A.o3), gy, zero16());
#pragma unroll
            for (int s = 0; s < 16; ++s) ds[b * 16 + s] *= dh[s];            // ds = d pre3 now
        }
        if constexpr (WHICH == 0) {
            // dW4 += gy (x) hid, db4 += column sums of gy
            __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0): the wave's own LDS writes have landed
            outer_accumulate<NE, NH, SA, SB>(tA, tB, j, half, acc);
            column_sums<DE, SA>(tA, lane, dbsum);
            // d xn = d e_out + W3^T d pre3; LayerNorm + residual backward (d e_out and ehat are re-read: they were released above)
            float dx[NE * 16];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int b = 0; b < NE; ++b) {
                const unsigned cur = A.o3t + (unsigned)b * KQ4 * 1024;
                const f32x16 t = mfma_block_p<KQ4>(wp, ws, cur, b + 1 < NE ? cur + KQ4 * 1024 : A.o3, ds, zero16());
                float sc[16], d2[16], d3[16], dob[16], xb[16];
                load16(ec2 + b * 32 + half * 16, sc);
                load16(A.dout + rc * DE + b * 32 + half * 16, dob);
                xhat_block(b, xb);
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float dxn = (valid ? dob[s] : 0.f) + t[s];
                    d2[s] = dxn;                                             // d es2
                    d3[s] = dxn * xb[s];                                     // d ec2
                    s2 = fmaf(dxn * (1.f + sc[s]), xb[s], s2);
                    dx[b * 16 + s] = dxn * (1.f + sc[s]);                    // gradient at the LayerNorm output
                }
                if (valid) {
                    store16(A.dmod_rows + ((size_t)row * 4 + 1) * DE + b * 32 + half * 16, d2);
                    store16(A.dmod_rows + ((size_t)row * 4 + 2) * DE + b * 32 + half * 16, d3);
                }
            }
#pragma unroll
            for (int i = 0; i < NE * 16; ++i) s1 += dx[i];
            const float m1 = pair_sum(s1) * (1.f / DE), m2 = pair_sum(s2) * (1.f / DE);
#pragma unroll
            for (int b = 0; b < NE; ++b) {
                float g[16], de[16], dh[16], dg[16], ehb[16], xb[16];
                load16(eg1 + b * 32 + half * 16, g);
                load16(A.ehat + rc * DE + b * 32 + half * 16, ehb);
                xhat_block(b, xb);
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float d1 = rstd * (dx[b * 16 + s] - m1 - xb[s] * m2);      // d x1
                    de[s] = d1; dh[s] = g[s] * d1; dg[s] = d1 * ehb[s];
                }
                if (valid) {
                    store16(A.d_e_in + (size_t)row * DE + b * 32 + half * 16, de);
                    store16(A.d_ehat + (size_t)row * DE + b * 32 + half * 16, dh);
                    store16(A.dmod_rows + ((size_t)row * 4 + 0) * DE + b * 32 + half * 16, dg);
                }
            }
        } else {
            // dW3 += d pre3 (x) xn, db3 += column sums of d pre3
            tile_store<NH, SB>(tB, j, half, ds);
            tile_store<NE, SA>(tA, j, half, xn);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            outer_accumulate<NH, NE, SB, SA>(tB, tA, j, half, acc);
            column_sums<HID, SB>(tB, lane, dbsum);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                                  // tile reads done before the next tile's writes (same wave: program order)
    }
    float* part = A.partial + (size_t)wave * PART;
    if constexpr (WHICH == 0) {
        write_tiles<NE, NH>(part + HID * DE + HID, HID, lane, acc);
        part[HID * DE + HID + DE * HID + lane] = dbsum[0];
    } else {
        write_tiles<NH, NE>(part, DE, lane, acc);
        part[HID * DE + lane] = dbsum[0];
        part[HID * DE + 64 + lane] = dbsum[WHICH == 1 ? 1 : 0];
    }
}

template <int WHICH>
__global__ __launch_bounds__(64, 1) void k_edge_ffn_bwd(BwdArgs A) {
    __shared__ float tA[32 * SA], tB[32 * SB];
    bwd_body<WHICH>(A, tA, tB);
}

// out[i] = sum over waves (fixed order) of partial[w][i]
__global__ void k_reduce_partials(const float* __restrict__ partial, int n_waves, float* dW3, float* db3, float* dW4, float* db4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PART) return;
    float s = 0.f;
    for (int w = 0; w < n_waves; ++w) s += partial[(size_t)w * PART + i];
    if (i < HID * DE) dW3[i] = s;
    else if (i < HID * DE + HID) db3[i - HID * DE] = s;
    else if (i < HID * DE + HID + DE * HID) dW4[i - HID * DE - HID] = s;
    else db4[i - HID * DE - HID - DE * HID] = s;
}

// d_mods[u][6 De]: chunks 2..5 (eg1, es2, ec2, eg2) = sums of the rows of modulation row u, chunks 0..1 (es1, ec1: not part
// of phase D) = 0.  Rows of one modulation row are contiguous: [mod_off[u], mod_off[u + 1]).
__global__ void k_segsum_mods(const float* __restrict__ dmod_rows, const int* __restrict__ mod_off, float* __restrict__ d_mods) {
    const int u = blockIdx.x, f = threadIdx.x;                // blockDim = 4 De
    const int c = f / DE, k = f % DE;
    float s = 0.f;
    for (int r = mod_off[u]; r < mod_off[u + 1]; ++r) s += dmod_rows[((size_t)r * 4 + c) * DE + k];
    d_mods[(size_t)u * 6 * DE + (2 + c) * DE + k] = s;
    if (c < 2) d_mods[(size_t)u * 6 * DE + c * DE + k] = 0.f;
}

}  // namespace

extern "C" size_t jodo_edge_ffn_backward_workspace(int rows, int De, int mlp_ratio) {
    if (De != DE || mlp_ratio != RR || rows <= 0) return 0;
    const int tiles = (rows + 31) / 32, n_waves = tiles < 512 ? tiles : 512;
    return ((size_t)rows * 4 * DE + (size_t)n_waves * PART) * sizeof(float);
}

extern "C" int jodo_edge_ffn_backward(int rows, int De, int mlp_ratio, int n_mod_rows, const float* e_in, const float* ehat,
                                      const int32_t* row_mod, const int32_t* mod_off, const float* mods, const float* packed_w,
                                      const int64_t* woff4, const float* b3, const float* b4, const float* d_out, float* d_e_in,
                                      float* d_ehat, float* d_mods, float* dW3, float* db3, float* dW4, float* db4, void* workspace,
                                      void* stream) {
    if (De != DE || mlp_ratio != RR)
        return jodo_set_error(JODO_ERR_UNSUPPORTED, "edge_ffn_backward: this slice covers De = %d, mlp_ratio = %d (the QM9 configs), got %d / %d",
                              DE, RR, De, mlp_ratio);
    if (rows <= 0 || n_mod_rows <= 0) return jodo_set_error(JODO_ERR_ARG, "edge_ffn_backward: bad shape");
    if (!e_in || !ehat || !row_mod || !mod_off || !mods || !packed_w || !woff4 || !b3 || !b4 || !d_out || !d_e_in || !d_ehat || !d_mods ||
        !dW3 || !db3 || !dW4 || !db4 || !workspace)
        return jodo_set_error(JODO_ERR_ARG, "edge_ffn_backward: null argument");
    const int tiles = (rows + 31) / 32, n_waves = tiles < 512 ? tiles : 512;   // two kinds of workgroups: 2 x 512 waves = one per SIMD
    BwdArgs A;
    A.rows = rows; A.U = n_mod_rows; A.n_waves = n_waves;
    A.e_in = e_in; A.ehat = ehat; A.mods = mods; A.dout = d_out; A.row_mod = row_mod;
    A.packed = packed_w;
    A.o3 = (unsigned)(woff4[0] * 4); A.o4 = (unsigned)(woff4[1] * 4); A.o4t = (unsigned)(woff4[2] * 4); A.o3t = (unsigned)(woff4[3] * 4);
    A.b3 = b3; A.b4 = b4; A.d_e_in = d_e_in; A.d_ehat = d_ehat;
    A.dmod_rows = static_cast<float*>(workspace);
    A.partial = A.dmod_rows + (size_t)rows * 4 * DE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_edge_ffn_bwd<0>, dim3(n_waves), dim3(64), 0, st, A);
    int rc = jodo_check_launch("k_edge_ffn_bwd<0>");
    if (rc != JODO_OK) return rc;
    hipLaunchKernelGGL(k_edge_ffn_bwd<1>, dim3(n_waves), dim3(64), 0, st, A);
    rc = jodo_check_launch("k_edge_ffn_bwd<1>");
    if (rc != JODO_OK) return rc;
    hipLaunchKernelGGL(k_reduce_partials, dim3((PART + 255) / 256), dim3(256), 0, st, A.partial, n_waves, dW3, db3, dW4, db4);
    rc = jodo_check_launch("k_reduce_partials");
    if (rc != JODO_OK) return rc;
    hipLaunchKernelGGL(k_segsum_mods, dim3(n_mod_rows), dim3(4 * DE), 0, st, A.dmod_rows, mod_off, d_mods);
    return jodo_check_launch("k_segsum_mods");
}
