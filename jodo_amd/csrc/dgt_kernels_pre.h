// Prologue kernels: time embedding / modulation vectors, input packing (node / edge embeddings: dgt_kernels_wide.h).
// Reference: DGT_concat.forward models/mol_gnn.py:509-557, time_mlp :481-489 + layers.py:283-288,
// Cond_DGT_concat context path :728-734.
#pragma once
#include "dgt_kernels_common.h"

namespace jd {

// flags: zero + decide whether all molecules share one time row (uncond model, equal noise levels)
__global__ void k_flags_init(KArgs A) {
    __shared__ int differs;
    if (threadIdx.x == 0) differs = 0;
    __syncthreads();
    const float x0 = A.noise[0];
    int d = 0;
    for (int b = threadIdx.x; b < A.pd.B; b += blockDim.x) d |= (A.noise[b] != x0);
    if (d) atomicOr(&differs, 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        A.flags[FLAG_NAN] = 0;
        A.flags[FLAG_FIRST] = 0;
        A.flags[FLAG_COND_NONZERO] = 0;
        A.flags[FLAG_ASYM] = (A.force_directed || A.pin_sym == 2) ? 1 : 0;
        A.flags[FLAG_UNIFORM_T] = (A.d.cond_ch == 0 && !differs) ? 1 : 0;
    }
}

// Symmetry test of the caller's edge tensors (edge_x, cond_edge_x): the samplers always pass symmetric
// tensors (symmetric noise, symmetrised predictions), which makes the edge hidden state exactly
// symmetric and lets the pair kernels (dgt_kernels_sym.h) do the symmetric work once per unordered
// pair.  Anything else selects the directed kernels — decided on the device, no host sync.
__global__ void k_check_sym(KArgs A) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (b, a, c)
    const size_t NN = (size_t)A.pd.N * A.pd.N;
    if (idx >= (size_t)A.pd.B * NN) return;
    const int b = (int)(idx / NN);
    const int a = (int)((idx % NN) / A.pd.N), c = (int)(idx % A.pd.N);
    const int n = A.pd.orig_n[b], ch = A.d.ch;
    if (a >= n || c >= n || a >= c) return;
    const size_t r1 = idx * ch, r2 = ((size_t)b * NN + (size_t)c * A.pd.N + a) * ch;
    bool diff = false;
    for (int f = 0; f < ch; ++f) {
        // (a NaN on both sides counts as equal: a state that went NaN stays on the pair path — also under a symmetric pin — and
        //  ends in the NaN guard like the reference's, mol_gnn.py:587-589, instead of as a pin violation)
        const float x1 = A.edge_x[r1 + f], x2 = A.edge_x[r2 + f];
        diff |= !(x1 == x2 || (x1 != x1 && x2 != x2));
        if (A.cond_edge_x) {
            const float c1 = A.cond_edge_x[r1 + f], c2 = A.cond_edge_x[r2 + f];
            diff |= !(c1 == c2 || (c1 != c1 && c2 != c2));
        }
    }
    if (diff) atomicOr(&A.flags[FLAG_ASYM], 1);
}

// hid1[b] = GELU(W1 * [x, sin(2 pi x w), cos(2 pi x w)] + b1)      (LearnedSinusodialposEmb + Linear + GELU)
__global__ void k_time1(KArgs A) {
    const int b = blockIdx.x;
    if (A.flags[FLAG_UNIFORM_T] && b > 0) return;
    __shared__ float ft[17];
    const float x = A.noise[b];
    if (threadIdx.x < 8) {
        const float fr = x * A.W[A.wg[JW_TIME_FREQ] + threadIdx.x] * 2.f * 3.14159265358979323846f;
        ft[1 + threadIdx.x] = sinf(fr);
        ft[9 + threadIdx.x] = cosf(fr);
    }
    if (threadIdx.x == 0) ft[0] = x;
    __syncthreads();
    const float* W1 = A.W + A.wg[JW_TIME_W1];
    const float* b1 = A.W + A.wg[JW_TIME_B1];
    for (int f = threadIdx.x; f < A.d.T; f += blockDim.x) {
        float acc = b1[f];
#pragma unroll
        for (int k = 0; k < 17; ++k) acc = fmaf(W1[f * 17 + k], ft[k], acc);
        A.hid1[(size_t)b * A.d.T + f] = gelu_erf(acc);
    }
}

// condh[(b*cc + c)][f] = GELU(w0[f] * context[b][c] + b0[f])   (cond_mlp.0 + GELU)
__global__ void k_cond1(KArgs A) {
    const int row = blockIdx.x;               // b * cond_ch + c
    const float x = A.context[row];
    const float* w0 = A.W + A.wg[JW_COND_W0];
    const float* b0 = A.W + A.wg[JW_COND_B0];
    for (int f = threadIdx.x; f < A.d.D; f += blockDim.x)
        A.condh[(size_t)row * A.d.D + f] = gelu_erf(fmaf(w0[f], x, b0[f]));
}

// Generic row GEMM on the strip model: Y[row, :] (+)= act_in(X[row, :]) * W^T + bias
// rows in lanes, K in chunks of 64 features, up to 4 output blocks per wave.  The weights go through the software-pipelined
// ring of dgt_device.h (one group of 8 quads in flight across the (chunk, block) sequence) and the next activation chunk is
// requested while the current one is multiplied: with per-molecule modulation rows (conditional model, B = 1250: 60 GFLOP per
// forward) the plain load -> wait -> MFMA form ran at 40 % of the matrix peak.  Ysilu (optional): SiLU of the result as a second
// output, so that the modulation projection reads SiLU(time_emb) instead of recomputing it in every one of its waves.
struct RowGemmArgs {
    const float* X; int64_t ldx;
    float* Y; int64_t ldy;
    const float* Wp; const float* bias;     // packed [NB][K/8][64][4]; bias in slot (= natural) order
    int rows, K, NB;
    int in_act;                              // 0 none, 1 SiLU
    int accumulate;                          // Y += result
    const int* uniform_flag;                 // if non-null and *flag != 0: only row 0 is computed
    int nob;                                 // output blocks per wave (1 for few rows: more waves; 4 otherwise)
    int groups_in_x;                         // grid = (output groups, row strips) instead of (row strips, output groups), see k_rowgemm
    float* Ysilu;                            // nullable: SiLU(Y) [rows, ldy]
};

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_rowgemm(RowGemmArgs G) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int rows = (G.uniform_flag && *G.uniform_flag) ? 1 : G.rows;
    // With a shared time row only strip 0 works.  Workgroups go to the 8 XCDs round-robin in dispatch order (x fastest): with the
    // strips in x, strip 0's workgroups are the ids 0, S, 2 S, ... — for S = 16 (GEOM B = 512) or 40 (B = 1250) all on XCD 0, and the
    // modulation projection ran on 32 of the 256 CUs (322 us per forward at GEOM nf 256; 43 us at QM9, where S = 79 is odd).
    const int r0 = (G.groups_in_x ? blockIdx.y : blockIdx.x) * 32;
    if (r0 >= rows) return;
    const int row = r0 + j;
    const int rowc = row < rows ? row : rows - 1;
    const int ob0 = (G.groups_in_x ? blockIdx.x : blockIdx.y) * G.nob;
    const int nv = G.NB - ob0 < G.nob ? G.NB - ob0 : G.nob;          // output blocks of this wave (1..4)
    const int kq = G.K / 8, nch = G.K / 64;  // quads per output block, activation chunks
    f32x16 acc[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[o] = zero16();
    const WSrc ws = make_wsrc(G.Wp, lane);
    auto woff = [&](int o, int c) { return (unsigned)(((size_t)(ob0 + o) * kq + (size_t)c * 8) * 1024); };
    WPipe<8> wp;
    wpipe_prime(wp, ws, woff(0, 0));
    const float* xrow = G.X + (size_t)rowc * G.ldx;
    float xn[32];
    load_nat<2>(xrow, half, xn);
    for (int c = 0; c < nch; ++c) {
        float x[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) x[s] = xn[s];
        if (c + 1 < nch) load_nat<2>(xrow + (c + 1) * 64, half, xn);      // next chunk: in flight behind this chunk's MFMAs
        if (G.in_act == 1) {
#pragma unroll
            for (int s = 0; s < 32; ++s) x[s] = silu_f(x[s]);
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o < nv) {
                const bool last_o = o + 1 >= nv;
                const unsigned nxt = !last_o ? woff(o + 1, c) : (c + 1 < nch ? woff(0, c + 1) : woff(0, 0));
                acc[o] = mfma_block_p<8>(wp, ws, woff(o, c), nxt, x, acc[o]);
            }
        }
    }
    if (row >= rows) return;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        if (o < nv) {
            float r[16];
            acc_bias(acc[o], G.bias + (ob0 + o) * 32 + half * 16, r);
            float* yp = G.Y + (size_t)row * G.ldy + (ob0 + o) * 32 + half * 16;
            if (G.accumulate) {
                float old[16];
                load16(yp, old);
#pragma unroll
                for (int s = 0; s < 16; ++s) r[s] += old[s];
            }
            store16(yp, r);
            if (G.Ysilu) {
#pragma unroll
                for (int s = 0; s < 16; ++s) r[s] = silu_f(r[s]);
                store16(G.Ysilu + (size_t)row * G.ldy + (ob0 + o) * 32 + half * 16, r);
            }
        }
    }
}

// gather node inputs into the packed layout; detect "any self-cond distance non-zero"
__global__ void k_pack_nodes(KArgs A) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= A.pd.Nn_pad) return;
    const int nd = A.d.nd, ndp = A.d.ndp, dims = 3 + nd;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), cp = p;
    float* ft = A.feat + (size_t)v * ndp;
    for (int f = 0; f < ndp; ++f) ft[f] = 0.f;
    if (v < A.pd.Nn) {
        const int b = A.pd.node_b[v], i = A.pd.node_i[v];
        const float* x = A.xh + ((size_t)b * A.pd.N + i) * dims;
        p = make_float4(x[0], x[1], x[2], 0.f);
        for (int f = 0; f < nd; ++f) ft[f] = x[3 + f];
        if (A.cond_x) {
            const float* c = A.cond_x + ((size_t)b * A.pd.N + i) * dims;
            cp = make_float4(c[0], c[1], c[2], 0.f);
            for (int f = 0; f < nd; ++f) ft[nd + f] = c[3 + f];
            const float* c0 = A.cond_x + ((size_t)b * A.pd.N) * dims;      // first atom of the molecule
            if (c[0] != c0[0] || c[1] != c0[1] || c[2] != c0[2]) atomicOr(&A.flags[FLAG_COND_NONZERO], 1);
        }
    }
    reinterpret_cast<float4*>(A.pos_in)[v] = p;
    reinterpret_cast<float4*>(A.cpos)[v] = cp;
}

}  // namespace jd
