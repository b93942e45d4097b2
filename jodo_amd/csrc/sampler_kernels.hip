// Fused caller-side kernels of the sampling loop: HBM-bound elementwise
// work on the dense API tensors, one pass each instead of ~25 small framework launches per step.
//
//   jodo_sampler_step   the ancestral update of AncestralSampler.sampling (sampling.py:536-589):
//                         mean = c_x * x_t + c_pred * pred          (nodes and edges)
//                         x_s  = mean + sigma * eps
//                       with eps built from raw N(0,1) draws exactly as the reference's noise samplers do
//                       (models/utils.py:67-99): node noise masked, its 3 position channels made
//                       centre-of-mass free per molecule; edge noise = strict lower triangle of a
//                       [B,ch,N,N] draw mirrored to the upper triangle, masked, diagonal zero.
//   jodo_decode         post_process + the per-molecule part of mol_process (sampling.py:53-97, :12-32,
//                       utils.py:71-105): undo the normalisation, atom type = argmax, formal charge =
//                       round, bond order by thresholds; compact u8/i8 outputs for ONE device->host copy.
//
// Both take the atom counts as a device int32[B] (masks are prefix masks, sampling.py:193-201); all
// tensors are the reference's dense row-major layouts.  Coalescing: one thread per innermost element,
// consecutive threads walk the contiguous last dimensions.
#include <hip/hip_runtime.h>
#include "../../include/jodo_hip.h"
#include "jodo_hip_internal.h"

namespace {

// ---- counter-based normal draws (SURVEY.md §8f row 1 "device noise"): Philox4x32-10 (Salmon et al., SC'11) keyed by the
// caller's 64-bit seed, counter = (element lo, element hi, draw index, stream id); four uniforms -> four normals by
// Box-Muller.  A value depends only on (seed, draw, stream, element): no generator state, so the same call replays in a
// captured hipGraph (draw index from a device counter), both triangle halves of the edge noise evaluate the SAME
// counter (exactly symmetric), and ranks with different seeds never share a stream.  oracle/philox_ref.py restates
// it in numpy (tests only).
struct Rng { unsigned long long seed; unsigned draw; unsigned draw_mul; int on; };
enum { RNG_POS = 0, RNG_FEAT = 1, RNG_EDGE = 2 };

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// four N(0,1) draws of element `elem` of stream `stream` at draw index `draw`
__device__ __forceinline__ void philox_normal4(unsigned long long seed, unsigned draw, unsigned stream, unsigned long long elem,
                                               float z[4]) {
    unsigned u[4];
    philox4x32_10((unsigned)elem, (unsigned)(elem >> 32), draw, stream, (unsigned)seed, (unsigned)(seed >> 32), u);
    // u1 in (0, 1], u2 in [0, 1): r = sqrt(-2 ln u1), angle = 2 pi u2
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float u1 = ((float)(u[2 * p] >> 8) + 1.0f) * (1.0f / 16777216.0f);
        const float u2 = (float)(u[2 * p + 1] >> 8) * (1.0f / 16777216.0f);
        const float r = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincospif(2.0f * u2, &sn, &cs);
        z[2 * p] = r * cs; z[2 * p + 1] = r * sn;
    }
}
__device__ __forceinline__ unsigned rng_draw(const Rng& g, const int* step) { return g.draw + (step ? (unsigned)(*step) * g.draw_mul : 0u); }
// raw position draw of atom (b, i): component f in 0..2
__device__ __forceinline__ void rng_pos3(const Rng& g, unsigned draw, size_t atom, float e[3]) {
    float z[4];
    philox_normal4(g.seed, draw, RNG_POS, atom, z);
    e[0] = z[0]; e[1] = z[1]; e[2] = z[2];
}
__device__ __forceinline__ float rng_feat(const Rng& g, unsigned draw, size_t atom, int k) {
    float z[4];
    philox_normal4(g.seed, draw, RNG_FEAT, atom * 64 + (size_t)(k >> 2), z);       // up to 256 feature channels per atom
    return z[k & 3];
}
// raw edge draw of the unordered pair (lo > hi) of molecule b, channel f < 4
__device__ __forceinline__ float rng_edge(const Rng& g, unsigned draw, size_t cell_lower, int f) {
    float z[4];
    philox_normal4(g.seed, draw, RNG_EDGE, cell_lower, z);
    return z[f & 3];
}

// one workgroup per molecule: mean / next state of the node tensor [N, F] (F = 3 + nd)
// coef (optional): device table [steps][4] = (c_x, c_pred, sigma, noise_level), row *step — lets a captured
// HIP graph of one sampling step be replayed for every step (host scalars would be baked into the graph)
__global__ __launch_bounds__(256) void k_step_nodes(int N, int F, const int* __restrict__ n_nodes, float c_x, float c_pred,
                                                    float sigma, const float* __restrict__ coef, const int* __restrict__ step,
                                                    const float* __restrict__ x, const float* __restrict__ pred,
                                                    const float* __restrict__ eps_pos, const float* __restrict__ eps_feat,
                                                    Rng rng, float* __restrict__ x_next, float* __restrict__ x_mean) {
    const int b = blockIdx.x, n = n_nodes[b], nd = F - 3;
    if (coef) { const float* c = coef + 4 * (size_t)(*step); c_x = c[0]; c_pred = c[1]; sigma = c[2]; }
    const unsigned draw = rng.on ? rng_draw(rng, coef ? step : nullptr) : 0u;
    __shared__ float red[3][256];
    // centre of mass of the masked position noise: sum over real atoms / n  (remove_mean_with_mask)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float ev[3];
        if (rng.on) rng_pos3(rng, draw, (size_t)b * N + i, ev);
        else { const float* e = eps_pos + ((size_t)b * N + i) * 3; ev[0] = e[0]; ev[1] = e[1]; ev[2] = e[2]; }
        s0 += ev[0]; s1 += ev[1]; s2 += ev[2];
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
            red[2][threadIdx.x] += red[2][threadIdx.x + w];
        }
        __syncthreads();
    }
    const float inv = 1.f / (float)n;
    const float m[3] = {red[0][0] * inv, red[1][0] * inv, red[2][0] * inv};
    for (int idx = threadIdx.x; idx < N * F; idx += blockDim.x) {
        const int i = idx / F, f = idx % F;
        const size_t g = (size_t)b * N * F + idx;
        // products rounded separately, then added: the op order of the framework expression c_x*x + c_pred*pred
        const float mean = __fadd_rn(__fmul_rn(c_x, x[g]), __fmul_rn(c_pred, pred[g]));
        float e = 0.f;
        if (i < n) {
            if (rng.on) {
                if (f < 3) { float ev[3]; rng_pos3(rng, draw, (size_t)b * N + i, ev); e = ev[f] - m[f]; }
                else e = rng_feat(rng, draw, (size_t)b * N + i, f - 3);
            } else e = f < 3 ? eps_pos[((size_t)b * N + i) * 3 + f] - m[f] : eps_feat[((size_t)b * N + i) * nd + (f - 3)];
        }
        x_mean[g] = mean;
        x_next[g] = __fadd_rn(mean, __fmul_rn(sigma, e));
    }
}

// one thread per (b, a, c, f) of the edge tensor [B, N, N, ch]; eps_edge is the raw draw [B, ch, N, N]
__global__ __launch_bounds__(256) void k_step_edges(int B, int N, int ch, const int* __restrict__ n_nodes, float c_x, float c_pred,
                                                    float sigma, const float* __restrict__ coef, const int* __restrict__ step,
                                                    const float* __restrict__ ex, const float* __restrict__ epred,
                                                    const float* __restrict__ eps, Rng rng, float* __restrict__ e_next,
                                                    float* __restrict__ e_mean) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t NN = (size_t)N * N, tot = (size_t)B * NN * ch;
    if (g >= tot) return;
    if (coef) { const float* c = coef + 4 * (size_t)(*step); c_x = c[0]; c_pred = c[1]; sigma = c[2]; }
    const int f = (int)(g % ch);
    const size_t cell = g / ch;
    const int c = (int)(cell % N), a = (int)((cell / N) % N), b = (int)(cell / NN);
    const int n = n_nodes[b];
    const float mean = __fadd_rn(__fmul_rn(c_x, ex[g]), __fmul_rn(c_pred, epred[g]));
    float e = 0.f;
    if (a < n && c < n && a != c) {
        const int lo = a > c ? a : c, hi = a > c ? c : a;          // strict lower triangle entry (row lo, col hi)
        e = rng.on ? rng_edge(rng, rng_draw(rng, coef ? step : nullptr), ((size_t)b * N + lo) * N + hi, f)
                   : eps[(((size_t)b * ch + f) * N + lo) * N + hi];
    }
    e_mean[g] = mean;
    e_next[g] = __fadd_rn(mean, __fmul_rn(sigma, e));
}

struct DecodeArgs {
    int B, N, atom_types, include_fc, ch, compress_edge, centered;
    float pos_norm, atom_norm, fc_norm, edge_norm;
};

// one thread per (b, i): positions, atom type, formal charge
__global__ __launch_bounds__(256) void k_decode_nodes(DecodeArgs A, const int* __restrict__ n_nodes, const float* __restrict__ xh,
                                                      float* __restrict__ pos, uint8_t* __restrict__ atom_type,
                                                      int8_t* __restrict__ fc) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= A.B * A.N) return;
    const int b = idx / A.N, i = idx % A.N;
    const int F = 3 + A.atom_types + (A.include_fc ? 1 : 0);
    const float* r = xh + (size_t)idx * F;
    const bool real = i < n_nodes[b];
    float* p = pos + (size_t)idx * 3;
    for (int k = 0; k < 3; ++k) p[k] = real ? r[k] * A.pos_norm : 0.f;
    // argmax over the inverse-scaled categories (an increasing affine map: same argmax; first maximum wins)
    int best = 0;
    float bv = -INFINITY;
    for (int k = 0; k < A.atom_types; ++k) {
        float v = r[3 + k] * A.atom_norm;
        if (A.centered) v = (v + 1.f) / 2.f;
        if (!real) v = 0.f;
        if (v > bv) { bv = v; best = k; }
    }
    atom_type[idx] = (uint8_t)best;
    float q = 0.f;
    if (A.include_fc && real) q = rintf(r[F - 1] * A.fc_norm);       // torch.round = round-half-to-even
    fc[idx] = (int8_t)q;
}

// one thread per (b, a, c): bond type
__global__ __launch_bounds__(256) void k_decode_edges(DecodeArgs A, const int* __restrict__ n_nodes, const float* __restrict__ ex,
                                                      uint8_t* __restrict__ edge_type) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t NN = (size_t)A.N * A.N;
    if (idx >= (size_t)A.B * NN) return;
    const int b = (int)(idx / NN), a = (int)((idx % NN) / A.N), c = (int)(idx % A.N);
    const int n = n_nodes[b];
    const bool real = a < n && c < n && a != c;
    const float* r = ex + idx * A.ch;
    float h[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < A.ch && k < 4; ++k) {
        float v = r[k] * A.edge_norm;
        if (A.centered) v = (v + 1.f) / 2.f;
        h[k] = real ? v : 0.f;
    }
    int t = 0;
    if (A.compress_edge) {
        const bool exist = h[0] >= 0.5f;
        const float o = h[1] * 3.f;
        int order = 0;
        if (o >= 0.5f) order = 1;
        if (o >= 1.5f) order = 2;
        if (o >= 2.5f) order = 3;
        t = exist ? order : 0;
        if (A.ch == 3 && exist && h[2] >= 0.5f && t == 0) t = 4;
    } else {
        bool any = false;
        int best = 0;
        float bv = -INFINITY;
        for (int k = 0; k < A.ch && k < 4; ++k) {
            any |= h[k] > 0.5f;
            if (h[k] > bv) { bv = h[k]; best = k; }
        }
        t = any ? best + 1 : 0;
    }
    edge_type[idx] = (uint8_t)t;
}

}  // namespace

namespace {
__global__ void k_step_begin(int B, const float* __restrict__ coef, const int* __restrict__ step, float* __restrict__ noise_level) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) noise_level[b] = coef[4 * (size_t)(*step) + 3];
}
__global__ void k_step_end(int* step) { *step += 1; }

int sampler_step_launch(int B, int N, int node_feats, int edge_ch, const int32_t* n_nodes_dev, float c_x, float c_pred, float sigma,
                        const float* coef, const int32_t* step, Rng rng, const float* x, const float* edge_x, const float* pred,
                        const float* edge_pred, const float* eps_pos, const float* eps_feat, const float* eps_edge, float* x_next,
                        float* edge_next, float* x_mean, float* edge_mean, void* stream) {
    if (B <= 0 || N <= 0 || node_feats < 4 || edge_ch < 1) return jodo_set_error(JODO_ERR_ARG, "sampler_step: bad shape");
    if (!n_nodes_dev || !x || !edge_x || !pred || !edge_pred || !x_next || !edge_next || !x_mean || !edge_mean)
        return jodo_set_error(JODO_ERR_ARG, "sampler_step: null argument");
    if (!rng.on && (!eps_pos || !eps_feat || !eps_edge)) return jodo_set_error(JODO_ERR_ARG, "sampler_step: null noise draw");
    if (rng.on && (edge_ch > 4 || node_feats - 3 > 256))
        return jodo_set_error(JODO_ERR_UNSUPPORTED, "sampler_step_rng: at most 4 edge channels and 256 node feature channels");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_step_nodes, dim3(B), dim3(256), 0, st, N, node_feats, n_nodes_dev, c_x, c_pred, sigma, coef, step, x, pred,
                       eps_pos, eps_feat, rng, x_next, x_mean);
    int rc = jodo_check_launch("k_step_nodes");
    if (rc != JODO_OK) return rc;
    const size_t tot = (size_t)B * N * N * edge_ch;
    hipLaunchKernelGGL(k_step_edges, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, B, N, edge_ch, n_nodes_dev, c_x, c_pred,
                       sigma, coef, step, edge_x, edge_pred, eps_edge, rng, edge_next, edge_mean);
    return jodo_check_launch("k_step_edges");
}
const Rng kNoRng{0ull, 0u, 0u, 0};
}  // namespace

extern "C" int jodo_sampler_step(int B, int N, int node_feats, int edge_ch, const int32_t* n_nodes_dev, float c_x, float c_pred,
                                 float sigma, const float* x, const float* edge_x, const float* pred, const float* edge_pred,
                                 const float* eps_pos, const float* eps_feat, const float* eps_edge, float* x_next,
                                 float* edge_next, float* x_mean, float* edge_mean, void* stream) {
    return sampler_step_launch(B, N, node_feats, edge_ch, n_nodes_dev, c_x, c_pred, sigma, nullptr, nullptr, kNoRng, x, edge_x, pred,
                               edge_pred, eps_pos, eps_feat, eps_edge, x_next, edge_next, x_mean, edge_mean, stream);
}

extern "C" int jodo_sampler_step_tab(int B, int N, int node_feats, int edge_ch, const int32_t* n_nodes_dev, const float* coef_tab_dev,
                                     const int32_t* step_dev, const float* x, const float* edge_x, const float* pred,
                                     const float* edge_pred, const float* eps_pos, const float* eps_feat, const float* eps_edge,
                                     float* x_next, float* edge_next, float* x_mean, float* edge_mean, void* stream) {
    if (!coef_tab_dev || !step_dev) return jodo_set_error(JODO_ERR_ARG, "sampler_step_tab: null table");
    return sampler_step_launch(B, N, node_feats, edge_ch, n_nodes_dev, 0.f, 0.f, 0.f, coef_tab_dev, step_dev, kNoRng, x, edge_x, pred,
                               edge_pred, eps_pos, eps_feat, eps_edge, x_next, edge_next, x_mean, edge_mean, stream);
}

extern "C" int jodo_sampler_step_rng(int B, int N, int node_feats, int edge_ch, const int32_t* n_nodes_dev, float c_x, float c_pred,
                                     float sigma, const float* coef_tab_dev, const int32_t* step_dev, uint64_t seed, uint32_t draw,
                                     const float* x, const float* edge_x, const float* pred, const float* edge_pred, float* x_next,
                                     float* edge_next, float* x_mean, float* edge_mean, void* stream) {
    if ((coef_tab_dev == nullptr) != (step_dev == nullptr))
        return jodo_set_error(JODO_ERR_ARG, "sampler_step_rng: pass both the device table and the step counter, or neither");
    return sampler_step_launch(B, N, node_feats, edge_ch, n_nodes_dev, c_x, c_pred, sigma, coef_tab_dev, step_dev, Rng{seed, draw, 1u, 1},
                               x, edge_x, pred, edge_pred, nullptr, nullptr, nullptr, x_next, edge_next, x_mean, edge_mean, stream);
}

extern "C" int jodo_step_begin(int B, const float* coef_tab_dev, const int32_t* step_dev, float* noise_level_out, void* stream) {
    if (B <= 0 || !coef_tab_dev || !step_dev || !noise_level_out) return jodo_set_error(JODO_ERR_ARG, "step_begin: bad argument");
    hipLaunchKernelGGL(k_step_begin, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, coef_tab_dev, step_dev, noise_level_out);
    return jodo_check_launch("k_step_begin");
}

extern "C" int jodo_step_end(int32_t* step_dev, void* stream) {
    if (!step_dev) return jodo_set_error(JODO_ERR_ARG, "step_end: null");
    hipLaunchKernelGGL(k_step_end, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    return jodo_check_launch("k_step_end");
}

// ---- hybrid DPM-Solver++ update (mix_dpm_solver.py:44-59 positions, :61-265 atom / charge / bond channels) ----------
//   positions   pos_out = cx * pos + cp * pos_pred + sigma * eps          (ancestral; eps: masked, centre-of-mass free)
//   the rest    out     = a * base - b * P - c * (c2 * (DA - DB))         (data-prediction DPM-Solver++; c = 0: first order)
// coefficients: {cx, cp, sigma, a, b, c, c2, noise_level} by value, or row (*step) * stride + col of a device table
// (captured HIP graphs).  Products are rounded one by one in the order of the framework expressions.
namespace {
struct DpmCoef { float cx, cp, sigma, a, b, c, c2, nl; };

__device__ __forceinline__ DpmCoef dpm_coef(DpmCoef k, const float* tab, const int* step, int stride, int col) {
    if (tab) {
        const float* r = tab + (size_t)(*step) * stride + col;
        k.cx = r[0]; k.cp = r[1]; k.sigma = r[2]; k.a = r[3]; k.b = r[4]; k.c = r[5]; k.c2 = r[6]; k.nl = r[7];
    }
    return k;
}
__device__ __forceinline__ float dpm_value(const DpmCoef& k, float base, float p, float da, float db) {
    float v = __fsub_rn(__fmul_rn(k.a, base), __fmul_rn(k.b, p));
    if (k.c != 0.f) {
        float d = __fsub_rn(da, db);
        if (k.c2 != 1.f) d = __fmul_rn(k.c2, d);
        v = __fsub_rn(v, __fmul_rn(k.c, d));
    }
    return v;
}

// one workgroup per molecule, node tensor [N, F]
__global__ __launch_bounds__(256) void k_dpm_nodes(int N, int F, const int* __restrict__ n_nodes, DpmCoef k, const float* __restrict__ tab,
                                                   const int* __restrict__ step, int stride, int col, const float* __restrict__ x_pos,
                                                   const float* __restrict__ x_base, const float* __restrict__ P,
                                                   const float* __restrict__ DA, const float* __restrict__ DB,
                                                   const float* __restrict__ PP, const float* __restrict__ eps_pos,
                                                   Rng rng, float* __restrict__ out) {
    const int b = blockIdx.x, n = n_nodes[b];
    k = dpm_coef(k, tab, step, stride, col);
    const unsigned draw = rng.on ? rng_draw(rng, tab ? step : nullptr) : 0u;
    __shared__ float red[3][256];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (k.sigma != 0.f)
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            float ev[3];
            if (rng.on) rng_pos3(rng, draw, (size_t)b * N + i, ev);
            else { const float* e = eps_pos + ((size_t)b * N + i) * 3; ev[0] = e[0]; ev[1] = e[1]; ev[2] = e[2]; }
            s0 += ev[0]; s1 += ev[1]; s2 += ev[2];
        }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
            red[2][threadIdx.x] += red[2][threadIdx.x + w];
        }
        __syncthreads();
    }
    const float inv = 1.f / (float)n;
    const float m[3] = {red[0][0] * inv, red[1][0] * inv, red[2][0] * inv};
    for (int idx = threadIdx.x; idx < N * F; idx += blockDim.x) {
        const int i = idx / F, f = idx % F;
        const size_t g = (size_t)b * N * F + idx;
        float v;
        if (f < 3) {
            v = __fadd_rn(__fmul_rn(k.cx, x_pos[g]), __fmul_rn(k.cp, PP[g]));
            if (k.sigma != 0.f) {                               // last update of a round: no noise (last_step)
                float e = 0.f;
                if (i < n) {
                    if (rng.on) { float ev[3]; rng_pos3(rng, draw, (size_t)b * N + i, ev); e = ev[f] - m[f]; }
                    else e = eps_pos[((size_t)b * N + i) * 3 + f] - m[f];
                }
                v = __fadd_rn(v, __fmul_rn(k.sigma, e));
            }
        } else {
            v = dpm_value(k, x_base[g], P[g], DA[g], DB[g]);
        }
        out[g] = v;
    }
}

__global__ __launch_bounds__(256) void k_dpm_edges(size_t tot, DpmCoef k, const float* __restrict__ tab, const int* __restrict__ step,
                                                   int stride, int col, const float* __restrict__ base, const float* __restrict__ P,
                                                   const float* __restrict__ DA, const float* __restrict__ DB, float* __restrict__ out) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= tot) return;
    k = dpm_coef(k, tab, step, stride, col);
    out[g] = dpm_value(k, base[g], P[g], DA[g], DB[g]);
}

__global__ void k_step_begin_at(int B, const float* __restrict__ tab, const int* __restrict__ step, int stride, int col,
                                float* __restrict__ noise_level) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) noise_level[b] = tab[(size_t)(*step) * stride + col + 7];
}
}  // namespace

namespace {
int dpm_update_launch(int B, int N, int node_feats, int edge_ch, const int32_t* n_nodes_dev, const float* coef8_host,
                      const float* coef_tab_dev, const int32_t* step_dev, int tab_stride, int tab_col, Rng rng, const float* x_pos,
                      const float* x_base, const float* edge_base, const float* P, const float* eP, const float* DA, const float* eDA,
                      const float* DB, const float* eDB, const float* PP, const float* eps_pos, float* x_out, float* edge_out,
                      void* stream) {
    if (B <= 0 || N <= 0 || node_feats < 4 || edge_ch < 1) return jodo_set_error(JODO_ERR_ARG, "dpm_update: bad shape");
    if (!n_nodes_dev || !x_pos || !x_base || !edge_base || !P || !eP || !DA || !eDA || !DB || !eDB || !PP || !x_out || !edge_out)
        return jodo_set_error(JODO_ERR_ARG, "dpm_update: null argument");
    if (!rng.on && !eps_pos) return jodo_set_error(JODO_ERR_ARG, "dpm_update: null noise draw");
    if ((coef8_host == nullptr) == (coef_tab_dev == nullptr) || (coef_tab_dev && !step_dev))
        return jodo_set_error(JODO_ERR_ARG, "dpm_update: pass either host coefficients or a device table + step counter");
    DpmCoef k{0, 0, 0, 0, 0, 0, 1, 0};
    if (coef8_host) k = DpmCoef{coef8_host[0], coef8_host[1], coef8_host[2], coef8_host[3], coef8_host[4], coef8_host[5], coef8_host[6], coef8_host[7]};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_dpm_nodes, dim3(B), dim3(256), 0, st, N, node_feats, n_nodes_dev, k, coef_tab_dev, step_dev, tab_stride, tab_col,
                       x_pos, x_base, P, DA, DB, PP, eps_pos, rng, x_out);
    int rc = jodo_check_launch("k_dpm_nodes");
    if (rc != JODO_OK) return rc;
    const size_t tot = (size_t)B * N * N * edge_ch;
    hipLaunchKernelGGL(k_dpm_edges, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, tot, k, coef_tab_dev, step_dev, tab_stride,
                       tab_col, edge_base, eP, eDA, eDB, edge_out);
    return jodo_check_launch("k_dpm_edges");
}
}  // namespace

extern "C" int jodo_dpm_update(int B, int N, int node_feats, int edge_ch, const int32_t* n_nodes_dev, const float* coef8_host,
                               const float* coef_tab_dev, const int32_t* step_dev, int tab_stride, int tab_col, const float* x_pos,
                               const float* x_base, const float* edge_base, const float* P, const float* eP, const float* DA,
                               const float* eDA, const float* DB, const float* eDB, const float* PP, const float* eps_pos,
                               float* x_out, float* edge_out, void* stream) {
    return dpm_update_launch(B, N, node_feats, edge_ch, n_nodes_dev, coef8_host, coef_tab_dev, step_dev, tab_stride, tab_col, kNoRng,
                             x_pos, x_base, edge_base, P, eP, DA, eDA, DB, eDB, PP, eps_pos, x_out, edge_out, stream);
}

extern "C" int jodo_dpm_update_rng(int B, int N, int node_feats, int edge_ch, const int32_t* n_nodes_dev, const float* coef8_host,
                                   const float* coef_tab_dev, const int32_t* step_dev, int tab_stride, int tab_col, uint64_t seed,
                                   uint32_t draw, uint32_t draw_mul, const float* x_pos, const float* x_base, const float* edge_base,
                                   const float* P, const float* eP, const float* DA, const float* eDA, const float* DB,
                                   const float* eDB, const float* PP, float* x_out, float* edge_out, void* stream) {
    return dpm_update_launch(B, N, node_feats, edge_ch, n_nodes_dev, coef8_host, coef_tab_dev, step_dev, tab_stride, tab_col,
                             Rng{seed, draw, draw_mul, 1}, x_pos, x_base, edge_base, P, eP, DA, eDA, DB, eDB, PP, nullptr, x_out,
                             edge_out, stream);
}

extern "C" int jodo_step_begin_at(int B, const float* coef_tab_dev, const int32_t* step_dev, int tab_stride, int tab_col,
                                  float* noise_level_out, void* stream) {
    if (B <= 0 || !coef_tab_dev || !step_dev || !noise_level_out) return jodo_set_error(JODO_ERR_ARG, "step_begin_at: bad argument");
    hipLaunchKernelGGL(k_step_begin_at, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, coef_tab_dev, step_dev, tab_stride,
                       tab_col, noise_level_out);
    return jodo_check_launch("k_step_begin_at");
}

extern "C" int jodo_decode(int B, int N, int atom_types, int include_fc, int edge_ch, int compress_edge, int centered,
                           float pos_norm, float atom_norm, float fc_norm, float edge_norm, const int32_t* n_nodes_dev,
                           const float* xh, const float* edge_x, float* pos_out, uint8_t* atom_type_out, int8_t* fc_out,
                           uint8_t* edge_type_out, void* stream) {
    if (B <= 0 || N <= 0 || atom_types < 1 || atom_types > 255 || edge_ch < 1 || edge_ch > 4)
        return jodo_set_error(JODO_ERR_ARG, "decode: bad shape");
    if (!n_nodes_dev || !xh || !edge_x || !pos_out || !atom_type_out || !fc_out || !edge_type_out)
        return jodo_set_error(JODO_ERR_ARG, "decode: null argument");
    DecodeArgs A{B, N, atom_types, include_fc, edge_ch, compress_edge, centered, pos_norm, atom_norm, fc_norm, edge_norm};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_decode_nodes, dim3((B * N + 255) / 256), dim3(256), 0, st, A, n_nodes_dev, xh, pos_out, atom_type_out, fc_out);
    int rc = jodo_check_launch("k_decode_nodes");
    if (rc != JODO_OK) return rc;
    const size_t tot = (size_t)B * N * N;
    hipLaunchKernelGGL(k_decode_edges, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, A, n_nodes_dev, edge_x, edge_type_out);
    return jodo_check_launch("k_decode_edges");
}
