// The small kernels of a training step outside the network: optimiser update, clipping decision, Kabsch rotations.
// Optimiser side ( /root/reference/losses.py:14-26 get_optimizer, :29-50 gradient_clipping,
// :75-94 optimization_manager) on FLAT buffers: every parameter of the module is a slice of one allocation (jodo_amd/optim.py
// flatten_parameters), the gradients already are (jodo_amd/train.py), so the update is ONE elementwise kernel over ~5.6 M floats
// instead of a multi-tensor pass over 351 tensors whose host side (list building, per-tensor state look-ups) cost 3 ms of a 23 ms step —
// and the adaptive clipping's history lives on the device, so that a step has no host synchronisation left between its backward and the
// next batch.  Both are HBM streams: p, g, m, v, vmax read + p, m, v, vmax written = 36 bytes per parameter.
// Loss side (losses.py:424-434 kabsch_batch): the rotation of the Kabsch alignment per molecule, in place of torch.linalg.svd — whose
// error check reads `info` back and was the last host synchronisation of a step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/jodo_hip.h"
#include "jodo_hip_internal.h"

namespace {

struct AdamArgs { float decay, wd, w1, beta2, w2, eps, bc2_sqrt, step_size; int decoupled, amsgrad; };   // scalars formed in double on the host, as torch's Python side does

// torch.optim.Adam / AdamW (single-tensor formulas of torch/optim/adam.py, adamw.py), one element:
//   AdamW: p *= 1 - lr wd          Adam: g += wd p
//   m = m + (1 - beta1) (g - m)    v = beta2 v + (1 - beta2) g g
//   amsgrad: vmax = max(vmax, v), denom = sqrt(vmax) / sqrt(1 - beta2^t) + eps     else denom = sqrt(v) / sqrt(1 - beta2^t) + eps
//   p -= (lr / (1 - beta1^t)) m / denom
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float& vm, const AdamArgs& a) {
    if (a.decoupled) p *= a.decay; else g += a.wd * p;       // decay = 1 - lr wd
    m = m + a.w1 * (g - m);                                  // w1 = 1 - beta1 (exp_avg.lerp_)
    v = a.beta2 * v + a.w2 * g * g;                          // w2 = 1 - beta2
    float d;
    if (a.amsgrad) { vm = fmaxf(vm, v); d = sqrtf(vm) / a.bc2_sqrt + a.eps; }
    else d = sqrtf(v) / a.bc2_sqrt + a.eps;
    p -= a.step_size * (m / d);                              // step_size = lr / (1 - beta1^t)
}

__global__ __launch_bounds__(256) void k_adam(long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                              float* __restrict__ vmax, AdamArgs a) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x, i = q * 4;
    if (i + 3 < n) {
        float4 P = reinterpret_cast<float4*>(p)[q], M = reinterpret_cast<float4*>(m)[q], V = reinterpret_cast<float4*>(v)[q];
        const float4 G = reinterpret_cast<const float4*>(g)[q];
        float4 X = a.amsgrad ? reinterpret_cast<float4*>(vmax)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        adam_one(P.x, G.x, M.x, V.x, X.x, a); adam_one(P.y, G.y, M.y, V.y, X.y, a);
        adam_one(P.z, G.z, M.z, V.z, X.z, a); adam_one(P.w, G.w, M.w, V.w, X.w, a);
        reinterpret_cast<float4*>(p)[q] = P; reinterpret_cast<float4*>(m)[q] = M; reinterpret_cast<float4*>(v)[q] = V;
        if (a.amsgrad) reinterpret_cast<float4*>(vmax)[q] = X;
    } else {
        for (long j = i; j < n; ++j) {
            float vm = a.amsgrad ? vmax[j] : 0.f;
            adam_one(p[j], g[j], m[j], v[j], vm, a);
            if (a.amsgrad) vmax[j] = vm;
        }
    }
}

// gradient_clipping (losses.py:29-50) with its history on the device.  state: double[52] = the last (at most 50) pushed norms, their
// count, the next slot.  One thread:  allowed = min(1.5 mean + 2 std, max_grad) over the history (population std, as numpy's);
// coef = min(1, allowed / (norm + 1e-6)) (torch.nn.utils.clip_grad_norm_'s formula, in float like its tensor arithmetic);
// history <- min(norm, allowed).  The caller scales the flat gradient by *coef.
__global__ void k_gradnorm_clip(const float* __restrict__ norm, double* __restrict__ st, double max_grad, float* __restrict__ coef, float* __restrict__ allowed_out) {
    if (threadIdx.x || blockIdx.x) return;
    const int cnt = (int)st[50];
    double mean = 0.0;
    for (int i = 0; i < cnt; ++i) mean += st[i];
    mean /= (double)cnt;
    double var = 0.0;
    for (int i = 0; i < cnt; ++i) var += (st[i] - mean) * (st[i] - mean);
    const double sd = sqrt(var / (double)cnt);
    double allowed = 1.5 * mean + 2.0 * sd;
    allowed = allowed < max_grad ? allowed : max_grad;
    const float nrm = *norm;
    const float c = (float)allowed / (nrm + 1e-6f);
    *coef = c < 1.f ? c : 1.f;
    *allowed_out = (float)allowed;
    const double pushed = (double)nrm < allowed ? (double)nrm : allowed;
    const int slot = (int)st[51];
    st[slot] = pushed;
    st[50] = (double)(cnt < 50 ? cnt + 1 : 50);
    st[51] = (double)((slot + 1) % 50);
}

// Kabsch rotation of one molecule from its 3 x 3 covariance A = P^T Q (losses.py:426-432):  A = U S V^T,  R = U diag(1, 1, sign det A) V^T.
// One thread per molecule, in double: Jacobi eigen-decomposition of A^T A gives V (a proper rotation, columns ordered by descending
// eigenvalue) ; u1 = A v1 / |.|, u2 = A v2 made orthogonal to u1, and the third pair follows from the other two — with V proper,
// sign(det A) u3 = u1 x u2 whatever the sign of det A:   R = u1 v1^T + u2 v2^T + (u1 x u2) v3^T.
// (det A == 0 exactly — an all-zero A — drops the third term like the reference's sign(0) = 0.)
__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
__global__ void k_kabsch(int B, const float* __restrict__ Ain, float* __restrict__ Rout) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double A[3][3], K[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = (double)Ain[b * 9 + i * 3 + j];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) K[i][j] = A[0][i] * A[0][j] + A[1][i] * A[1][j] + A[2][i] * A[2][j];
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = K[0][1] * K[0][1] + K[0][2] * K[0][2] + K[1][2] * K[1][2];
        const double dia = K[0][0] * K[0][0] + K[1][1] * K[1][1] + K[2][2] * K[2][2];
        if (off <= 1e-40 * dia || off == 0.0) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (K[p][q] == 0.0) continue;
                const double theta = (K[q][q] - K[p][p]) / (2.0 * K[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double kp = K[k][p], kq = K[k][q]; K[k][p] = c * kp - s * kq; K[k][q] = s * kp + c * kq; }
                for (int k = 0; k < 3; ++k) { const double kp = K[p][k], kq = K[q][k]; K[p][k] = c * kp - s * kq; K[q][k] = s * kp + c * kq; }
                for (int k = 0; k < 3; ++k) { const double vp = V[k][p], vq = V[k][q]; V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq; }
            }
    }
    // columns by descending eigenvalue
    int o[3] = {0, 1, 2};
    double ev[3] = {K[0][0], K[1][1], K[2][2]};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2 - i; ++j)
        if (ev[o[j]] < ev[o[j + 1]]) { const int tmp = o[j]; o[j] = o[j + 1]; o[j + 1] = tmp; }
    double v1[3], v2[3], v3[3], u1[3], u2[3], u3[3];
    for (int k = 0; k < 3; ++k) { v1[k] = V[k][o[0]]; v2[k] = V[k][o[1]]; }
    cross3(v1, v2, v3);                                        // V proper
    for (int i = 0; i < 3; ++i) { u1[i] = A[i][0] * v1[0] + A[i][1] * v1[1] + A[i][2] * v1[2]; u2[i] = A[i][0] * v2[0] + A[i][1] * v2[1] + A[i][2] * v2[2]; }
    double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    if (n1 > 0.0) { for (int i = 0; i < 3; ++i) u1[i] /= n1; } else { u1[0] = 1.0; u1[1] = u1[2] = 0.0; }
    const double d12 = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
    for (int i = 0; i < 3; ++i) u2[i] -= d12 * u1[i];
    double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    if (n2 > 1e-150) { for (int i = 0; i < 3; ++i) u2[i] /= n2; }
    else {                                                     // rank one: any unit vector orthogonal to u1
        const int m = fabs(u1[0]) <= fabs(u1[1]) ? (fabs(u1[0]) <= fabs(u1[2]) ? 0 : 2) : (fabs(u1[1]) <= fabs(u1[2]) ? 1 : 2);
        double e[3] = {0, 0, 0}; e[m] = 1.0;
        cross3(u1, e, u2);
        n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
        for (int i = 0; i < 3; ++i) u2[i] /= n2;
    }
    cross3(u1, u2, u3);
    const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    const double w3 = det == 0.0 ? 0.0 : 1.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rout[b * 9 + i * 3 + j] = (float)(u1[i] * v1[j] + u2[i] * v2[j] + w3 * u3[i] * v3[j]);
}

}  // namespace

extern "C" {

int jodo_adam_step(int64_t n, float* p, const float* g, float* m, float* v, float* vmax, double lr, double beta1, double beta2, double eps, double weight_decay,
                   int64_t step, int decoupled, int amsgrad, void* stream) {
    if (n <= 0 || !p || !g || !m || !v || (amsgrad && !vmax)) return jodo_set_error(JODO_ERR_ARG, "jodo_adam_step: null / empty argument");
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vmax) & 15)
        return jodo_set_error(JODO_ERR_ARG, "jodo_adam_step: buffers must be 16-byte aligned");
    if (step < 1) return jodo_set_error(JODO_ERR_ARG, "jodo_adam_step: step counts from 1");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamArgs a{(float)(1.0 - lr * weight_decay), (float)weight_decay, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)sqrt(bc2),
               (float)(lr / bc1), decoupled, amsgrad};
    const long quads = (n + 3) / 4;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)n, p, g, m, v, vmax, a);
    return jodo_check_launch("k_adam");
}

int jodo_kabsch_rotations(int B, const float* A_dev, float* R_dev, void* stream) {
    if (B <= 0 || !A_dev || !R_dev) return jodo_set_error(JODO_ERR_ARG, "jodo_kabsch_rotations: null / empty argument");
    hipLaunchKernelGGL(k_kabsch, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, B, A_dev, R_dev);
    return jodo_check_launch("k_kabsch");
}

int jodo_gradnorm_clip(const float* norm_dev, double* state_dev, double max_grad, float* coef_dev, float* allowed_dev, void* stream) {
    if (!norm_dev || !state_dev || !coef_dev || !allowed_dev) return jodo_set_error(JODO_ERR_ARG, "jodo_gradnorm_clip: null argument");
    hipLaunchKernelGGL(k_gradnorm_clip, dim3(1), dim3(64), 0, (hipStream_t)stream, norm_dev, state_dev, max_grad, coef_dev, allowed_dev);
    return jodo_check_launch("k_gradnorm_clip");
}

}
