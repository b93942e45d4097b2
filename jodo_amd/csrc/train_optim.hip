// Optimiser side of the training step (/root/reference/losses.py:14-26 get_optimizer, :29-50 gradient_clipping,
// :75-94 optimization_manager) on FLAT buffers: every parameter of the module is a slice of one allocation (jodo_amd/optim.py
// flatten_parameters), the gradients already are (jodo_amd/train.py), so the update is ONE elementwise kernel over ~5.6 M floats
// instead of a multi-tensor pass over 351 tensors whose host side (list building, per-tensor state look-ups) cost 3 ms of a 23 ms step —
// and the adaptive clipping's history lives on the device, so that a step has no host synchronisation left between its backward and the
// next batch.  Both are HBM streams: p, g, m, v, vmax read + p, m, v, vmax written = 36 bytes per parameter.
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

int jodo_gradnorm_clip(const float* norm_dev, double* state_dev, double max_grad, float* coef_dev, float* allowed_dev, void* stream) {
    if (!norm_dev || !state_dev || !coef_dev || !allowed_dev) return jodo_set_error(JODO_ERR_ARG, "jodo_gradnorm_clip: null argument");
    hipLaunchKernelGGL(k_gradnorm_clip, dim3(1), dim3(64), 0, (hipStream_t)stream, norm_dev, state_dev, max_grad, coef_dev, allowed_dev);
    return jodo_check_launch("k_gradnorm_clip");
}

}
