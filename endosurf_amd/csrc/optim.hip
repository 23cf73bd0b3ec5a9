// Adam (torch.optim.Adam defaults as used by the reference trainer, trainer_endosurf.py:70: betas (0.9, 0.999), eps 1e-8, no
// weight decay / amsgrad) over the flat parameter buffer in ONE launch: 1.65 M parameters = 46 MB of HBM traffic (~10 us)
// instead of the multi-tensor path's three launches over 82 small tensors (~0.23 ms).
#include <hip/hip_runtime.h>

#include "launch.h"

namespace es {

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long long n, float beta1, float beta2, float eps, float step_size,
                                              float bc2_sqrt, float grad_scale, const float* __restrict__ g_extra, long long extra_index,
                                              const float* __restrict__ scal) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (scal != nullptr) { step_size = scal[0]; bc2_sqrt = scal[1]; grad_scale = scal[2]; }      // device-resident schedule (captured step)
    float gi = g[i];
    if (g_extra != nullptr && i == extra_index) gi += g_extra[0];
    gi *= grad_scale;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;        // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= step_size * (mi / denom);
}

int adam_step(float* p, const float* g, float* m, float* v, long long n, float beta1, float beta2, float eps, float step_size,
              float bc2_sqrt, float grad_scale, const float* g_extra, long long extra_index, hipStream_t st) {
    if (n <= 0) return ST_OK;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n, beta1, beta2, eps, step_size, bc2_sqrt,
                       grad_scale, g_extra, extra_index, (const float*)nullptr);
    return hip_last("adam_step");
}
int adam_step_dev(float* p, const float* g, float* m, float* v, long long n, float beta1, float beta2, float eps, const float* scal,
                  const float* g_extra, long long extra_index, hipStream_t st) {
    if (n <= 0) return ST_OK;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n, beta1, beta2, eps, 0.f, 1.f, 1.f, g_extra,
                       extra_index, scal);
    return hip_last("adam_step_dev");
}

// The per-step scalars of a hipGraph-captured training step, computed ON the device from a device-resident step counter (nothing is
// passed from the host between replays): state[0] = global step, state[1] = Adam step count, both incremented first; then
//   lr        = lr_init * (step < warm_up_end ? step / warm_up_end : (cos(pi (step - warm) / (n_iter - warm)) + 1) / 2 (1 - alpha) + alpha)
//               (update_learning_rate, trainer_endosurf.py:183-203)
//   scal[0..2] = lr / (1 - beta1^t), sqrt(1 - beta2^t), grad_scale          (es_adam_step_dev)
//   scal[3]    = anneal_end == 0 ? 1 : min(1, step / anneal_end)            (get_cos_anneal_ratio, endosurf.py:215-219)
// in double precision like the host code they replace.
__global__ void k_train_schedule(double* state, double lr_init, double n_iter, double warm_up_end, double lr_alpha, double beta1, double beta2,
                                 float grad_scale, double anneal_end, float* scal) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double step = state[0] + 1.0, t = state[1] + 1.0;
    state[0] = step; state[1] = t;
    double f;
    if (step < warm_up_end) f = step / warm_up_end;
    else f = (cos(3.14159265358979323846 * (step - warm_up_end) / (n_iter - warm_up_end)) + 1.0) * 0.5 * (1.0 - lr_alpha) + lr_alpha;
    scal[0] = (float)(lr_init * f / (1.0 - pow(beta1, t)));
    scal[1] = (float)sqrt(1.0 - pow(beta2, t));
    scal[2] = grad_scale;
    scal[3] = anneal_end == 0.0 ? 1.f : (float)fmin(1.0, step / anneal_end);
}
int train_schedule(double* state, double lr_init, double n_iter, double warm_up_end, double lr_alpha, double beta1, double beta2, float grad_scale,
                   double anneal_end, float* scal, hipStream_t st) {
    hipLaunchKernelGGL(k_train_schedule, dim3(1), dim3(64), 0, st, state, lr_init, n_iter, warm_up_end, lr_alpha, beta1, beta2, grad_scale, anneal_end, scal);
    return hip_last("train_schedule");
}

}  // namespace es
