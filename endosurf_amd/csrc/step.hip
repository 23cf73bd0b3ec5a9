// The small launches that keep a training step inside the library (round 4): what used to be ~19 element-wise PyTorch kernels per
// step (seven zero fills, five clones, two concatenations, an add, a divide, a multi-tensor scale, two uniform draws -- ~0.1 ms of GPU
// time and, more to the point, ~19 of a step's ~170 host launches) is one memset of a step arena, one Philox launch, one epilogue of
// the render forward and (only when a loss is back-propagated with a non-unit seed) one scale.  The reference performs these
// operations as separate ATen kernels (trainer_endosurf.py:106-162 around render_rays; torch.rand at endosurf.py:81 and :331).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "launch.h"

namespace es {

// ---- uniform draws: Philox4x32-10 (Salmon et al., SC'11), counter = (draw index / 4, 0, subsequence, 0), key = seed ------------------
// u = (x >> 8) * 2^-24 in [0, 1): 24 random mantissa bits, the resolution torch.rand has for float32.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
__global__ __launch_bounds__(256) void k_uniform(float* __restrict__ out, long long n, unsigned long long seed, unsigned long long subseq,
                                                 const double* __restrict__ subseq_dev) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;          // quad of four draws
    if (4 * q >= n) return;
    if (subseq_dev != nullptr) subseq += (unsigned long long)subseq_dev[0];      // the device-resident step counter of a captured step
    uint32_t c[4] = {(uint32_t)q, (uint32_t)((unsigned long long)q >> 32), (uint32_t)subseq, (uint32_t)(subseq >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * q + j < n) out[4 * q + j] = (float)(c[j] >> 8) * (1.f / 16777216.f);
}
int uniform(float* out, long long n, unsigned long long seed, unsigned long long subseq, const double* subseq_dev, hipStream_t st) {
    if (n <= 0) return ST_OK;
    const long long quads = (n + 3) / 4;
    hipLaunchKernelGGL(k_uniform, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, out, n, seed, subseq, subseq_dev);
    return hip_last("uniform");
}

// ---- out = in * s[0] ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scale(float* __restrict__ out, const float* __restrict__ in, long long n, const float* __restrict__ s) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i] * s[0];
}
int scale(float* out, const float* in, long long n, const float* s, hipStream_t st) {
    if (n <= 0) return ST_OK;
    hipLaunchKernelGGL(k_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, in, n, s);
    return hip_last("scale");
}

int zero(void* p, long long nbytes, hipStream_t st) {
    if (nbytes <= 0) return ST_OK;
    if (hipMemsetAsync(p, 0, (size_t)nbytes, st) != hipSuccess) return hip_last("zero");
    return ST_OK;
}

// ---- epilogue of the render forward ---------------------------------------------------------------------------------------------------
// gradient_o_error = sum(relax * err) / (sum(relax) + 1e-6) (endosurf.py:187-190) from the two batch sums of es_composite_forward, the
// normaliser itself (the backward and the exact data-parallel mode need it), and own-storage copies of the auxiliary points' (sdf, g_o)
// rows of the point workspace (so that the step's outputs do not keep the workspace alive).
__global__ __launch_bounds__(256) void k_render_finish(const float* __restrict__ eik_acc, const float* __restrict__ aux_sdf_ws,
                                                       const float* __restrict__ aux_go_ws, int n_aux, float* __restrict__ eik,
                                                       float* __restrict__ eik_den, float* __restrict__ aux_sdf, float* __restrict__ aux_go) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        const float den = eik_acc[1] + 1e-6f;
        eik_den[0] = den;
        eik_den[1] = den;                                   // second copy: the differentiable render output and the saved normaliser
        eik[0] = eik_acc[0] / den;
    }
    if (i < n_aux) aux_sdf[i] = aux_sdf_ws[i];
    if (i < 3 * n_aux) aux_go[i] = aux_go_ws[i];
}
int render_finish(const float* eik_acc, const float* aux_sdf_ws, const float* aux_go_ws, int n_aux, float* eik, float* eik_den, float* aux_sdf,
                  float* aux_go, hipStream_t st) {
    const int n = 3 * n_aux > 1 ? 3 * n_aux : 1;
    hipLaunchKernelGGL(k_render_finish, dim3((n + 255) / 256), dim3(256), 0, st, eik_acc, aux_sdf_ws, aux_go_ws, n_aux, eik, eik_den, aux_sdf, aux_go);
    return hip_last("render_finish");
}

}  // namespace es
