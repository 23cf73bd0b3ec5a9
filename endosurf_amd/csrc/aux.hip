// The reference renderer's two auxiliary methods as STAND-ALONE calls (ABI v8): what the reference trainer's own loop calls between
// ``renderer(rays)`` and ``loss.backward()`` (src/trainer/trainer_endosurf.py:139-143, :155):
//   errorondepth            (src/renderer/endosurf.py:289-317): points o + d_z * d_gt, then the masked |sdf| and relu(cos) means
//   surface_neighbour_error (src/renderer/endosurf.py:319-342): surface points + random neighbours, then the masked mean of |n - n'|
// Each piece is ONE launch where the reference (and this package until round 5) issued a chain of element-wise / reduction framework
// kernels: ~80 launches per training step, forward and autograd backward together.  The fused training step has its own single-launch
// form of the same arithmetic (loss.hip k_train_loss, rays.hip k_train_aux_points); the formulas here are the same ones.
#include <hip/hip_runtime.h>

#include "launch.h"

namespace es {

__device__ __forceinline__ float sgn1(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// ---- points ---------------------------------------------------------------------------------------------------------------------------
// errorondepth points (endosurf.py:297-300): x = o + d / (d.z + 1e-6) * d_gt, t = rays[:, 8]
// (optionally also inside = (|x| < 1) * mask, endosurf.py:306-309: the one output of errorondepth that does not need the networks)
__global__ __launch_bounds__(256) void k_eod_points(const float* __restrict__ rays, const float* __restrict__ depth_gt, const float* __restrict__ mask,
                                                    int N, float* __restrict__ x, float* __restrict__ t, float* __restrict__ inside) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float* r = rays + 9 * (size_t)i;
    const float inv = r[5] + 1e-6f, dg = depth_gt[i];
    float p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { p[c] = r[c] + (r[3 + c] / inv) * dg; x[3 * (size_t)i + c] = p[c]; }
    t[i] = r[8];
    if (inside != nullptr) inside[i] = (sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) < 1.f ? 1.f : 0.f) * mask[i];
}
// surface points and their neighbours (endosurf.py:323-332): rows [0,N) o + d_z * d_i (d_i = 0 where the ray has no valid hit), rows
// [N,2N) the same + (u - 0.5) * rad; valid = isfinite(d_i) && d_i != 0 && mask == 1
__global__ __launch_bounds__(256) void k_sn_points(const float* __restrict__ rays, const float* __restrict__ mask, const float* __restrict__ d_i,
                                                   const float* __restrict__ u, float rad, int N, float* __restrict__ x, float* __restrict__ t,
                                                   unsigned char* __restrict__ valid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float* r = rays + 9 * (size_t)i;
    const float inv = r[5] + 1e-6f;
    const float di = d_i[i];
    const bool ok = !__builtin_isinf(di) && !__builtin_isnan(di) && di != 0.f && mask[i] == 1.f;
    const float ds = ok ? di : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ps = r[c] + ds * (r[3 + c] / inv);
        x[3 * (size_t)i + c] = ps;
        x[3 * (size_t)(N + i) + c] = ps + (u[3 * (size_t)i + c] - 0.5f) * rad;
    }
    t[i] = r[8]; t[N + i] = r[8];
    valid[i] = ok ? 1 : 0;
}

// ---- block sums of a single 1024-thread workgroup (fixed order: bit-reproducible) ----------------------------------------------------------
template <int K>
__device__ __forceinline__ void block_sums(float (&s)[K], float (&part)[16][K], float (&sums)[K]) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        float v = s[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) part[wv][j] = v;
    }
    __syncthreads();
    if (tid < K) {
        float v = 0.f;
        for (int w = 0; w < 16; ++w) v += part[w][tid];
        sums[tid] = v;
    }
    __syncthreads();
}

// ---- errorondepth's reductions (endosurf.py:302-317) -------------------------------------------------------------------------------------
// inside = (|pts| < 1) * mask; sdf_error = sum |inside * sdf| / (sum inside + 1e-6); angle_error = sum relu(d . g_o) / the same denominator
// (NOT masked, like the reference).  out[3] = {sdf_error, angle_error, denominator}.
__global__ __launch_bounds__(1024) void k_eod_loss(const float* __restrict__ rays, const float* __restrict__ pts, const float* __restrict__ mask,
                                                   const float* __restrict__ sdf, const float* __restrict__ go, int N, float* __restrict__ out,
                                                   float* __restrict__ inside_out) {
    __shared__ float part[16][3];
    __shared__ float sums[3];
    float s[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < N; i += 1024) {
        const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        const float inside = (sqrtf(px * px + py * py + pz * pz) < 1.f ? 1.f : 0.f) * mask[i];
        inside_out[i] = inside;
        s[0] += fabsf(inside * sdf[i]);
        s[1] += inside;
        const float* r = rays + 9 * (size_t)i;
        s[2] += fmaxf(r[3] * go[3 * i] + r[4] * go[3 * i + 1] + r[5] * go[3 * i + 2], 0.f);
    }
    block_sums<3>(s, part, sums);
    if (threadIdx.x == 0) {
        const float den = sums[1] + 1e-6f;
        out[0] = sums[0] / den; out[1] = sums[2] / den; out[2] = den;
    }
}
// adjoints for incoming g[2] = {d sdf_error, d angle_error} (device scalars; a null g means 0)
__global__ __launch_bounds__(256) void k_eod_loss_bwd(const float* __restrict__ rays, const float* __restrict__ inside, const float* __restrict__ sdf,
                                                      const float* __restrict__ go, const float* __restrict__ out, const float* __restrict__ g_sdf_err,
                                                      const float* __restrict__ g_ang_err, int N, float* __restrict__ d_sdf, float* __restrict__ d_go) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float den = out[2];
    const float gs = g_sdf_err ? g_sdf_err[0] : 0.f, ga = g_ang_err ? g_ang_err[0] : 0.f;
    const float in = inside[i];
    d_sdf[i] = gs * sgn1(in * sdf[i]) * in / den;
    const float* r = rays + 9 * (size_t)i;
    const float cs = r[3] * go[3 * i] + r[4] * go[3 * i + 1] + r[5] * go[3 * i + 2];
    const float k = cs > 0.f ? ga / den : 0.f;
    d_go[3 * i] = k * r[3]; d_go[3 * i + 1] = k * r[4]; d_go[3 * i + 2] = k * r[5];
}

// ---- surface_neighbour_error's reduction (endosurf.py:334-339) ---------------------------------------------------------------------------
// n = g / (|g| + 1e-10) for the surface point (row i) and its neighbour (row N + i); loss = sum_valid |n - n'| / max(3 n_valid, 1).
// out[2] = {loss, denominator}.
__global__ __launch_bounds__(1024) void k_sn_loss(const float* __restrict__ g, const unsigned char* __restrict__ valid, int N, float* __restrict__ out) {
    __shared__ float part[16][2];
    __shared__ float sums[2];
    float s[2] = {0.f, 0.f};
    for (int i = threadIdx.x; i < N; i += 1024) {
        if (!valid[i]) continue;
        const float* g1 = g + 3 * (size_t)i;
        const float* g2 = g + 3 * (size_t)(N + i);
        const float n1 = sqrtf(g1[0] * g1[0] + g1[1] * g1[1] + g1[2] * g1[2]) + 1e-10f;
        const float n2 = sqrtf(g2[0] * g2[0] + g2[1] * g2[1] + g2[2] * g2[2]) + 1e-10f;
#pragma unroll
        for (int k = 0; k < 3; ++k) s[0] += fabsf(g1[k] / n1 - g2[k] / n2);
        s[1] += 1.f;
    }
    block_sums<2>(s, part, sums);
    if (threadIdx.x == 0) {
        const float den = fmaxf(3.f * sums[1], 1.f);
        out[0] = sums[0] / den; out[1] = den;
    }
}
__global__ __launch_bounds__(256) void k_sn_loss_bwd(const float* __restrict__ g, const unsigned char* __restrict__ valid, const float* __restrict__ out,
                                                     const float* __restrict__ g_loss, int N, float* __restrict__ d_g) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const size_t i1 = (size_t)i, i2 = (size_t)(N + i);
    float o1[3] = {0.f, 0.f, 0.f}, o2[3] = {0.f, 0.f, 0.f};
    if (valid[i]) {
        const float scale = g_loss[0] / out[1];
        const float* g1 = g + 3 * i1;
        const float* g2 = g + 3 * i2;
        const float r1 = sqrtf(g1[0] * g1[0] + g1[1] * g1[1] + g1[2] * g1[2]), r2 = sqrtf(g2[0] * g2[0] + g2[1] * g2[1] + g2[2] * g2[2]);
        const float d1 = r1 + 1e-10f, d2 = r2 + 1e-10f;
        float n1[3], n2[3], nb[3];
        float dot1 = 0.f, dot2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            n1[k] = g1[k] / d1; n2[k] = g2[k] / d2;
            nb[k] = scale * sgn1(n1[k] - n2[k]);            // adjoint of n1 (and minus the adjoint of n2)
            dot1 += n1[k] * nb[k]; dot2 += n2[k] * nb[k];
        }
        // n = g / d, d = r + eps: gbar = (nbar - n (n . nbar) d / r) / d; torch's norm backward gives 0 at r = 0 (loss.hip, same formula)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o1[k] = r1 > 0.f ? (nb[k] - n1[k] * dot1 * (d1 / r1)) / d1 : nb[k] / d1;
            o2[k] = r2 > 0.f ? -(nb[k] - n2[k] * dot2 * (d2 / r2)) / d2 : -nb[k] / d2;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { d_g[3 * i1 + k] = o1[k]; d_g[3 * i2 + k] = o2[k]; }
}

// ---- two copies in one launch (the (sdf, g_o) rows of an evaluation out of / the adjoints into a shared workspace) --------------------------
__global__ __launch_bounds__(256) void k_copy2(float* __restrict__ da, const float* __restrict__ sa, long long na, float* __restrict__ db,
                                               const float* __restrict__ sb, long long nb) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < na) da[i] = sa[i];
    if (i < nb) db[i] = sb[i];
}

static inline dim3 g256(long long n) { return dim3((unsigned)((n + 255) / 256)); }

int eod_points(const float* rays, const float* depth_gt, const float* mask, int N, float* x, float* t, float* inside, hipStream_t st) {
    if (N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_eod_points, g256(N), dim3(256), 0, st, rays, depth_gt, mask, N, x, t, inside);
    return hip_last("eod_points");
}
int sn_points(const float* rays, const float* mask, const float* d_i, const float* u, float rad, int N, float* x, float* t, unsigned char* valid,
              hipStream_t st) {
    if (N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_sn_points, g256(N), dim3(256), 0, st, rays, mask, d_i, u, rad, N, x, t, valid);
    return hip_last("sn_points");
}
int eod_loss(const float* rays, const float* pts, const float* mask, const float* sdf, const float* go, int N, float* out, float* inside, hipStream_t st) {
    hipLaunchKernelGGL(k_eod_loss, dim3(1), dim3(1024), 0, st, rays, pts, mask, sdf, go, N, out, inside);
    return hip_last("eod_loss");
}
int eod_loss_bwd(const float* rays, const float* inside, const float* sdf, const float* go, const float* out, const float* g_sdf_err,
                 const float* g_ang_err, int N, float* d_sdf, float* d_go, hipStream_t st) {
    if (N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_eod_loss_bwd, g256(N), dim3(256), 0, st, rays, inside, sdf, go, out, g_sdf_err, g_ang_err, N, d_sdf, d_go);
    return hip_last("eod_loss_bwd");
}
int sn_loss(const float* g, const unsigned char* valid, int N, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_sn_loss, dim3(1), dim3(1024), 0, st, g, valid, N, out);
    return hip_last("sn_loss");
}
int sn_loss_bwd(const float* g, const unsigned char* valid, const float* out, const float* g_loss, int N, float* d_g, hipStream_t st) {
    if (N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_sn_loss_bwd, g256(N), dim3(256), 0, st, g, valid, out, g_loss, N, d_g);
    return hip_last("sn_loss_bwd");
}
int copy2(float* da, const float* sa, long long na, float* db, const float* sb, long long nb, hipStream_t st) {
    const long long n = na > nb ? na : nb;
    if (n <= 0) return ST_OK;
    hipLaunchKernelGGL(k_copy2, g256(n), dim3(256), 0, st, da, sa, na, db, sb, nb);
    return hip_last("copy2");
}

}  // namespace es
