// The training-step loss of the reference (compute_loss, src/trainer/trainer_endosurf.py:133-162, with errorondepth's and
// surface_neighbour_error's reductions, src/renderer/endosurf.py:306-315, :337-339) and its gradient w.r.t. the renderer
// outputs, as ONE single-workgroup launch instead of ~80 tiny element-wise / reduction kernels: pass 1 accumulates the nine
// batch sums, pass 2 (after a block barrier) writes the six loss terms and all adjoints (for d total = 1).
#include <hip/hip_runtime.h>

#include "launch.h"
#include "loss_args.h"

namespace es {

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(1024) void k_train_loss(LossArgs a) {
    __shared__ float part[16][9];
    __shared__ float sums[9];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = a.N;
    // sums: 0 |col err|, 1 cmask, 2 |inside*sdf|, 3 inside, 4 relu(cos), 5 |depth err|, 6 valid*mask, 7 sn diff, 8 n_valid
    float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < N; i += 1024) {
        const float cm = a.cmask[i], m = a.mask[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) s[0] += fabsf((a.color_map[3 * i + k] - a.color_gt[3 * i + k]) * cm);
        s[1] += cm;
        const float px = a.eod_pts[3 * i], py = a.eod_pts[3 * i + 1], pz = a.eod_pts[3 * i + 2];
        const float inside = (sqrtf(px * px + py * py + pz * pz) < 1.f ? 1.f : 0.f) * m;
        s[2] += fabsf(inside * a.aux_sdf[i]);
        s[3] += inside;
        const float* r = a.rays + 9 * (size_t)i;
        const float cs = r[3] * a.aux_go[3 * i] + r[4] * a.aux_go[3 * i + 1] + r[5] * a.aux_go[3 * i + 2];
        s[4] += fmaxf(cs, 0.f);
        const float v = inside * m;
        s[5] += fabsf((a.depth_map[i] - a.depth_gt[i]) * v);
        s[6] += v;
        if (a.valid_sn[i]) {
            const float* g1 = a.aux_go + 3 * (size_t)(N + i);
            const float* g2 = a.aux_go + 3 * (size_t)(2 * N + i);
            const float n1 = sqrtf(g1[0] * g1[0] + g1[1] * g1[1] + g1[2] * g1[2]) + 1e-10f;
            const float n2 = sqrtf(g2[0] * g2[0] + g2[1] * g2[1] + g2[2] * g2[2]) + 1e-10f;
#pragma unroll
            for (int k = 0; k < 3; ++k) s[7] += fabsf(g1[k] / n1 - g2[k] / n2);
            s[8] += 1.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        float v = s[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) part[wv][j] = v;
    }
    __syncthreads();
    if (tid < 9) {
        float v = 0.f;
        for (int w = 0; w < 16; ++w) v += part[w][tid];
        sums[tid] = v;
    }
    __syncthreads();
    if (a.den_out != nullptr) {      // the local normalisers only (exact data-parallel mode, first of two launches)
        if (tid == 0) { a.den_out[0] = sums[1]; a.den_out[1] = sums[3]; a.den_out[2] = sums[6]; a.den_out[3] = sums[8]; }
        return;
    }
    const bool glob = a.den_global != nullptr;
    const float ws = glob ? a.world : 1.f;           // folded into the normalisers: term = world * sum / (global denominator)
    const float den_c = ((glob ? a.den_global[0] : sums[1]) + 1e-10f) / ws, den_i = ((glob ? a.den_global[1] : sums[3]) + 1e-6f) / ws,
                den_d = ((glob ? a.den_global[2] : sums[6]) + 1e-10f) / ws;
    const float den_sn = fmaxf(3.f * (glob ? a.den_global[3] : sums[8]), 1.f) / ws;
    if (tid == 0) {
        const float lc = sums[0] / den_c, ls = sums[2] / den_i, la = sums[4] / den_i, ld = sums[5] / den_d, lsn = sums[7] / den_sn;
        const float le = a.eik[0];                   // (exact mode: the caller has rescaled the eikonal term the same way)
        a.terms[0] = lc; a.terms[1] = ld; a.terms[2] = ls; a.terms[3] = la; a.terms[4] = le; a.terms[5] = lsn;
        a.terms[6] = a.w_color * lc + a.w_depth * ld + a.w_sdf * ls + a.w_angle * la + a.w_eik * le + a.w_sn * lsn;
        a.terms[7] = sums[8];
        if (a.total_out != nullptr) a.total_out[0] = a.terms[6];
        a.g_eik[0] = a.w_eik;
    }
    for (int i = tid; i < N; i += 1024) {
        const float cm = a.cmask[i], m = a.mask[i];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            a.g_color[3 * i + k] = a.w_color * sgn((a.color_map[3 * i + k] - a.color_gt[3 * i + k]) * cm) * cm / den_c;
        const float px = a.eod_pts[3 * i], py = a.eod_pts[3 * i + 1], pz = a.eod_pts[3 * i + 2];
        const float inside = (sqrtf(px * px + py * py + pz * pz) < 1.f ? 1.f : 0.f) * m;
        a.g_aux_sdf[i] = a.w_sdf * sgn(inside * a.aux_sdf[i]) * inside / den_i;
        const float* r = a.rays + 9 * (size_t)i;
        const float cs = r[3] * a.aux_go[3 * i] + r[4] * a.aux_go[3 * i + 1] + r[5] * a.aux_go[3 * i + 2];
        const float ga = cs > 0.f ? a.w_angle / den_i : 0.f;
        a.g_aux_go[3 * i] = ga * r[3]; a.g_aux_go[3 * i + 1] = ga * r[4]; a.g_aux_go[3 * i + 2] = ga * r[5];
        const float v = inside * m;
        a.g_depth[i] = a.w_depth * sgn((a.depth_map[i] - a.depth_gt[i]) * v) * v / den_d;
        // surface-neighbour term: n = g / (|g| + eps) for the surface point (row N+i) and its neighbour (row 2N+i)
        const size_t i1 = (size_t)(N + i), i2 = (size_t)(2 * N + i);
        a.g_aux_sdf[i1] = 0.f; a.g_aux_sdf[i2] = 0.f;
        float o1[3] = {0.f, 0.f, 0.f}, o2[3] = {0.f, 0.f, 0.f};
        if (a.valid_sn[i]) {
            const float* g1 = a.aux_go + 3 * i1;
            const float* g2 = a.aux_go + 3 * i2;
            const float r1 = sqrtf(g1[0] * g1[0] + g1[1] * g1[1] + g1[2] * g1[2]), r2 = sqrtf(g2[0] * g2[0] + g2[1] * g2[1] + g2[2] * g2[2]);
            const float d1 = r1 + 1e-10f, d2 = r2 + 1e-10f;
            float n1[3], n2[3], nb[3];
            float dot1 = 0.f, dot2 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                n1[k] = g1[k] / d1; n2[k] = g2[k] / d2;
                nb[k] = a.w_sn * sgn(n1[k] - n2[k]) / den_sn;        // adjoint of n1 (and minus the adjoint of n2)
                dot1 += n1[k] * nb[k]; dot2 += n2[k] * nb[k];
            }
            // n = g/d, d = r + eps:  dn_k/dg_j = delta_kj / d - g_k g_j / (r d^2)  =>  gbar = (nbar - n (n . nbar) d/r) / d;
            // torch's norm backward gives 0 at r = 0
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o1[k] = r1 > 0.f ? (nb[k] - n1[k] * dot1 * (d1 / r1)) / d1 : nb[k] / d1;
                o2[k] = r2 > 0.f ? -(nb[k] - n2[k] * dot2 * (d2 / r2)) / d2 : -nb[k] / d2;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { a.g_aux_go[3 * i1 + k] = o1[k]; a.g_aux_go[3 * i2 + k] = o2[k]; }
    }
}

int train_loss(const LossArgs& a, hipStream_t st) {
    if (a.N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_train_loss, dim3(1), dim3(1024), 0, st, a);
    return hip_last("train_loss");
}

}  // namespace es
