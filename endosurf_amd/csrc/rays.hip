// Per-ray kernels: one wavefront (64 lanes) per ray, lanes = samples; scans/reductions are wave-level
// (DPP/__shfl), no LDS traffic except small per-wave staging arrays.
//   ray_setup      get_sphere_intersection (utils.py:194-210) + coarse z sampling (endosurf.py:71-82)
//   upsample_step  up_sample (endosurf.py:221-266) + sample_pdf(det=True) (utils.py:160-191) + the sort of
//                  cat_z_vals (endosurf.py:268-273), emitted as a merge permutation
//   merge_sdf      the gather of cat_z_vals (endosurf.py:282-285)
//   mid_z          section mid-points of render_core (endosurf.py:148-150)
//   composite_fwd  NeuS alpha compositing of render_core (endosurf.py:168-203)
//   composite_bwd  its analytic backward
//   march_find / secant_update  ray_marching + secant (endosurf.py:344-449), fixed-shape, no host branches
#include <hip/hip_runtime.h>

#include "launch.h"
#include "ray_args.h"

namespace es {

constexpr int RAY_NMAX = 256;    // max samples per ray handled by the per-ray kernels
constexpr int WAVES_PER_BLOCK = 4;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// inclusive scans across the 64 lanes
__device__ __forceinline__ float wscan_add(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    return v;
}
__device__ __forceinline__ float wscan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o, 64); if (lane >= o) v *= t; }
    return v;
}
__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.f / (1.f + expf(-x)); }

// torch.linspace(start, end, steps)[i] in fp32 (symmetric evaluation used by ATen's CPU kernel)
__device__ __forceinline__ float linspace_f32(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

struct RayGeom { float o[3], d[3], dz[3], t; };
__device__ __forceinline__ RayGeom load_ray(const float* __restrict__ rays, int ray) {
    RayGeom g;
    const float* r = rays + 9 * (size_t)ray;
    g.o[0] = r[0]; g.o[1] = r[1]; g.o[2] = r[2];
    g.d[0] = r[3]; g.d[1] = r[4]; g.d[2] = r[5];
    const float inv = r[5] + 1e-6f;                                   // rays_d / (rays_d.z + 1e-6)  endosurf.py:66
    g.dz[0] = r[3] / inv; g.dz[1] = r[4] / inv; g.dz[2] = r[5] / inv;
    g.t = r[8];
    return g;
}
__device__ __forceinline__ void sphere_near_far(const RayGeom& g, float& near, float& far) {
    const float dd = g.d[0] * g.d[0] + g.d[1] * g.d[1] + g.d[2] * g.d[2];
    const float d1 = -(g.d[0] * g.o[0] + g.d[1] * g.o[1] + g.d[2] * g.o[2]) / dd;
    const float p0 = g.o[0] + d1 * g.d[0], p1 = g.o[1] + d1 * g.d[1], p2 = g.o[2] + d1 * g.d[2];
    const float tmp = 1.f - (p0 * p0 + p1 * p1 + p2 * p2);
    const float d2 = sqrtf(fmaxf(tmp, 0.f)) / sqrtf(dd);
    near = fmaxf(d1 - d2, 0.f);
    far = d1 + d2;
}
__device__ __forceinline__ float pt_norm(const RayGeom& g, float z) {
    const float x = g.o[0] + g.dz[0] * z, y = g.o[1] + g.dz[1] * z, w = g.o[2] + g.dz[2] * z;
    return sqrtf(x * x + y * y + w * w);
}

// ---------------------------------------------------------------------------------------------------------
// z[ray][s] = near + (far-near) * linspace(0,1,n)[s] (+ (u-0.5) * 2/n_samples if u given); near/far optional outputs.
// ``lin_mode`` 0: render sampling (endosurf.py:78-82); 1: ray-marching proposals near*(1-t)+far*t (endosurf.py:359-360)
__global__ __launch_bounds__(256) void k_ray_setup(const float* __restrict__ rays, const float* __restrict__ u, int N, int n,
                                                   float sample_dist, int lin_mode, float* __restrict__ z, int ldz,
                                                   float* __restrict__ near_out, float* __restrict__ far_out) {
    const int ray = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ray >= N) return;
    const RayGeom g = load_ray(rays, ray);
    float near, far;
    sphere_near_far(g, near, far);
    const float shift = u ? (u[ray] - 0.5f) * sample_dist : 0.f;
    for (int s = lane; s < n; s += 64) {
        const float tv = linspace_f32(0.f, 1.f, n, s);
        float v = lin_mode == 0 ? near + (far - near) * tv : near * (1.f - tv) + far * tv;
        if (u) v = v + shift;
        z[(size_t)ray * ldz + s] = v;
    }
    if (lane == 0) { if (near_out) near_out[ray] = near; if (far_out) far_out[ray] = far; }
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_upsample_step(const float* __restrict__ rays, const float* __restrict__ z_in, int ld_in,
                                                       const float* __restrict__ sdf_in, int ld_sdf, int N, int n, int n_imp,
                                                       float inv_s, float* __restrict__ z_new, float* __restrict__ z_out,
                                                       int ld_out, int* __restrict__ src_idx) {
    __shared__ float s_z[WAVES_PER_BLOCK][RAY_NMAX], s_a[WAVES_PER_BLOCK][RAY_NMAX], s_c[WAVES_PER_BLOCK][RAY_NMAX + 1],
        s_new[WAVES_PER_BLOCK][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ray = blockIdx.x * WAVES_PER_BLOCK + wv;
    if (ray >= N) return;                                    // wave-uniform exit; no block-level barriers below
    float* zs = s_z[wv]; float* av = s_a[wv]; float* cdf0 = s_c[wv]; float* nz = s_new[wv];
    const RayGeom g = load_ray(rays, ray);
    const int nsec = n - 1;

    for (int e = lane; e < n; e += 64) { zs[e] = z_in[(size_t)ray * ld_in + e]; av[e] = sdf_in[(size_t)ray * ld_sdf + e]; }
    __builtin_amdgcn_wave_barrier();
    // section cos, stored after reading sdf (av currently holds sdf; cos goes to cdf0 temporarily)
    for (int e = lane; e < nsec; e += 64) cdf0[e] = (av[e + 1] - av[e]) / (zs[e + 1] - zs[e] + 1e-6f);
    __builtin_amdgcn_wave_barrier();
    // alpha per section and 1 - alpha + 1e-7
    float carry = 1.f, wsum_acc = 0.f;
    const int nchunk = (nsec + 63) / 64;
    float wloc[RAY_NMAX / 64];
#pragma unroll
    for (int c = 0; c < RAY_NMAX / 64; ++c) {
        wloc[c] = 0.f;
        if (c < nchunk) {
            const int e = c * 64 + lane;
            float alpha = 0.f, om = 1.f;
            if (e < nsec) {
                const float cosv = cdf0[e], prev = e == 0 ? 0.f : cdf0[e - 1];
                const bool inside = pt_norm(g, zs[e]) < 1.f || pt_norm(g, zs[e + 1]) < 1.f;
                float cv = fminf(fmaxf(fminf(prev, cosv), -1e3f), 0.f);
                cv = inside ? cv : cv * 0.f;
                const float dist = zs[e + 1] - zs[e];
                const float mid = (av[e] + av[e + 1]) * 0.5f;
                const float pe = mid - cv * dist * 0.5f, ne = mid + cv * dist * 0.5f;
                const float pc = sigmoidf_acc(pe * inv_s), nc = sigmoidf_acc(ne * inv_s);
                alpha = (pc - nc + 1e-6f) / (pc + 1e-6f);
                om = 1.f - alpha + 1e-7f;
            }
            const float incl = wscan_mul(om, lane);                 // inclusive product within the chunk
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.f;
            const float T = carry * excl;
            carry = carry * __shfl(incl, 63, 64);
            const float w = e < nsec ? alpha * T + 1e-5f : 0.f;     // weights + 1e-5 (utils.py:164)
            wloc[c] = w;
            wsum_acc += wsum(w);
        }
    }
    __builtin_amdgcn_wave_barrier();
    // cdf with leading zero
    float run = 0.f;
#pragma unroll
    for (int c = 0; c < RAY_NMAX / 64; ++c) {
        if (c < nchunk) {
            const int e = c * 64 + lane;
            const float pdf = wloc[c] / wsum_acc;
            const float incl = wscan_add(pdf, lane) + run;
            run = __shfl(incl, 63, 64);
            if (e < nsec) cdf0[e + 1] = incl;
        }
    }
    if (lane == 0) cdf0[0] = 0.f;
    __builtin_amdgcn_wave_barrier();
    // inverse-CDF sampling (deterministic u), n_imp <= 64
    if (lane < n_imp) {
        const float uu = linspace_f32(0.5f / (float)n_imp, 1.f - 0.5f / (float)n_imp, n_imp, lane);
        int lo = 0, hi = n;                                          // first index with cdf0[idx] > u  (searchsorted right=True)
        while (lo < hi) { const int m = (lo + hi) >> 1; if (cdf0[m] <= uu) lo = m + 1; else hi = m; }
        const int below = max(lo - 1, 0), above = min(lo, n - 1);
        float denom = cdf0[above] - cdf0[below];
        if (denom < 1e-5f) denom = 1.f;
        const float tt = (uu - cdf0[below]) / denom;
        const float v = zs[below] + tt * (zs[above] - zs[below]);
        nz[lane] = v;
        z_new[(size_t)ray * n_imp + lane] = v;
    }
    __builtin_amdgcn_wave_barrier();
    // stable merge (old before new on ties): rank of old e = e + #{new < z_e}; rank of new j = j' + #{old <= new_j}
    for (int e = lane; e < n; e += 64) {
        const float v = zs[e];
        int cnt = 0;
        for (int j = 0; j < n_imp; ++j) cnt += nz[j] < v ? 1 : 0;
        const int r = e + cnt;
        z_out[(size_t)ray * ld_out + r] = v;
        src_idx[(size_t)ray * ld_out + r] = e;
    }
    if (lane < n_imp) {
        const float v = nz[lane];
        int lo = 0, hi = n;                                          // #{old <= v}
        while (lo < hi) { const int m = (lo + hi) >> 1; if (zs[m] <= v) lo = m + 1; else hi = m; }
        int before = 0;
        for (int j = 0; j < n_imp; ++j) before += (nz[j] < v || (nz[j] == v && j < lane)) ? 1 : 0;
        const int r = lo + before;
        z_out[(size_t)ray * ld_out + r] = v;
        src_idx[(size_t)ray * ld_out + r] = n + lane;
    }
}

__global__ __launch_bounds__(256) void k_merge_sdf(const float* __restrict__ sdf_in, int ld_in, const float* __restrict__ sdf_new,
                                                   int n_imp, const int* __restrict__ src_idx, int ld_out, int N, int n,
                                                   float* __restrict__ sdf_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int tot = n + n_imp;
    if (i >= N * tot) return;
    const int ray = i / tot, r = i - ray * tot;
    const int s = src_idx[(size_t)ray * ld_out + r];
    sdf_out[(size_t)ray * ld_out + r] = s < n ? sdf_in[(size_t)ray * ld_in + s] : sdf_new[(size_t)ray * n_imp + (s - n)];
}

__global__ __launch_bounds__(256) void k_mid_z(const float* __restrict__ z, int ldz, int N, int S, float sample_dist,
                                               float* __restrict__ mid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * S) return;
    const int ray = i / S, s = i - ray * S;
    const float z0 = z[(size_t)ray * ldz + s];
    const float dist = s + 1 < S ? z[(size_t)ray * ldz + s + 1] - z0 : sample_dist;
    mid[(size_t)ray * S + s] = z0 + dist * 0.5f;
}

// ---------------------------------------------------------------------------------------------------------
// NeuS compositing. Per-sample inputs are row-major [N*S] (sdf), [N*S][3] (g_o, rgb).

__device__ __forceinline__ float inv_s_from_variance(float var) { return fminf(fmaxf(expf(var * 10.f), 1e-6f), 1e6f); }

template <bool BWD>
__global__ __launch_bounds__(256) void k_composite(CompositeArgs a) {
    if (BWD && a.n_aux > 0) {      // the auxiliary points' adjoint rows behind the samples' (one strided pass over the whole grid)
        const size_t P = (size_t)a.N * a.S;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n_aux; i += gridDim.x * 256) {
            a.d_sdf[P + i] = a.g_aux_sdf ? a.g_aux_sdf[i] : 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) a.d_go[3 * (P + i) + k] = a.g_aux_go ? a.g_aux_go[3 * (size_t)i + k] : 0.f;
        }
    }
    const int ray = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ray >= a.N) return;
    const RayGeom g = load_ray(a.rays, ray);
    const float inv_s = inv_s_from_variance(a.variance[0]);
    const float r = a.cos_anneal_dev != nullptr ? a.cos_anneal_dev[0] : a.cos_anneal;
    const int S = a.S;
    const int nchunk = (S + 63) / 64;
    constexpr int CH = RAY_NMAX / 64;
    float alpha[CH], Texc[CH], w[CH], mid[CH], dist[CH], pcv[CH], ncv[CH], sdfv[CH], tc[CH], gnorm[CH], relax[CH], ratio[CH];
    float go[CH][3], rgb[CH][3];
    float carry = 1.f;
    float csum[3] = {0.f, 0.f, 0.f}, dsum = 0.f, wmx = -1.f, e_num = 0.f, e_den = 0.f;
    int wmx_i = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (c < nchunk) {
            const int s = c * 64 + lane;
            const bool ok = s < S;
            const size_t p = (size_t)ray * S + (ok ? s : 0);
            const float z0 = a.z[(size_t)ray * a.ldz + (ok ? s : 0)];
            dist[c] = (ok && s + 1 < S) ? a.z[(size_t)ray * a.ldz + s + 1] - z0 : a.sample_dist;
            mid[c] = z0 + dist[c] * 0.5f;
            sdfv[c] = a.sdf[p];
#pragma unroll
            for (int k = 0; k < 3; ++k) { go[c][k] = a.g_o[3 * p + k]; rgb[c][k] = a.rgb[3 * p + k]; }
            tc[c] = g.d[0] * go[c][0] + g.d[1] * go[c][1] + g.d[2] * go[c][2];
            const float ic = -(fmaxf(-tc[c] * 0.5f + 0.5f, 0.f) * (1.f - r) + fmaxf(-tc[c], 0.f) * r);
            const float nx = sdfv[c] + ic * dist[c] * 0.5f, pv = sdfv[c] - ic * dist[c] * 0.5f;
            pcv[c] = sigmoidf_acc(pv * inv_s);
            ncv[c] = sigmoidf_acc(nx * inv_s);
            ratio[c] = (pcv[c] - ncv[c] + 1e-6f) / (pcv[c] + 1e-6f);
            alpha[c] = ok ? fminf(fmaxf(ratio[c], 0.f), 1.f) : 0.f;
            const float om = ok ? 1.f - alpha[c] + 1e-7f : 1.f;
            const float incl = wscan_mul(om, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.f;
            Texc[c] = carry * excl;
            carry = carry * __shfl(incl, 63, 64);
            w[c] = alpha[c] * Texc[c];
            gnorm[c] = sqrtf(go[c][0] * go[c][0] + go[c][1] * go[c][1] + go[c][2] * go[c][2]);
            relax[c] = (ok && pt_norm(g, mid[c]) < 1.2f) ? 1.f : 0.f;
            if (ok) {
                csum[0] += w[c] * rgb[c][0]; csum[1] += w[c] * rgb[c][1]; csum[2] += w[c] * rgb[c][2];
                dsum += w[c] * mid[c];
                e_num += relax[c] * (gnorm[c] - 1.f) * (gnorm[c] - 1.f);
                e_den += relax[c];
                if (w[c] > wmx) { wmx = w[c]; wmx_i = s; }
            }
        }
    }
    // arg-max of the weights across lanes (first index on ties, like torch.max)
    {
        float m = wmx; int mi = wmx_i;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(m, o, 64); const int oi = __shfl_xor(mi, o, 64);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
        wmx = m; wmx_i = mi;
    }
    if (!BWD) {
#pragma unroll
        for (int k = 0; k < 3; ++k) csum[k] = wsum(csum[k]);
        dsum = wsum(dsum); e_num = wsum(e_num); e_den = wsum(e_den);
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (c < nchunk) {
                const int s = c * 64 + lane;
                if (s < S) {
                    a.weights[(size_t)ray * S + s] = w[c]; a.cdf[(size_t)ray * S + s] = pcv[c];
                    if (a.go_copy != nullptr) {
                        const size_t p = (size_t)ray * S + s;
                        a.go_copy[3 * p] = go[c][0]; a.go_copy[3 * p + 1] = go[c][1]; a.go_copy[3 * p + 2] = go[c][2];
                    }
                }
            }
        if (lane == 0) {
            a.color[3 * (size_t)ray + 0] = csum[0]; a.color[3 * (size_t)ray + 1] = csum[1]; a.color[3 * (size_t)ray + 2] = csum[2];
            a.depth[ray] = dsum; a.weight_max[ray] = wmx; a.wmax_idx[ray] = wmx_i;
            if (a.ray_part) { a.ray_part[2 * (size_t)ray] = e_num; a.ray_part[2 * (size_t)ray + 1] = e_den; }
            else { atomicAdd(a.eik_acc + 0, e_num); atomicAdd(a.eik_acc + 1, e_den); }
        }
        return;
    }
    // ---------------- backward ----------------
    const float gc0 = a.g_color[3 * (size_t)ray], gc1 = a.g_color[3 * (size_t)ray + 1], gc2 = a.g_color[3 * (size_t)ray + 2];
    const float gd = a.g_depth[ray];
    const float gwm = a.g_wmax ? a.g_wmax[ray] : 0.f;
    const float geik = a.g_eik[0] / a.eik_den[0];
    // suffix sums of wbar*w over later samples, processed from the last chunk to the first
    float suffix_carry = 0.f, dinvs = 0.f;
#pragma unroll
    for (int c = CH - 1; c >= 0; --c) {
        if (c < nchunk) {
            const int s = c * 64 + lane;
            const bool ok = s < S;
            const size_t p = (size_t)ray * S + (ok ? s : 0);
            float wbar = 0.f;
            if (ok) {
                wbar = (a.g_weights ? a.g_weights[p] : 0.f) + gc0 * rgb[c][0] + gc1 * rgb[c][1] + gc2 * rgb[c][2] + gd * mid[c];
                if (s == wmx_i) wbar += gwm;
            }
            const float ww = ok ? wbar * w[c] : 0.f;
            // inclusive suffix sum within the chunk: reverse scan
            float incl = ww;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_down(incl, o, 64); if (lane + o < 64) incl += t; }
            const float later = incl - ww + suffix_carry;             // sum over s' > s
            suffix_carry += __shfl(incl, 0, 64);
            if (ok) {
                float abar = wbar * Texc[c] - later / (1.f - alpha[c] + 1e-7f);
                if (ratio[c] < 0.f || ratio[c] > 1.f) abar = 0.f;        // clip(0,1) passes gradient only inside
                const float cden = pcv[c] + 1e-6f;
                const float pbar = abar / cden;                          // adj of p = pc - nc
                const float cbar = -abar * (pcv[c] - ncv[c] + 1e-6f) / (cden * cden) + (a.g_cdf ? a.g_cdf[p] : 0.f);
                const float pcbar = pbar + cbar, ncbar = -pbar;
                const float dpv = pcbar * pcv[c] * (1.f - pcv[c]);       // adj of (prev*inv_s)
                const float dnx = ncbar * ncv[c] * (1.f - ncv[c]);
                const float ic = -(fmaxf(-tc[c] * 0.5f + 0.5f, 0.f) * (1.f - r) + fmaxf(-tc[c], 0.f) * r);
                const float nx = sdfv[c] + ic * dist[c] * 0.5f, pv = sdfv[c] - ic * dist[c] * 0.5f;
                dinvs += dpv * pv + dnx * nx;
                const float prevbar = dpv * inv_s, nextbar = dnx * inv_s;
                a.d_sdf[p] = prevbar + nextbar;
                const float icbar = (nextbar - prevbar) * dist[c] * 0.5f;
                // ic = -(relu(-tc/2+1/2)(1-r) + relu(-tc) r)
                const float tcbar = icbar * ((-tc[c] * 0.5f + 0.5f > 0.f ? 0.5f * (1.f - r) : 0.f) + (-tc[c] > 0.f ? r : 0.f));
                const float en = relax[c] * geik * 2.f * (gnorm[c] - 1.f);
                const float invn = gnorm[c] > 0.f ? 1.f / gnorm[c] : 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float v = tcbar * g.d[k] + en * go[c][k] * invn;
                    if (a.g_gradients_o) v += a.g_gradients_o[3 * p + k];
                    a.d_go[3 * p + k] = v;
                    a.d_rgb[3 * p + k] = w[c] * (k == 0 ? gc0 : (k == 1 ? gc1 : gc2));
                }
            }
        }
    }
    dinvs = wsum(dinvs);
    if (lane == 0) {
        if (a.ray_part) a.ray_part[2 * (size_t)ray] = dinvs;
        else atomicAdd(a.d_invs_acc, dinvs);
    }
}
// deterministic mode: acc[j] += sum_ray part[ray][j] (j < ncol) in a fixed order: strided per-thread sums, then a fixed tree
__global__ __launch_bounds__(1024) void k_ray_part_reduce(const float* __restrict__ part, int N, int ncol, float* __restrict__ acc) {
    __shared__ float sh[1024];
    for (int j = 0; j < ncol; ++j) {
        float s = 0.f;
        for (int i = threadIdx.x; i < N; i += 1024) s += part[2 * (size_t)i + j];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) acc[j] += sh[0];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Ray marching post-process: first sign change of val = -sdf along the 128 proposals (endosurf.py:379-406) and the
// initial secant estimate (endosurf.py:428). state[ray] = {d_low, f_low, d_high, f_high}; flags: bit0 mask, bit1 mask_0_not_occupied
__global__ __launch_bounds__(256) void k_march_find(const float* __restrict__ sdf, const float* __restrict__ dprop, int N, int n,
                                                    float tau, float* __restrict__ state, int* __restrict__ flags,
                                                    float* __restrict__ d_pred) {
    const int ray = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ray >= N) return;
    const float* v = sdf + (size_t)ray * n;
    int first = 0x7fffffff;
    for (int s = lane; s + 1 < n; s += 64) {
        const float a0 = -(v[s] - tau), a1 = -(v[s + 1] - tau);
        if (a0 * a1 < 0.f) first = min(first, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
    if (lane == 0) {
        const float v0 = -(v[0] - tau);
        const bool m0 = v0 < 0.f;
        bool mask = false;
        float dl = 0.f, fl = -1.f, dh = 1.f, fh = 1.f;
        if (first != 0x7fffffff) {
            const int i2 = min(first + 1, n - 1);
            fl = -(v[first] - tau); fh = -(v[i2] - tau);
            dl = dprop[(size_t)ray * n + first]; dh = dprop[(size_t)ray * n + i2];
            mask = (fl < 0.f) && m0;
        }
        state[4 * (size_t)ray + 0] = dl; state[4 * (size_t)ray + 1] = fl; state[4 * (size_t)ray + 2] = dh; state[4 * (size_t)ray + 3] = fh;
        flags[ray] = (mask ? 1 : 0) | (m0 ? 2 : 0);
        d_pred[ray] = -fl * (dh - dl) / (fh - fl) + dl;
    }
}
// Block-wise marching: done[ray] = 1 once the outcome of k_march_find is decided by the first n_valid proposals (first point
// occupied => the ray is masked out; or a sign change among them => later proposals cannot matter, endosurf.py:383-392).
__global__ __launch_bounds__(256) void k_march_progress(const float* __restrict__ sdf, int N, int n, int n_valid, float tau,
                                                        int* __restrict__ done) {
    const int ray = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ray >= N) return;
    const float* v = sdf + (size_t)ray * n;
    int found = 0;
    for (int s = lane; s + 1 < n_valid; s += 64) {
        const float a0 = -(v[s] - tau), a1 = -(v[s + 1] - tau);
        if (a0 * a1 < 0.f) found = 1;
    }
    if (lane == 0 && !(-(v[0] - tau) < 0.f)) found = 1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) found |= __shfl_xor(found, o, 64);
    if (lane == 0) done[ray] = found;
}
// points of a secant iteration: p = o + d_pred * d / d.z   (no epsilon: reference endosurf.py:427)
__global__ __launch_bounds__(256) void k_secant_points(const float* __restrict__ rays, const float* __restrict__ d_pred, int N,
                                                       float* __restrict__ x, float* __restrict__ t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float* r = rays + 9 * (size_t)i;
    const float dp = d_pred[i];
    x[3 * (size_t)i + 0] = r[0] + dp * (r[3] / r[5]);
    x[3 * (size_t)i + 1] = r[1] + dp * (r[4] / r[5]);
    x[3 * (size_t)i + 2] = r[2] + dp * (r[5] / r[5]);
    t[i] = r[8];
}
// one secant update (endosurf.py:438-448); f_mid = sdf - tau with the reference's sign convention kept as is
__global__ __launch_bounds__(256) void k_secant_update(const float* __restrict__ sdf_mid, int N, float tau, float* __restrict__ state,
                                                       float* __restrict__ d_pred) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float dl = state[4 * (size_t)i], fl = state[4 * (size_t)i + 1], dh = state[4 * (size_t)i + 2], fh = state[4 * (size_t)i + 3];
    const float fm = sdf_mid[i] - tau, dp = d_pred[i];
    if (fm < 0.f) { dl = dp; fl = fm; } else { dh = dp; fh = fm; }
    state[4 * (size_t)i] = dl; state[4 * (size_t)i + 1] = fl; state[4 * (size_t)i + 2] = dh; state[4 * (size_t)i + 3] = fh;
    d_pred[i] = -fl * (dh - dl) / (fh - fl) + dl;
}
// d_pred_out: masked -> d_pred, not masked -> inf, first point occupied -> 0 (endosurf.py:416-420)
__global__ __launch_bounds__(256) void k_march_finish(const float* __restrict__ d_pred, const int* __restrict__ flags, int N,
                                                      float* __restrict__ d_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int f = flags[i];
    float v = (f & 1) ? d_pred[i] : __builtin_inff();
    if (!(f & 2)) v = 0.f;
    d_out[i] = v;
}

// The 3N auxiliary points of a training step in one launch (instead of ~28 element-wise torch kernels):
//   rows [0,N):   errorondepth points  o + d_z * depth_gt                          (endosurf.py:297-300)
//   rows [N,2N):  surface points       o + d_z * d_i  (d_i = 0 where the ray has no valid hit)   (endosurf.py:325-329)
//   rows [2N,3N): their neighbours     surface + (u - 0.5) * rad                    (endosurf.py:331-332)
// valid[i] = isfinite(d_i) && d_i != 0 && mask == 1  (endosurf.py:323);  t = rays[:, 8] for every block of rows.
__global__ __launch_bounds__(256) void k_train_aux_points(const float* __restrict__ rays, const float* __restrict__ depth_gt,
                                                          const float* __restrict__ mask, const float* __restrict__ d_i,
                                                          const float* __restrict__ u, float rad, int N, float* __restrict__ x,
                                                          float* __restrict__ t, unsigned char* __restrict__ valid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float* r = rays + 9 * (size_t)i;
    const float inv = r[5] + 1e-6f;
    const float dz[3] = {r[3] / inv, r[4] / inv, r[5] / inv};
    const float di = d_i[i];
    const bool ok = !__builtin_isinf(di) && !__builtin_isnan(di) && di != 0.f && mask[i] == 1.f;
    const float ds = ok ? di : 0.f, dg = depth_gt[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ps = r[c] + ds * dz[c];
        x[3 * (size_t)i + c] = r[c] + dz[c] * dg;
        x[3 * (size_t)(N + i) + c] = ps;
        x[3 * (size_t)(2 * N + i) + c] = ps + (u[3 * (size_t)i + c] - 0.5f) * rad;
    }
    t[i] = r[8]; t[N + i] = r[8]; t[2 * N + i] = r[8];
    valid[i] = ok ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------
static inline dim3 ray_grid(int N) { return dim3((N + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK); }

int ray_setup(const float* rays, const float* u, int N, int n, float sample_dist, int lin_mode, float* z, int ldz, float* near_out,
              float* far_out, hipStream_t st) {
    if (N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_ray_setup, ray_grid(N), dim3(256), 0, st, rays, u, N, n, sample_dist, lin_mode, z, ldz, near_out, far_out);
    return hip_last("ray_setup");
}
int upsample_step(const float* rays, const float* z_in, int ld_in, const float* sdf_in, int ld_sdf, int N, int n, int n_imp,
                  float inv_s, float* z_new, float* z_out, int ld_out, int* src_idx, hipStream_t st) {
    ES_REQUIRE(n >= 2 && n <= RAY_NMAX && n_imp >= 1 && n_imp <= 64 && n + n_imp <= ld_out, "upsample_step sizes");
    if (N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_upsample_step, ray_grid(N), dim3(256), 0, st, rays, z_in, ld_in, sdf_in, ld_sdf, N, n, n_imp, inv_s, z_new,
                       z_out, ld_out, src_idx);
    return hip_last("upsample_step");
}
int merge_sdf(const float* sdf_in, int ld_in, const float* sdf_new, int n_imp, const int* src_idx, int ld_out, int N, int n,
              float* sdf_out, hipStream_t st) {
    if (N <= 0) return ST_OK;
    const int tot = N * (n + n_imp);
    hipLaunchKernelGGL(k_merge_sdf, dim3((tot + 255) / 256), dim3(256), 0, st, sdf_in, ld_in, sdf_new, n_imp, src_idx, ld_out, N, n, sdf_out);
    return hip_last("merge_sdf");
}
int mid_z(const float* z, int ldz, int N, int S, float sample_dist, float* mid, hipStream_t st) {
    if (N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_mid_z, dim3((N * S + 255) / 256), dim3(256), 0, st, z, ldz, N, S, sample_dist, mid);
    return hip_last("mid_z");
}
int composite(const CompositeArgs& a, int backward, hipStream_t st) {
    ES_REQUIRE(a.S >= 1 && a.S <= RAY_NMAX, "composite: samples per ray out of range");
    if (a.N <= 0) return ST_OK;
    if (backward) hipLaunchKernelGGL(k_composite<true>, ray_grid(a.N), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_composite<false>, ray_grid(a.N), dim3(256), 0, st, a);
    if (a.ray_part)
        hipLaunchKernelGGL(k_ray_part_reduce, dim3(1), dim3(1024), 0, st, a.ray_part, a.N, backward ? 1 : 2, backward ? a.d_invs_acc : a.eik_acc);
    return hip_last("composite");
}
int march_find(const float* sdf, const float* dprop, int N, int n, float tau, float* state, int* flags, float* d_pred, hipStream_t st) {
    if (N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_march_find, ray_grid(N), dim3(256), 0, st, sdf, dprop, N, n, tau, state, flags, d_pred);
    return hip_last("march_find");
}
// scalar epilogue of the deviation network (endosurf.py:168, :205, :845-852): s_val = 1 / inv_s and the chain rule
// d var = d inv_s * 10 exp(10 var) inside the clip range (0 outside)
__global__ void k_variance_terms(const float* __restrict__ variance, const float* __restrict__ d_invs_acc, float* __restrict__ s_val,
                                 float* __restrict__ d_var) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float e = expf(variance[0] * 10.f);
    if (s_val) s_val[0] = 1.f / fminf(fmaxf(e, 1e-6f), 1e6f);
    if (d_var) d_var[0] = (e >= 1e-6f && e <= 1e6f) ? d_invs_acc[0] * 10.f * e : 0.f;
}
int variance_terms(const float* variance, const float* d_invs_acc, float* s_val, float* d_var, hipStream_t st) {
    hipLaunchKernelGGL(k_variance_terms, dim3(1), dim3(64), 0, st, variance, d_invs_acc, s_val, d_var);
    return hip_last("variance_terms");
}

int march_progress(const float* sdf, int N, int n, int n_valid, float tau, int* done, hipStream_t st) {
    if (N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_march_progress, ray_grid(N), dim3(256), 0, st, sdf, N, n, n_valid, tau, done);
    return hip_last("march_progress");
}
int secant_points(const float* rays, const float* d_pred, int N, float* x, float* t, hipStream_t st) {
    if (N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_secant_points, dim3((N + 255) / 256), dim3(256), 0, st, rays, d_pred, N, x, t);
    return hip_last("secant_points");
}
int secant_update(const float* sdf_mid, int N, float tau, float* state, float* d_pred, hipStream_t st) {
    if (N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_secant_update, dim3((N + 255) / 256), dim3(256), 0, st, sdf_mid, N, tau, state, d_pred);
    return hip_last("secant_update");
}
int train_aux_points(const float* rays, const float* depth_gt, const float* mask, const float* d_i, const float* u, float rad, int N,
                     float* x, float* t, unsigned char* valid, hipStream_t st) {
    if (N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_train_aux_points, dim3((N + 255) / 256), dim3(256), 0, st, rays, depth_gt, mask, d_i, u, rad, N, x, t, valid);
    return hip_last("train_aux_points");
}
int march_finish(const float* d_pred, const int* flags, int N, float* d_out, hipStream_t st) {
    if (N <= 0) return ST_OK;
    hipLaunchKernelGGL(k_march_finish, dim3((N + 255) / 256), dim3(256), 0, st, d_pred, flags, N, d_out);
    return hip_last("march_finish");
}

}  // namespace es
