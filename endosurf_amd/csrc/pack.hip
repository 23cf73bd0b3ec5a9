// weight-norm (w = g * v / ||v||_row, reference utils.py:57-58,108-109) + packing of the effective weights
// into MFMA B-fragment order, once per optimizer step instead of once per nn.Linear call (>=117x/step in
// the reference); and the matching backward dW_eff -> (dg, dv, db).
#include <hip/hip_runtime.h>

#include "arch.h"
#include "launch.h"

namespace es {

struct SegDev {
    int net, layer, dir, row0, col0, kreal, nreal, skip_scale, kg, nt;
    unsigned long long off4;
};
__constant__ SegDev c_segs[SEG_COUNT];
__constant__ int c_layer_k[NETS * LAYERS];
__constant__ int c_layer_n[NETS * LAYERS];
__constant__ int c_param_off[NETS * LAYERS];
__constant__ int c_weff_off[NETS * LAYERS];
__constant__ int c_p16_segs[P16_COUNT];
#define P16_SEGS_DEV(i) c_p16_segs[i]

static DeviceOnce g_tables_ready;      // __constant__ tables live per device: upload once on each device that is used

int init_tables() {
    if (!g_tables_ready.first()) return 0;
    SegDev h[SEG_COUNT];
    size_t off = 0;
    for (int i = 0; i < SEG_COUNT; ++i) {
        const SegDesc& s = SEGS[i];
        h[i] = {s.net, s.layer, s.dir, s.row0, s.col0, s.kreal, s.nreal, s.skip_scale, seg_kg(s), seg_nt(s), (unsigned long long)off};
        off += (size_t)seg_kg(s) * seg_nt(s) * 64;
    }
    int lk[NETS * LAYERS], ln[NETS * LAYERS], po[NETS * LAYERS], wo[NETS * LAYERS];
    int p = 0, w = 0;
    for (int n = 0; n < NETS; ++n)
        for (int l = 0; l < LAYERS; ++l) {
            lk[n * LAYERS + l] = LAYER_K[n][l];
            ln[n * LAYERS + l] = LAYER_N[n][l];
            po[n * LAYERS + l] = p;
            wo[n * LAYERS + l] = w;
            p += LAYER_N[n][l] * (2 + LAYER_K[n][l]);
            w += LAYER_N[n][l] * (1 + LAYER_K[n][l]);
        }
    ES_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_segs), h, sizeof(h)));
    ES_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_layer_k), lk, sizeof(lk)));
    ES_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_layer_n), ln, sizeof(ln)));
    ES_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_param_off), po, sizeof(po)));
    ES_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_weff_off), wo, sizeof(wo)));
    ES_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_p16_segs), P16_SEGS, sizeof(P16_SEGS)));
    g_tables_ready.done();
    return 0;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one wavefront per weight row: Weff[n][:] = g[n] * v[n][:] / ||v[n]||, bias copied
__global__ __launch_bounds__(256) void k_weff(const float* __restrict__ params, float* __restrict__ weff, int first_layer) {
    const int li = first_layer + blockIdx.y;
    const int N = c_layer_n[li], K = c_layer_k[li];
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const float* pb = params + c_param_off[li];
    const float* pg = pb + N;
    const float* pv = pg + N + (size_t)n * K;
    float ss = 0.f;
    for (int k = lane; k < K; k += 64) { const float v = pv[k]; ss = fmaf(v, v, ss); }
    ss = wave_sum(ss);
    const float scale = pg[n] / sqrtf(ss);
    float* w = weff + c_weff_off[li];
    for (int k = lane; k < K; k += 64) w[(size_t)n * K + k] = scale * pv[k];
    if (lane == 0) w[(size_t)N * K + n] = pb[n];
}

// one thread per packed float4
__global__ __launch_bounds__(256) void k_pack(const float* __restrict__ weff, float4* __restrict__ packed, int first_net) {
    const unsigned long long idx = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= PACKED_FLOAT4) return;
    int s = 0;
#pragma unroll 1
    for (int i = 1; i < SEG_COUNT; ++i)
        if (idx >= c_segs[i].off4) s = i;
    const SegDev sd = c_segs[s];
    if (sd.net < first_net) return;
    const unsigned rel = (unsigned)(idx - sd.off4);
    const int lane = rel & 63;
    const int g = (rel >> 6) % sd.kg;
    const int nt = (rel >> 6) / sd.kg;
    const int li = sd.net * LAYERS + sd.layer;
    const int K = c_layer_k[li];
    const float* W = weff + c_weff_off[li];
    const float sc = sd.skip_scale ? INV_SQRT2 : 1.f;
    const int n = 32 * nt + (lane & 31);
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = 8 * g + 2 * j + (lane >> 5);
        float v = 0.f;
        if (k < sd.kreal && n < sd.nreal) {
            const int row = sd.dir == 0 ? sd.row0 + n : sd.row0 + k;
            const int col = sd.dir == 0 ? sd.col0 + k : sd.col0 + n;
            v = sc * W[(size_t)row * K + col];
        }
        o[j] = v;
    }
    packed[idx] = make_float4(o[0], o[1], o[2], o[3]);
}

// one wavefront per weight row: (dW_eff row, g, v) -> (dg, dv row); db copied.
// dW_eff of skip layers is the gradient w.r.t. the 1/sqrt(2)-scaled weight actually used by the kernels.
__global__ __launch_bounds__(256) void k_weightnorm_bwd(const float* __restrict__ params, const float* __restrict__ dweff,
                                                        float* __restrict__ dparams, int first_layer) {
    const int li = first_layer + blockIdx.y;
    const int N = c_layer_n[li], K = c_layer_k[li];
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const float* pb = params + c_param_off[li];
    const float* pg = pb + N;
    const float* pv = pg + N + (size_t)n * K;
    const float* dw = dweff + c_weff_off[li] + (size_t)n * K;
    const float lscale = (li % LAYERS) == 4 ? INV_SQRT2 : 1.f;
    float ss = 0.f, dot = 0.f;
    for (int k = lane; k < K; k += 64) { const float v = pv[k]; ss = fmaf(v, v, ss); dot = fmaf(dw[k] * lscale, v, dot); }
    ss = wave_sum(ss); dot = wave_sum(dot);
    const float inv = 1.f / sqrtf(ss);
    const float dg = dot * inv;                 // d/dg = <dW, v/||v||>
    const float gs = pg[n] * inv;
    float* ob = dparams + c_param_off[li];
    float* og = ob + N;
    float* ov = og + N + (size_t)n * K;
    for (int k = lane; k < K; k += 64) ov[k] = gs * (dw[k] * lscale - dg * inv * pv[k]);
    if (lane == 0) { og[n] = dg; ob[n] = dweff[c_weff_off[li] + (size_t)N * K + n]; }
}

// one thread per float4 of the 16x16x4 packing (forward query segments only)
__global__ __launch_bounds__(256) void k_pack16(const float* __restrict__ weff, float4* __restrict__ packed, int first_net) {
    const unsigned long long idx = PACKED_FLOAT4 + (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= PACKED_TOTAL_FLOAT4) return;
    int pi = 0;
    unsigned long long base = PACKED_FLOAT4, off = PACKED_FLOAT4;
#pragma unroll 1
    for (int i = 0; i < P16_COUNT; ++i) {
        const unsigned long long sz = (unsigned long long)((c_segs[P16_SEGS_DEV(i)].kreal + 15) / 16) * 16 * 64;
        if (idx >= off) { pi = i; base = off; }
        off += sz;
    }
    const SegDev sd = c_segs[P16_SEGS_DEV(pi)];
    if (sd.net < first_net) return;
    const int kg = (sd.kreal + 15) / 16;
    const unsigned rel = (unsigned)(idx - base);
    const int lane = rel & 63;
    const int g = (rel >> 6) % kg;
    const int nt = (rel >> 6) / kg;
    const int li = sd.net * LAYERS + sd.layer;
    const int K = c_layer_k[li];
    const float* W = weff + c_weff_off[li];
    const float sc = sd.skip_scale ? INV_SQRT2 : 1.f;
    const int n = 16 * nt + (lane & 15);
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = 16 * g + 4 * j + (lane >> 4);
        o[j] = (k < sd.kreal && n < sd.nreal) ? sc * W[(size_t)(sd.row0 + n) * K + sd.col0 + k] : 0.f;   // forward orientation only
    }
    packed[idx] = make_float4(o[0], o[1], o[2], o[3]);
}

int weightnorm_pack(const float* params, float* weff, float* packed, int use_deform, hipStream_t st) {
    if (int e = init_tables()) return e;
    const int first_layer = use_deform ? 0 : LAYERS;
    dim3 g1(65, NETS * LAYERS - first_layer);
    hipLaunchKernelGGL(k_weff, g1, dim3(256), 0, st, params, weff, first_layer);
    const unsigned nb = (unsigned)((PACKED_FLOAT4 + 255) / 256);
    hipLaunchKernelGGL(k_pack, dim3(nb), dim3(256), 0, st, (const float*)weff, reinterpret_cast<float4*>(packed), use_deform ? 0 : 1);
    const unsigned nb16 = (unsigned)((PACKED_TOTAL_FLOAT4 - PACKED_FLOAT4 + 255) / 256);
    hipLaunchKernelGGL(k_pack16, dim3(nb16), dim3(256), 0, st, (const float*)weff, reinterpret_cast<float4*>(packed), use_deform ? 0 : 1);
    return hip_last("weightnorm_pack");
}

// layers [first_layer, first_layer + n_layers) of the 27 (3 networks x 9) only: the pieces of a pipelined gradient all-reduce
int weightnorm_backward_layers(const float* params, const float* dweff, float* dparams, int first_layer, int n_layers, hipStream_t st) {
    if (int e = init_tables()) return e;
    if (n_layers <= 0) return ST_OK;
    dim3 g1(65, n_layers);
    hipLaunchKernelGGL(k_weightnorm_bwd, g1, dim3(256), 0, st, params, dweff, dparams, first_layer);
    return hip_last("weightnorm_backward_layers");
}

int weightnorm_backward(const float* params, const float* dweff, float* dparams, int use_deform, hipStream_t st) {
    if (int e = init_tables()) return e;
    const int first_layer = use_deform ? 0 : LAYERS;
    dim3 g1(65, NETS * LAYERS - first_layer);
    hipLaunchKernelGGL(k_weightnorm_bwd, g1, dim3(256), 0, st, params, dweff, dparams, first_layer);
    return hip_last("weightnorm_backward");
}

}  // namespace es
