// Opt-in split-precision INFERENCE kernels of the deformation network on the register-resident GEMM core (x3r_core.h; formulation and
// arithmetic in query_x3r.hip / query_x3.hip).  They replace the fp32 deform_fwd / deform_vjp launches of point_fwd.hip when a no-grad
// point evaluation is requested with PF_X3 (no PF_SAVE):
//   k_deform_jvp_x3r   DeformNetwork (reference endosurf.py:724-738) value + forward-mode tangent along the ray direction:
//                      x_c = x + MLP(x, t) and v = J d.  A wave owns 16 points = 32 columns: lanes 0-15 of a lane half hold the value
//                      column of a point, lanes 16-31 its tangent column; the ReLU mask of a tangent element is the sign of the value
//                      column's pre-activation, fetched from the partner lane.  The masks are kept (128 bits per lane half and layer)
//                      for the VJP sweep.
//   k_deform_vjp_x3r   reverse sweep for the covector g_c: g_o = J^T g_c = g_c + E(x)^T W_0^T M_0 W_1^T ... M_7 W_8^T g_c
//                      (get_sdf_grad_from_observed_space, endosurf.py:581-601, is this product); a wave owns 32 points.
#include "chain_common.h"
#include "launch.h"
#include "x3r_core.h"
#include "tabs.h"
#include "timing.h"
#include "workspace.h"

namespace es {

constexpr int XI_LDS_BYTES = XR_RING * XR_CHUNK_BYTES + (128 * XR_ENC_LD + 8 * 256 + 3 * 256 + 4) * 4;
static_assert(XI_LDS_BYTES <= 160 * 1024, "LDS carve");

// mask bit of register r of feature block b inside the 128-bit word of a (layer, point, lane half)
__device__ __forceinline__ void mask_set(u32x4& mk, int b, int r, bool m) { mk[b >> 1] |= (m ? 1u : 0u) << ((b & 1) * 16 + r); }
__device__ __forceinline__ bool mask_get(const u32x4& mk, int b, int r) { return (mk[b >> 1] >> ((b & 1) * 16 + r)) & 1u; }

// ---- value + tangent ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(XR_THREADS, 1) void k_deform_jvp_x3r(PointSrc src, Tabs tb, const u32x4* __restrict__ chunks, const float* __restrict__ weff,
                                                                float* __restrict__ ws_xc, float* __restrict__ ws_v, u32x4* __restrict__ masks, int Mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsr[];
    float* encs = reinterpret_cast<float*>(ldsr + XR_RING * XR_CHUNK_BYTES);      // [128 columns][68]
    float* biasL = encs + 128 * XR_ENC_LD;                                         // [8 layers][256]
    float* w8L = biasL + 8 * 256;                                                  // [3][256] last-layer rows, then its 3 biases
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const bool tan = n >= 16;                                 // tangent column of point (n & 15)
    const int point = blockIdx.x * 64 + wave * 16 + (n & 15);
    float* erow = encs + (wave * 32 + n) * XR_ENC_LD;
    float x[3], t, d[3];
    load_point(src, point, x, t, d);
    for (int i = tid; i < 8 * 256; i += XR_THREADS) {
        const int l = i >> 8, f = i & 255;
        biasL[i] = f < (l == 3 ? 204 : 256) ? weff[tb.boff[NET_D * LAYERS + l] + f] : 0.f;
    }
    for (int i = tid; i < 3 * 256; i += XR_THREADS) w8L[i] = weff[tb.woff[NET_D * LAYERS + 8] + i];
    if (tid < 3) w8L[3 * 256 + tid] = weff[tb.boff[NET_D * LAYERS + 8] + tid];
    const float* b8 = w8L + 3 * 256;
    // encoding column: value [x, sin, cos ..., t, sin, cos ...]; tangent (d enc / d x) d, the time part has none
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) {
        const int i = 3 * hi + ii;
        const float f = (float)(1 << i);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s, co;
            sincosf(x[c] * f, &s, &co);
            erow[enc_index(3, i, 0, c)] = tan ? f * co * d[c] : s;
            erow[enc_index(3, i, 1, c)] = tan ? -f * s * d[c] : co;
        }
        float s, co;
        sincosf(t * f, &s, &co);
        erow[39 + enc_index(1, i, 0, 0)] = tan ? 0.f : s;
        erow[39 + enc_index(1, i, 1, 0)] = tan ? 0.f : co;
    }
    if (hi == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) erow[c] = tan ? d[c] : x[c];
        erow[39] = tan ? 0.f : t;
    } else {
#pragma unroll
        for (int k = 52; k < 64; ++k) erow[k] = 0.f;
    }
    __syncthreads();
    WStream ws;
    ws.g = chunks; ws.ring = ldsr; ws.k = 0; ws.wave = wave; ws.lane = lane;
    ws.start();

    const auto enc_val = [&](int s, int j) -> float { return erow[16 * s + xr_kperm(hi, j)]; };
    const auto init = [&](f32x16(&C)[8], int l) {            // bias on the value columns only
        init8(C, biasL + l * 256, hi);
        if (tan) {
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) C[b][r] = 0.f;
        }
    };
    f32x16 P[8], C[8];
    init(C, 0);
    gemm_r<4>(C, ws, enc_val);
    copy8(P, C);
    const size_t mrow = ((size_t)point) * 2 + hi;
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        const bool skip = l == 4;                          // IDR skip: input of layer 4 = [h(204) | enc(52)] (1/sqrt2 folded into W4)
        u32x4 mk = {0u, 0u, 0u, 0u};
        init(C, l);
        gemm_r<16>(C, ws, [&](int s, int j) -> float {
            const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
            const int f = 32 * b + 8 * q + 4 * hi + i;
            const float z = P[b][4 * q + i];
            const float zo = __shfl_xor(z, 16);              // the partner column's element
            const bool m = (tan ? zo : z) > 0.f;             // ReLU mask of the VALUE column gates both
            mask_set(mk, b, 4 * q + i, m);
            const float h = m ? z : 0.f;
            if (32 * b + 8 * q + 4 + i < 204) return h;
            return (skip && f >= 204) ? erow[f - 204] : h;
        });
        if (!tan && point < Mp) masks[((size_t)(l - 1) * Mp) * 2 + mrow] = mk;
        copy8(P, C);
    }
    {   // x_c = x + W8 relu(z_7) + b8 on the value columns, v = d + W8 (mask_7 . tau_7) on the tangent columns
        u32x4 mk = {0u, 0u, 0u, 0u};
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = 32 * b + 8 * (r >> 2) + 4 * hi + (r & 3);
                const float z = P[b][r];
                const float zo = __shfl_xor(z, 16);
                const bool m = (tan ? zo : z) > 0.f;
                mask_set(mk, b, r, m);
                const float h = m ? z : 0.f;
                d0 = fmaf(w8L[f], h, d0); d1 = fmaf(w8L[256 + f], h, d1); d2 = fmaf(w8L[512 + f], h, d2);
            }
        d0 += __shfl_xor(d0, 32); d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32);
        if (point < Mp) {
            if (!tan) masks[((size_t)7 * Mp) * 2 + mrow] = mk;
            if (hi == 0) {
                float* o = (tan ? ws_v : ws_xc) + (size_t)point * 3;
                o[0] = tan ? d[0] + d0 : x[0] + d0 + b8[0];
                o[1] = tan ? d[1] + d1 : x[1] + d1 + b8[1];
                o[2] = tan ? d[2] + d2 : x[2] + d2 + b8[2];
            }
        }
    }
}

// ---- reverse sweep ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(XR_THREADS, 1) void k_deform_vjp_x3r(PointSrc src, Tabs tb, const u32x4* __restrict__ chunks, const float* __restrict__ weff,
                                                                const float* __restrict__ ws_gc, float* __restrict__ ws_go,
                                                                const u32x4* __restrict__ masks, int Mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsr[];
    float* adj = reinterpret_cast<float*>(ldsr + XR_RING * XR_CHUNK_BYTES);       // [128 points][68]: adjoint of the 52 encoding inputs
    float* w8L = adj + 128 * XR_ENC_LD + 8 * 256;                                  // [3][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int point = blockIdx.x * 128 + wave * 32 + n;
    const bool live = point < Mp;
    float* arow = adj + (wave * 32 + n) * XR_ENC_LD;
    float x[3], t, d[3];
    load_point(src, point, x, t, d);
    float g[3] = {0.f, 0.f, 0.f};
    if (live) { g[0] = ws_gc[(size_t)point * 3]; g[1] = ws_gc[(size_t)point * 3 + 1]; g[2] = ws_gc[(size_t)point * 3 + 2]; }
    for (int i = tid; i < 3 * 256; i += XR_THREADS) w8L[i] = weff[tb.woff[NET_D * LAYERS + 8] + i];
    const size_t mrow = ((size_t)(live ? point : 0)) * 2 + hi;
    u32x4 mk = masks[((size_t)7 * Mp) * 2 + mrow];
    __syncthreads();
    WStream ws;
    ws.g = chunks; ws.ring = ldsr; ws.k = XR_DR_CHUNK0; ws.wave = wave; ws.lane = lane;
    ws.start();

    f32x16 P[8], C[8];
    const auto zero = [&](f32x16(&A)[8]) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) A[b][r] = 0.f;
    };
    // r_7 = mask_7 . (W8^T g_c)  ->  adjoint of h_6 = W_7^T r_7
    zero(C);
    gemm_r<16>(C, ws, [&](int s, int j) -> float {
        const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
        const int f = 32 * b + 8 * q + 4 * hi + i;
        const float v = fmaf(w8L[f], g[0], fmaf(w8L[256 + f], g[1], w8L[512 + f] * g[2]));
        return mask_get(mk, b, 4 * q + i) ? v : 0.f;
    });
    copy8(P, C);
    // layers 6 .. 1: r_l = mask_l . (adjoint of h_l),  adjoint of h_{l-1} = W_l^T r_l
#pragma unroll 1
    for (int l = 6; l >= 1; --l) {
        mk = masks[((size_t)l * Mp) * 2 + mrow];
        if (l == 3) {           // the output of DR4 is the adjoint of [h_3 (204) | enc (52)]: keep the encoding part
#pragma unroll
            for (int b = 6; b < 8; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 32 * b + 8 * (r >> 2) + 4 * hi + (r & 3);
                    if (f >= 204) arow[f - 204] = P[b][r];
                }
        }
        zero(C);
        const auto val = [&](int s, int j) -> float {
            const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
            return mask_get(mk, b, 4 * q + i) ? P[b][4 * q + i] : 0.f;      // layer 3: mask bits of features >= 204 are 0
        };
        if (l == 3) gemm_r<14>(C, ws, val);     // 204 outputs of layer 3: 14 k-steps (the rest is zero padding in DR3)
        else gemm_r<16>(C, ws, val);
        copy8(P, C);
    }
    // r_0 = mask_0 . (adjoint of h_0);  adjoint of the encoding += W_0^T r_0 (52 outputs: accumulator group 0 only)
    mk = masks[mrow];
    zero(C);
    gemm_r<16, 1>(C, ws, [&](int s, int j) -> float {
        const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
        return mask_get(mk, b, 4 * q + i) ? P[b][4 * q + i] : 0.f;
    });
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * b + 8 * (r >> 2) + 4 * hi + (r & 3);
            if (f < 52) arow[f] += C[b][r];
        }
    // the two lane halves of a point wrote disjoint features of its row; same wave, so the LDS writes are ordered before the reads
    if (hi == 0 && live) {   // g_o[j] = g_c[j] + sum_k adj[k] * d enc_k / d x_j   (position part of the encoding, observed-space x)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float go = g[j] + arow[j];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float f = (float)(1 << i);
                float s, co;
                sincosf(x[j] * f, &s, &co);
                go += f * (arow[enc_index(3, i, 0, j)] * co - arow[enc_index(3, i, 1, j)] * s);
            }
            ws_go[(size_t)point * 3 + j] = go;
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------------------
static int infer_attrs() {
    static DeviceOnce attr_done;
    if (attr_done.first()) {
        if (int e = allow_big_lds(k_deform_jvp_x3r, XI_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_deform_vjp_x3r, XI_LDS_BYTES)) return e;
        attr_done.done();
    }
    return ST_OK;
}
// packed_r = the k-step-ordered split weights of pack_x3r (query_x3r.hip)
int deform_jvp_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, hipStream_t st) {
    if (int e = infer_attrs()) return e;
    const Tabs tb = make_tabs();
    ScopedTimer tm(KID_DEFORM_FWD, src.M, st);
    hipLaunchKernelGGL(k_deform_jvp_x3r, dim3(L.Mp / 64), dim3(XR_THREADS), XI_LDS_BYTES, st, src, tb, reinterpret_cast<const u32x4*>(packed_r), weff,
                       ws + L.off[WS_XC], ws + L.off[WS_V], reinterpret_cast<u32x4*>(ws + L.off[WS_D_MASK]), L.Mp);
    return hip_last("deform_jvp_x3r");
}
int deform_vjp_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, hipStream_t st) {
    if (int e = infer_attrs()) return e;
    const Tabs tb = make_tabs();
    ScopedTimer tm(KID_DEFORM_VJP, src.M, st);
    hipLaunchKernelGGL(k_deform_vjp_x3r, dim3((L.Mp + 127) / 128), dim3(XR_THREADS), XI_LDS_BYTES, st, src, tb, reinterpret_cast<const u32x4*>(packed_r), weff,
                       ws + L.off[WS_GC], ws + L.off[WS_GO], reinterpret_cast<const u32x4*>(ws + L.off[WS_D_MASK]), L.Mp);
    return hip_last("deform_vjp_x3r");
}

}  // namespace es
