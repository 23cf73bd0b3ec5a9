// Opt-in split-precision TRAINING chain on the register-resident GEMM core (x3r_core.h): the backward counterparts of the kernels in
// infer_x3r.hip (which, with SAVE, are the training forward).  Same arithmetic as every split-precision kernel (three exact bf16 planes per
// fp32 operand, six partial products on v_mfma_f32_32x32x16_bf16, fp32 accumulation), same workspace buffers and layouts as the fp32
// kernels of point_bwd.hip, so that the weight-gradient GEMMs (wgrad.hip) and the other networks' kernels do not care which family
// produced a stack.  The only family-internal buffers are the ReLU mask words (WS_D_MASK: 128 bits per (layer, point, lane half), as
// k_deform_jvp_x3r writes them): a workspace region evaluated by this family's forward must go through this family's backward.
//   k_deform_tan_x3r   forward tangent sweep of the deformation network along gbar_o (point_bwd.hip deform_tan_tile): the adjoint of
//                      g_o = J^T g_c with respect to g_c is J gbar_o.  tau_0 = E(x) gbar_o, tau_{l+1} = M_l (W_l tau_l),
//                      J gbar_o = gbar_o + W_8 tau_8; tau_0 .. tau_8 are kept (WS_D_T0 / WS_D_T).  A wave owns 32 points.
//   k_deform_bwd_x3r   reverse sweep of the value row (seed xbar_c) and of the J d row (seed vbar) of the deformation network
//                      (deform_bwd_tile): abar_7 = M_7 (W_8^T abar_8), abar_{l-1} = M_{l-1} (W_l^T abar_l); abar_0 .. abar_7 are kept
//                      (WS_D_A, 2 rows per point like WS_D_U).  A wave owns 16 points = 32 columns.
#include "chain_common.h"
#include "launch.h"
#include "x3r_core.h"
#include "tabs.h"
#include "timing.h"
#include "workspace.h"
#include "point_bwd_bodies.h"

namespace es {

constexpr int XT_LDS_BYTES = XR_RING * XR_CHUNK_BYTES + (128 * XR_ENC_LD + 3 * 256 + 4) * 4 + XR_TILE_BYTES;      // + the save tiles (RowTile)
static_assert(XT_LDS_BYTES <= 160 * 1024, "LDS carve");

__device__ __forceinline__ void zero8(f32x16 (&A)[8]) {
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) A[b][r] = 0.f;
}

// ---- tangent sweep along gbar_o -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(XR_THREADS, 1) void k_deform_tan_x3r(PointSrc src, Tabs tb, const u32x4* __restrict__ chunks, const float* __restrict__ weff,
                                                                const float* __restrict__ d_go, const u32x4* __restrict__ masks,
                                                                float* __restrict__ T0, float* __restrict__ T, float* __restrict__ ws_ju, int Mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsr[];
    float* encs = reinterpret_cast<float*>(ldsr + XR_RING * XR_CHUNK_BYTES);      // [128 points][68]: tau_0
    float* w8L = encs + 128 * XR_ENC_LD;                                           // [3][256] last-layer rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int point = blockIdx.x * 128 + wave * 32 + n;
    float* erow = encs + (wave * 32 + n) * XR_ENC_LD;      // (every point of the block is a workspace row: Mp is a multiple of 128)
    float x[3], t, d[3];
    load_point(src, point, x, t, d);
    float g[3] = {0.f, 0.f, 0.f};
    if (point < src.M) { g[0] = d_go[(size_t)point * 3]; g[1] = d_go[(size_t)point * 3 + 1]; g[2] = d_go[(size_t)point * 3 + 2]; }
    for (int i = tid; i < 3 * 256; i += XR_THREADS) w8L[i] = weff[tb.woff[NET_D * LAYERS + 8] + i];
    // tau_0 = (d enc / d x) gbar_o; the time part of the encoding has no tangent
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) {
        const int i = 3 * hi + ii;
        const float f = (float)(1 << i);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s, co;
            sincosf(x[c] * f, &s, &co);
            erow[enc_index(3, i, 0, c)] = f * co * g[c];
            erow[enc_index(3, i, 1, c)] = -f * s * g[c];
        }
        erow[39 + enc_index(1, i, 0, 0)] = 0.f;
        erow[39 + enc_index(1, i, 1, 0)] = 0.f;
    }
    if (hi == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) erow[c] = g[c];
        erow[39] = 0.f;
    } else {
#pragma unroll
        for (int k = 52; k < 64; ++k) erow[k] = 0.f;
    }
    __syncthreads();
    const size_t prow = point;
#pragma unroll
    for (int k = 0; k < 32; k += 4) st4(T0 + prow * 64 + 32 * hi + k, erow[32 * hi + k], erow[32 * hi + k + 1], erow[32 * hi + k + 2], erow[32 * hi + k + 3]);
    WStream ws;
    ws.g = chunks; ws.ring = ldsr; ws.k = 0; ws.wave = wave; ws.lane = lane;
    ws.start();

    const auto enc_val = [&](int s, int j) -> float { return erow[16 * s + xr_kperm(hi, j)]; };
    const size_t mrow = prow * 2 + hi;
    const size_t tstride = (size_t)Mp * 256;
    float* Trow = T + prow * 256 + 4 * hi;
    const RowTile rt{w8L + 3 * 256 + 4 + wave * 32 * XR_TILE_LD, n, hi, lane};
    float* Twave = T + ((size_t)blockIdx.x * 128 + wave * 32) * 256;
    f32x16 P[8], C[8];
    zero8(C);
    gemm_r<4>(C, ws, enc_val);
    copy8_acc(P, C);
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        const bool skip = l == 4;                          // IDR skip: input of layer 4 = [tau(204) | tau_0(52)] (1/sqrt2 folded into W4)
        const u32x4 mk = masks[((size_t)(l - 1) * Mp) * 2 + mrow];
        float* Tl = Twave + (size_t)(l - 1) * tstride;     // tau_l = this GEMM's operand
        zero8(C);
        gemm_rs<16, 2, false, 4>(C, ws, [&](int s, int j) -> float {
            const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
            const int f = 32 * b + 8 * q + 4 * hi + i;
            const float h = mask_get(mk, b, 4 * q + i) ? P[b][4 * q + i] : 0.f;      // layer 3: mask bits of features >= 204 are 0
            if (32 * b + 8 * q + 4 + i < 204) return h;
            return (skip && f >= 204) ? erow[f - 204] : h;
        }, NoSide(), [&](int s, const float (&v)[8]) { rt.put<256>(s, v, Tl); });
        copy8_acc(P, C);
    }
    {   // tau_8 = mask_7 . (W_7 tau_7);  J gbar_o = gbar_o + W_8 tau_8
        const u32x4 mk = masks[((size_t)7 * Mp) * 2 + mrow];
        float* T8 = Trow + (size_t)7 * tstride;
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float h4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = 32 * b + 8 * q + 4 * hi + i;
                    const float h = mask_get(mk, b, 4 * q + i) ? P[b][4 * q + i] : 0.f;
                    h4[i] = h;
                    d0 = fmaf(w8L[f], h, d0); d1 = fmaf(w8L[256 + f], h, d1); d2 = fmaf(w8L[512 + f], h, d2);
                }
                st4(T8 + 32 * b + 8 * q, h4[0], h4[1], h4[2], h4[3]);
            }
        d0 += __shfl_xor(d0, 32); d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32);
        if (hi == 0) {
            float* o = ws_ju + (size_t)point * 3;
            o[0] = g[0] + d0; o[1] = g[1] + d1; o[2] = g[2] + d2;
        }
    }
}

// ---- reverse sweep of the value and J d rows --------------------------------------------------------------------------------------
// lanes 0-15 of a lane half: the value row's adjoint of a point, lanes 16-31: the J d row's; both are gated by the value row's ReLU mask
__device__ __forceinline__ void deform_bwd_x3r_body(const Tabs& tb, const u32x4* __restrict__ chunks, const float* __restrict__ weff,
                                                    const float* __restrict__ xcbar, const float* __restrict__ vbar, int M_color,
                                                    const u32x4* __restrict__ masks, float* __restrict__ A, float* __restrict__ A8, int Mp, const int blk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsr[];
    float* w8L = reinterpret_cast<float*>(ldsr + XR_RING * XR_CHUNK_BYTES) + 128 * XR_ENC_LD;      // [3][256] (same carve as the tangent sweep)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const bool tan = n >= 16;
    const int point = blk * 64 + wave * 16 + (n & 15);             // < Mp
    const size_t arow = (size_t)point * 2 + (tan ? 1 : 0);
    const size_t rows2 = (size_t)Mp * 2;
    float a8[3];
    {
        const bool has_v = point < M_color;                     // vbar exists for the points that went through the colour network
#pragma unroll
        for (int i = 0; i < 3; ++i) a8[i] = tan ? (has_v ? vbar[(size_t)point * 3 + i] : 0.f) : xcbar[(size_t)point * 3 + i];
        if (hi == 0) *reinterpret_cast<float4*>(A8 + arow * 4) = make_float4(a8[0], a8[1], a8[2], 0.f);
    }
    for (int i = tid; i < 3 * 256; i += XR_THREADS) w8L[i] = weff[tb.woff[NET_D * LAYERS + 8] + i];
    const size_t mrow = (size_t)point * 2 + hi;
    u32x4 mk = masks[((size_t)7 * Mp) * 2 + mrow];
    __syncthreads();
    WStream ws;
    ws.g = chunks; ws.ring = ldsr; ws.k = XR_DR_CHUNK0; ws.wave = wave; ws.lane = lane;
    ws.start();

    float* Arow = A + arow * 256 + 4 * hi;
    const size_t astride = rows2 * 256;
    int lsave = 7;
    const RowTile rt{w8L + 3 * 256 + 4 + wave * 32 * XR_TILE_LD, 2 * (n & 15) + (tan ? 1 : 0), hi, lane};
    float* Awave = A + ((size_t)blk * 64 + wave * 16) * 2 * 256;              // the wave's 32 consecutive rows
    const auto asink = [&](int s, const float (&v)[8]) { rt.put<256>(s, v, Awave + lsave * astride); };
    f32x16 P[8], C[8];
    // abar_7 = mask_7 . (W8^T abar_8)  ->  adjoint of h_6 = W_7^T abar_7
    zero8(C);
    gemm_rs<16, 2, false, 4>(C, ws, [&](int s, int j) -> float {
        const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
        const int f = 32 * b + 8 * q + 4 * hi + i;
        const float v = fmaf(w8L[f], a8[0], fmaf(w8L[256 + f], a8[1], w8L[512 + f] * a8[2]));
        return mask_get(mk, b, 4 * q + i) ? v : 0.f;
    }, NoSide(), asink);
    copy8_acc(P, C);
#pragma unroll 1
    for (int l = 6; l >= 1; --l) {
        mk = masks[((size_t)l * Mp) * 2 + mrow];
        zero8(C);
        const auto val = [&](int s, int j) -> float {
            const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
            return mask_get(mk, b, 4 * q + i) ? P[b][4 * q + i] : 0.f;      // layer 3: mask bits of features >= 204 are 0 (the skip's
        };                                                                   // encoding part carries no parameter gradient)
        lsave = l;
        if (l == 3) {
            gemm_rs<14, 2, false, 4>(C, ws, val, NoSide(), asink);
            const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            st_kstep(Arow + 3 * astride, 14, z8); st_kstep(Arow + 3 * astride, 15, z8);
        } else gemm_rs<16, 2, false, 4>(C, ws, val, NoSide(), asink);
        copy8_acc(P, C);
    }
    // abar_0 = mask_0 . (W_1^T abar_1): no further GEMM (the adjoint of the encoding input has no parameter gradient)
    mk = masks[mrow];
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float h4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) h4[i] = mask_get(mk, b, 4 * q + i) ? P[b][4 * q + i] : 0.f;
            st4(Arow + 32 * b + 8 * q, h4[0], h4[1], h4[2], h4[3]);
        }
}

__global__ __launch_bounds__(XR_THREADS, 1) void k_deform_bwd_x3r(Tabs tb, const u32x4* __restrict__ chunks, const float* __restrict__ weff,
                                                                const float* __restrict__ xcbar, const float* __restrict__ vbar, int M_color,
                                                                const u32x4* __restrict__ masks, float* __restrict__ A, float* __restrict__ A8, int Mp) {
    deform_bwd_x3r_body(tb, chunks, weff, xcbar, vbar, M_color, masks, A, A8, Mp, (int)blockIdx.x);
}
// the backward counterpart of k_deform_jvp_x3r_tail (infer_x3r.hip): the colour-less tail's tangent sweep + SDF backward as fp32 tile bodies
// (blocks [0, n0): tiles t0 ..) at the head of the main batch's deformation reverse sweep; the tail's own deformation reverse sweep
// follows as a small fp32 launch
__global__ __launch_bounds__(XR_THREADS, 1) void k_deform_bwd_x3r_tail(BwdArgs ba, int n0, int t0, const u32x4* __restrict__ chunks,
                                                                     const u32x4* __restrict__ masks) {
    if ((int)blockIdx.x < n0) {
        deform_tan_tile(ba, t0 + (int)blockIdx.x);
        __syncthreads();
        sdf_bwd_tile(ba, t0 + (int)blockIdx.x);
        return;
    }
    deform_bwd_x3r_body(ba.tb, chunks, ba.weff, ba.ws + ba.L.off[WS_XCBAR], ba.ws + ba.L.off[WS_VBAR_C], ba.M_color, masks, ba.ws + ba.L.off[WS_D_A],
                        ba.ws + ba.L.off[WS_D_A8], ba.L.Mp, (int)blockIdx.x - n0);
}

// ---- colour network, reverse sweep -------------------------------------------------------------------------------------------------
// ColorNetwork backward (point_bwd.hip color_bwd_tile): ybar_8 = rgbbar . rgb (1 - rgb), ybar_7 = M_7 (U_8^T ybar_8),
// ybar_{l-1} = M_{l-1} (U_l^T ybar_l); the adjoint of the network input [small(93) | feat(256)] comes from layer 0 and from the skip
// layer: its feature part goes to WS_FEATBAR (through HBM between the two layers), its small part accumulates in an LDS row per
// point and is turned into the adjoints of x_c (enc10), g_c and v = J d (enc4 of d_c = v / (|v| + 1e-10)) at the end.  ybar_0 .. ybar_7
// -> WS_C_Y, ybar_8 -> WS_C_Y8 (the dA operands of the weight-gradient GEMMs).  A wave owns 32 points; launched over whole blocks.
constexpr int XCB_SB_LD = 97;        // floats per point row of the small part's adjoint (96 used; odd: conflict-free)
constexpr int XCB_LDS_BYTES = XR_RING * XR_CHUNK_BYTES + (128 * XCB_SB_LD + 3 * 256 + 4) * 4;
static_assert(XCB_LDS_BYTES <= 160 * 1024, "LDS carve");

template <int L>
__device__ __forceinline__ float enc3_adjoint_row(const float* adj, int j, float x) {      // sum_k adj[k] d enc_k / d x_j, 3-D encoding, L frequencies
    float g = adj[j];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const float f = (float)(1 << i);
        float s, co;
        sincosf(x * f, &s, &co);
        g += f * (adj[enc_index(3, i, 0, j)] * co - adj[enc_index(3, i, 1, j)] * s);
    }
    return g;
}

template <bool DEFORM>
__global__ __launch_bounds__(XR_THREADS, 1) void k_color_bwd_x3r(PointSrc src, Tabs tb, const u32x4* __restrict__ chunks, const float* __restrict__ weff,
                                                               const float* __restrict__ d_rgb, int M_color, const float* __restrict__ ws_rgb,
                                                               const float* __restrict__ ws_xc, const float* __restrict__ ws_v,
                                                               const u32x4* __restrict__ masks, float* __restrict__ CY, float* __restrict__ CY8,
                                                               float* __restrict__ FB, float* __restrict__ xcbar_c, float* __restrict__ gcbar_c,
                                                               float* __restrict__ vbar_c, int Mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsr[];
    float* sbar = reinterpret_cast<float*>(ldsr + XR_RING * XR_CHUNK_BYTES);       // [128 points][97]
    float* w8L = sbar + 128 * XCB_SB_LD;                                            // [3][256] last-layer rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int point = blockIdx.x * 128 + wave * 32 + n;          // a workspace row (Mp is a multiple of 128)
    float* srow = sbar + (wave * 32 + n) * XCB_SB_LD;
    float y8[3];
    {
        const bool valid = point < M_color;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float c = ws_rgb[(size_t)point * 3 + i];
            y8[i] = valid ? d_rgb[(size_t)point * 3 + i] * c * (1.f - c) : 0.f;      // sigmoid'
        }
        if (hi == 0) *reinterpret_cast<float4*>(CY8 + (size_t)point * 4) = make_float4(y8[0], y8[1], y8[2], 0.f);
    }
    for (int i = tid; i < 3 * 256; i += XR_THREADS) w8L[i] = weff[tb.woff[NET_C * LAYERS + 8] + i];
    const size_t mrow = (size_t)point * 2 + hi;
    u32x4 mk = masks[((size_t)7 * Mp) * 2 + mrow];
    __syncthreads();
    WStream ws;
    ws.g = chunks; ws.ring = ldsr; ws.k = 0; ws.wave = wave; ws.lane = lane; ws.b0 = XR_CR_CHUNK0;
    ws.start();

    float* Yrow = CY + (size_t)point * 256 + 4 * hi;
    const size_t ystride = (size_t)Mp * 256;
    int lsave = 7;
    const auto ysink = [&](int s, const float (&v)[8]) { st_kstep(Yrow + lsave * ystride, s, v); };
    f32x16 P[8], C[8];
    zero8(C);
    gemm_rs<16, 2, false, 2>(C, ws, [&](int s, int j) -> float {               // ybar_7 = mask_7 . (U8^T ybar_8)
        const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
        const int f = 32 * b + 8 * q + 4 * hi + i;
        const float v = fmaf(w8L[f], y8[0], fmaf(w8L[256 + f], y8[1], w8L[512 + f] * y8[2]));
        return mask_get(mk, b, 4 * q + i) ? v : 0.f;
    }, NoSide(), ysink);
    copy8_acc(P, C);
    const auto val = [&](int s, int j) -> float {                // ybar_l = mask_l . (adjoint of h_{l+1})
        const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
        return mask_get(mk, b, 4 * q + i) ? P[b][4 * q + i] : 0.f;
    };
    float* FBrow = FB + (size_t)point * 256 + 4 * hi;
#pragma unroll 1
    for (int l = 6; l >= 0; --l) {
        mk = masks[((size_t)l * Mp) * 2 + mrow];
        lsave = l;
        if (l == 4 || l == 0) {          // adjoint of the network input: feature part (256), then small part (93: accumulator group 0)
            zero8(C);
            gemm_rs<16, 2, false, 2>(C, ws, val, NoSide(), ysink);
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float* o = FBrow + 32 * b + 8 * q;
                    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (l == 0) p = *reinterpret_cast<const float4*>(o);
                    *reinterpret_cast<float4*>(o) = make_float4(p.x + C[b][4 * q], p.y + C[b][4 * q + 1], p.z + C[b][4 * q + 2], p.w + C[b][4 * q + 3]);
                }
            zero8(C);
            gemm_r<16, 1>(C, ws, val);
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 32 * b + 8 * (r >> 2) + 4 * hi + (r & 3);
                    srow[f] = l == 4 ? C[b][r] : srow[f] + C[b][r];
                }
            if (l == 0) break;
            zero8(C);
            gemm_r<16>(C, ws, val);       // hidden part of the skip layer's input
        } else {
            zero8(C);
            gemm_rs<16, 2, false, 2>(C, ws, val, NoSide(), ysink);
        }
        copy8_acc(P, C);
    }
    // the two lane halves of a point wrote disjoint features of its row; same wave, so the LDS writes are ordered before the reads
    if (hi == 0) {
        float xc[3], dc[3], v[3];
        {
            float x[3], t, d[3];
            load_point(src, point, x, t, d);
#pragma unroll
            for (int c = 0; c < 3; ++c) { xc[c] = ws_xc[(size_t)point * 3 + c]; v[c] = DEFORM ? ws_v[(size_t)point * 3 + c] : d[c]; }
        }
        const float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const float den = nrm + 1e-10f;
#pragma unroll
        for (int c = 0; c < 3; ++c) dc[c] = v[c] / den;
        float db[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            xcbar_c[(size_t)point * 3 + j] = enc3_adjoint_row<10>(srow, j, xc[j]);
            gcbar_c[(size_t)point * 3 + j] = srow[63 + j];
            db[j] = enc3_adjoint_row<4>(srow + 66, j, dc[j]);
        }
        if (DEFORM) {      // d_c = v / (|v| + eps), v = J d  ->  vbar (seeds the J d row of the deformation backward)
            const float dot = v[0] * db[0] + v[1] * db[1] + v[2] * db[2];
            const float k2 = nrm > 0.f ? dot / (nrm * den * den) : 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) vbar_c[(size_t)point * 3 + i] = db[i] / den - v[i] * k2;
        } else {
            vbar_c[(size_t)point * 3] = vbar_c[(size_t)point * 3 + 1] = vbar_c[(size_t)point * 3 + 2] = 0.f;
        }
    }
}

// ---- SDF network backward: tangent sweep + reverse sweep ---------------------------------------------------------------------------
// point_bwd.hip sdf_bwd_tile as two kernels (each keeps two accumulator sets, the fragments and one register-staged operand stream):
//   k_sdf_tan_x3r   (i) forward tangent sweep along gbar_c -- the adjoint of the reverse-mode input gradient g_c is a forward-mode
//                   directional derivative: tau_0 = (d enc6 / d x_c) gbar_c, pi_l = W_l tau_l, tau_{l+1} = phi'(z_l) pi_l, and the
//                   second-order terms zeta_l = 100 (1 - phi'(z_l)) rho_l pi_l (softplus'' / softplus' = 100 (1 - softplus'));
//                   tau_0 .. tau_8 -> WS_S_TAU0 / WS_S_TAU, zeta_0 .. zeta_7 -> WS_S_ZB
//   k_sdf_rev_x3r   (ii) reverse sweep of the value pass seeded with [sdfbar | featbar]: zbar_l = phi'(z_l) sbar_{l+1} + zeta_l,
//                   sbar_l = W_l^T zbar_l; zbar_0 .. zbar_7 overwrite WS_S_ZB; adjoint of x_c (enc6 adjoint + the encoding's own
//                   second-order term + the colour network's) -> WS_XCBAR
// phi'(z_l) comes from s_{l+1} (WS_S_ACT, a side stream of direct loads into LDS as in k_sdf_fwd_x3r); the second per-element operand
// (rho_l / zeta_l / featbar) is staged through registers: two 16-B loads per lane and k-step, issued with the weight pipeline's loads
// two k-steps ahead into four rotating register sets.  All stacks row-major (this family's SDF layout, see k_sdf_fwd_x3r).
constexpr int XSB_LDS_BYTES = XR_RING * XR_CHUNK_BYTES + XR_RING * 4 * 2048 + (128 * 41 + 256 + 4) * 4;
static_assert(XSB_LDS_BYTES <= 160 * 1024, "LDS carve");
__device__ __forceinline__ void ld8(float (&r)[8], const float* p) {      // the two 16-B pieces of a k-step's operand in a row-major row (p holds + 4 hi)
    const v4f_frag a = __builtin_nontemporal_load(reinterpret_cast<const v4f_frag*>(p));
    const v4f_frag b = __builtin_nontemporal_load(reinterpret_cast<const v4f_frag*>(p + 8));
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
}

__global__ __launch_bounds__(XR_THREADS, 1) void k_sdf_tan_x3r(PointSrc src, Tabs tb, const u32x4* __restrict__ chunks, const float* __restrict__ ws_xc,
                                                             const float* __restrict__ gbar_main, const float* __restrict__ gbar_color, int M_color, int M,
                                                             const float* __restrict__ SACT, const float* __restrict__ RHO, float* __restrict__ TAU0,
                                                             float* __restrict__ TAU, float* __restrict__ ZB, int Mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsr[];
    unsigned char* sring = ldsr + XR_RING * XR_CHUNK_BYTES;
    float* encs = reinterpret_cast<float*>(sring + XR_RING * 4 * 2048);            // [128 points][41]: tau_0
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int point = blockIdx.x * 128 + wave * 32 + n;          // a workspace row (Mp is a multiple of 128)
    float* erow = encs + (wave * 32 + n) * 41;
    float x[3], g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        x[c] = ws_xc[(size_t)point * 3 + c];
        // gbar_c = J gbar_o from the deformation tangent sweep (or gbar_o itself without a deformation network) + the colour network's
        g[c] = (point < M ? gbar_main[(size_t)point * 3 + c] : 0.f) + (point < M_color ? gbar_color[(size_t)point * 3 + c] : 0.f);
    }
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) {
        const int i = 3 * hi + ii;
        const float f = (float)(1 << i);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s, co;
            sincosf(x[c] * f, &s, &co);
            erow[enc_index(3, i, 0, c)] = f * co * g[c];
            erow[enc_index(3, i, 1, c)] = -f * s * g[c];
        }
    }
    if (hi == 0) { erow[0] = g[0]; erow[1] = g[1]; erow[2] = g[2]; erow[39] = 0.f; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 20; k += 4) st4(TAU0 + (size_t)point * 64 + 20 * hi + k, erow[20 * hi + k], erow[20 * hi + k + 1], erow[20 * hi + k + 2], erow[20 * hi + k + 3]);
    const size_t lstride = (size_t)Mp * 256;
    const float* Srow = SACT + (size_t)point * 256 + 4 * hi;
    const float* Rrow = RHO + (size_t)point * 256 + 4 * hi;
    WStream ws;
    ws.g = chunks; ws.ring = ldsr; ws.k = 0; ws.wave = wave; ws.lane = lane; ws.b0 = XR_SDF_CHUNK0;
    ws.start();

    // logical k-step kk -> (layer l = 1 .. 7 whose GEMM it belongs to, operand k-step s) or none (SF0, SF4A): SF0 [0, 4), l = 1 .. 4 at
    // 4 + 16 (l - 1), SF4A [68, 72), l = 5 .. 7 at 72 + 16 (l - 5)
    const auto where = [&](int kk, int& l, int& s) -> bool {
        if (kk < 4 || (kk >= 68 && kk < 72) || kk >= 120) return false;
        const int r = kk < 68 ? kk - 4 : kk - 8;
        l = 1 + (r >> 4); s = r & 15;
        return true;
    };
    float rb[4][8];                                               // rho_{l-1} of operand k-steps (set = k-step & 3)
    const auto side = [&](int kk, int t, int sloc) {
        int l, s;
        if ((t != 1 && t != 4 && t != 2) || !where(kk, l, s)) return;
        if (t == 2) { ld8(rb[sloc & 3], Rrow + (size_t)(l - 1) * lstride + 16 * s); return; }
        const int piece = t == 4;
        const float* srcp = Srow + (size_t)(l - 1) * lstride + 16 * s + 8 * piece;              // s_l = softplus(z_{l-1})
        unsigned char* dst = sring + (((kk & (XR_RING - 1)) * 4 + wave) * 2 + piece) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcp, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    const auto enc_val = [&](int s, int j) -> float {
        const int k = 16 * s + xr_kperm(hi, j);
        return k < 39 ? erow[k] : 0.f;
    };
    f32x16 P[8], C[8];
    zero8(C);
    gemm_r<4>(C, ws, enc_val, side);                              // pi_0 (its k-steps 2, 3 stage the first two operand k-steps of layer 1)
    copy8_acc(P, C);
    int kb = 0;
    float z2v[8];
    float* Trow = TAU + (size_t)point * 256 + 4 * hi;
    float* Zrow = ZB + (size_t)point * 256 + 4 * hi;
    int lcur = 1;
    const auto tau_val = [&](int s, int j) -> float {            // tau_l = phi'(z_{l-1}) pi_{l-1};  zeta_{l-1} alongside
        const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
        const float sv = reinterpret_cast<const float*>(sring + ((((kb + s) & (XR_RING - 1)) * 4 + wave) * 2 + (j >> 2)) * 1024)[lane * 4 + i];
        const float dphi = dphi_from_s(sv), pi = P[b][4 * q + i];
        z2v[j] = 100.f * (1.f - dphi) * rb[s & 3][j] * pi;
        return dphi * pi;
    };
    const auto tsink = [&](int s, const float (&v)[8]) {
        st_kstep(Trow + (size_t)(lcur - 1) * lstride, s, v);
        st_kstep(Zrow + (size_t)(lcur - 1) * lstride, s, z2v);
    };
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        lcur = l;
        kb = ws.k;
        zero8(C);
        gemm_rs<16, 2, true, 4>(C, ws, tau_val, side, tsink);
        if (l == 4) gemm_r<4>(C, ws, enc_val, side);              // NeRF skip: + W_4[:, 256:] tau_0
        copy8_acc(P, C);
    }
    {   // tau_8 = phi'(z_7) pi_7 and zeta_7: no further GEMM (the tangent of the last layer is not needed)
        const float* S8 = Srow + (size_t)7 * lstride;
        const float* R7 = Rrow + (size_t)7 * lstride;
        float* T8 = Trow + (size_t)7 * lstride;
        float* Z7 = Zrow + (size_t)7 * lstride;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4f_frag sv = __builtin_nontemporal_load(reinterpret_cast<const v4f_frag*>(S8 + 32 * b + 8 * q));
                const v4f_frag rv = __builtin_nontemporal_load(reinterpret_cast<const v4f_frag*>(R7 + 32 * b + 8 * q));
                float t4[4], z4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float dphi = dphi_from_s(sv[i]), pi = P[b][4 * q + i];
                    t4[i] = dphi * pi;
                    z4[i] = 100.f * (1.f - dphi) * rv[i] * pi;
                }
                st4(T8 + 32 * b + 8 * q, t4[0], t4[1], t4[2], t4[3]);
                st4(Z7 + 32 * b + 8 * q, z4[0], z4[1], z4[2], z4[3]);
            }
    }
}

template <bool COLOR>
__global__ __launch_bounds__(XR_THREADS, 1) void k_sdf_rev_x3r(Tabs tb, const u32x4* __restrict__ chunks, const float* __restrict__ weff,
                                                             const float* __restrict__ ws_xc, const float* __restrict__ d_sdf, int M,
                                                             const float* __restrict__ gbar_main, const float* __restrict__ gbar_color,
                                                             const float* __restrict__ xcbar_color, const float* __restrict__ FBAR, int M_color,
                                                             const float* __restrict__ SACT, float* __restrict__ ZB, const float* __restrict__ ADJEPS,
                                                             float* __restrict__ xcbar, int Mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsr[];
    unsigned char* sring = ldsr + XR_RING * XR_CHUNK_BYTES;
    float* encs = reinterpret_cast<float*>(sring + XR_RING * 4 * 2048);            // [128 points][41]: adjoint of the encoding
    float* w8L = encs + 128 * 41;                                                  // [256] row 0 of the last layer
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int point = blockIdx.x * 128 + wave * 32 + n;          // a workspace row (Mp is a multiple of 128)
    float* erow = encs + (wave * 32 + n) * 41;
    const bool colored = COLOR && point < M_color;
    const float sb = point < M ? d_sdf[point] : 0.f;
    for (int i = tid; i < 256; i += XR_THREADS) w8L[i] = weff[tb.woff[NET_S * LAYERS + 8] + i];
    __syncthreads();
    const size_t lstride = (size_t)Mp * 256;
    const float* Srow = SACT + (size_t)point * 256 + 4 * hi;
    float* Zrow = ZB + (size_t)point * 256 + 4 * hi;
    const float* Frow = FBAR + (size_t)point * 256 + 4 * hi;
    // logical k-steps: (COLOR: SR8F [0, 16),) then SR7 SR6 SR5 SR4A SR4M SR3 SR2 SR1 SR0; operand layer of GEMM gi: 7 6 5 4 4 3 2 1 0
    constexpr int K0 = COLOR ? 16 : 0;
    WStream ws;
    ws.g = chunks; ws.ring = ldsr; ws.k = 0; ws.wave = wave; ws.lane = lane;
    if (COLOR) { ws.e0 = 16; ws.b0 = XR_SR8F_CHUNK0; ws.b1 = XR_SI_CHUNK0 + 16; }
    else ws.b0 = XR_SI_CHUNK0 + 16;
    float rb[4][8];                                               // featbar / zeta_l of operand k-steps (set = k-step & 3)
    const auto side = [&](int kk, int t, int sloc) {
        if (t != 1 && t != 4 && t != 2) return;
        if (kk < K0) {                                            // SR8F: the operand is featbar itself (registers only)
            if (t == 2) ld8(rb[sloc & 3], Frow + 16 * kk);
            return;
        }
        const int r = kk - K0;
        if (r >= 9 * 16) return;
        const int gi = r >> 4, s = r & 15, layer = gi <= 3 ? 7 - gi : 8 - gi;
        if (t == 2) { ld8(rb[sloc & 3], Zrow + (size_t)layer * lstride + 16 * s); return; }
        const int piece = t == 4;
        const float* srcp = Srow + (size_t)layer * lstride + 16 * s + 8 * piece;                 // s_{layer+1} = softplus(z_layer)
        unsigned char* dst = sring + (((kk & (XR_RING - 1)) * 4 + wave) * 2 + piece) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcp, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // the loads of the first two k-steps (the weight pipeline's start() stages its own): issued before it, landed at its barrier
    side(0, 1, 0); side(0, 4, 0); side(0, 2, 0); side(1, 1, 1); side(1, 4, 1); side(1, 2, 1);
    ws.start();

    f32x16 P[8], C[8];
    zero8(C);
    if (COLOR) {       // adjoint of s_8 from the feature rows: W_8[1:, :]^T featbar
        gemm_r<16, 2, true>(C, ws, [&](int s, int j) -> float { return colored ? rb[s & 3][j] : 0.f; }, side);
    }
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) P[b][r] = C[b][r] + sb * w8L[32 * b + 8 * (r >> 2) + 4 * hi + (r & 3)];      // + sdfbar W_8[0, :]
    int kb = 0;
    int lsave = 7;
    const auto zb_val = [&](int s, int j) -> float {             // zbar_l = phi'(z_l) sbar_{l+1} + zeta_l
        const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
        const float sv = reinterpret_cast<const float*>(sring + ((((kb + s) & (XR_RING - 1)) * 4 + wave) * 2 + (j >> 2)) * 1024)[lane * 4 + i];
        return fmaf(dphi_from_s(sv), P[b][4 * q + i], rb[s & 3][j]);
    };
    const auto zsink = [&](int s, const float (&v)[8]) { st_kstep(Zrow + (size_t)lsave * lstride, s, v); };
    f32x16 E[8];                                                 // adjoint of the encoding input (blocks 0, 1): skip part + layer 0
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) E[b][r] = 0.f;
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
        lsave = l;
        if (l == 4) {      // encoding part of the skip layer's input adjoint first (same operand zbar_4; only the second GEMM stores it:
            kb = ws.k;     // the first one's operand build still needs zeta_4 where zbar_4 goes)
            gemm_r<16, 0, true>(E, ws, zb_val, side);
        }
        kb = ws.k;
        zero8(C);
        gemm_rs<16, 2, true, 2>(C, ws, zb_val, side, zsink);
        copy8_acc(P, C);
    }
    kb = ws.k;
    lsave = 0;
    gemm_rs<16, 0, true, 2>(E, ws, zb_val, side, zsink);            // += W_0^T zbar_0
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * b + 8 * (r >> 2) + 4 * hi + (r & 3);
            if (f < 39) erow[f] = E[b][r];
        }
    if (hi == 0) {      // (same wave: the LDS writes above are ordered before these reads)
        const float* AE = ADJEPS + (size_t)point * 64;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float xj = ws_xc[(size_t)point * 3 + j];
            const float gb = (point < M ? gbar_main[(size_t)point * 3 + j] : 0.f) + (colored ? gbar_color[(size_t)point * 3 + j] : 0.f);
            float gv = erow[j], h = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float f = (float)(1 << i);
                float s, co;
                sincosf(xj * f, &s, &co);
                gv += f * (erow[enc_index(3, i, 0, j)] * co - erow[enc_index(3, i, 1, j)] * s);
                // second-order encoding term: sum_k adj_eps[k] d2 enc_k / d x_j^2 gbar_c[j]
                h -= f * f * (AE[enc_index(3, i, 0, j)] * s + AE[enc_index(3, i, 1, j)] * co);
            }
            xcbar[(size_t)point * 3 + j] = gv + h * gb + (colored ? xcbar_color[(size_t)point * 3 + j] : 0.f);
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------------------
static int train_attrs() {
    static DeviceOnce attr_done;
    if (attr_done.first()) {
        if (int e = allow_big_lds(k_deform_tan_x3r, XT_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_deform_bwd_x3r, XT_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_deform_bwd_x3r_tail, XT_LDS_BYTES > LEAN_LDS_BYTES ? XT_LDS_BYTES : LEAN_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_sdf_tan_x3r, XSB_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_sdf_rev_x3r<true>, XSB_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_sdf_rev_x3r<false>, XSB_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_color_bwd_x3r<true>, XCB_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_color_bwd_x3r<false>, XCB_LDS_BYTES)) return e;
        attr_done.done();
    }
    return ST_OK;
}
// tangent sweep of the points [0, m_rows) (a multiple of 128; 0: all Mp) (J gbar_o -> WS_JU, tau_0 .. tau_8 -> WS_D_T0 / WS_D_T)
int deform_tan_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, const float* d_go, hipStream_t st, int m_rows) {
    if (int e = train_attrs()) return e;
    const Tabs tb = make_tabs();
    ScopedTimer tm(KID_DEFORM_TAN_X3, m_rows > 0 ? m_rows : src.M, st);
    hipLaunchKernelGGL(k_deform_tan_x3r, dim3(((m_rows > 0 ? m_rows : L.Mp) + 127) / 128), dim3(XR_THREADS), XT_LDS_BYTES, st, src, tb, reinterpret_cast<const u32x4*>(packed_r), weff,
                       d_go, reinterpret_cast<const u32x4*>(ws + L.off[WS_D_MASK]), ws + L.off[WS_D_T0], ws + L.off[WS_D_T], ws + L.off[WS_JU], L.Mp);
    return hip_last("deform_tan_x3r");
}
// the main batch [0, m_main) through this family's reverse sweep with the tail's fp32 tangent + SDF-backward bodies at the head of the launch
int deform_bwd_x3r_with_tail(const BwdArgs& ba, const void* packed_r, int m_main, hipStream_t st) {
    if (int e = train_attrs()) return e;
    ScopedTimer tm(KID_DEFORM_BWD_X3, ba.src.M, st);
    const int n0 = (ba.L.Mp - m_main) / TM;
    hipLaunchKernelGGL(k_deform_bwd_x3r_tail, dim3(n0 + m_main / 64), dim3(XR_THREADS), XT_LDS_BYTES > LEAN_LDS_BYTES ? XT_LDS_BYTES : LEAN_LDS_BYTES, st, ba, n0,
                       m_main / TM, reinterpret_cast<const u32x4*>(packed_r), reinterpret_cast<const u32x4*>(ba.ws + ba.L.off[WS_D_MASK]));
    return hip_last("deform_bwd_x3r_with_tail");
}
// reverse sweep of all Mp points; vbar is read for the points [0, m_color)
int deform_bwd_x3r(const void* packed_r, const float* weff, float* ws, const WsLayout& L, int M, int m_color, hipStream_t st) {
    if (int e = train_attrs()) return e;
    const Tabs tb = make_tabs();
    ScopedTimer tm(KID_DEFORM_BWD_X3, M, st);
    hipLaunchKernelGGL(k_deform_bwd_x3r, dim3(L.Mp / 64), dim3(XR_THREADS), XT_LDS_BYTES, st, tb, reinterpret_cast<const u32x4*>(packed_r), weff,
                       ws + L.off[WS_XCBAR], ws + L.off[WS_VBAR_C], m_color, reinterpret_cast<const u32x4*>(ws + L.off[WS_D_MASK]),
                       ws + L.off[WS_D_A], ws + L.off[WS_D_A8], L.Mp);
    return hip_last("deform_bwd_x3r");
}

// SDF network backward of all Mp points (tangent sweep, then reverse sweep); d_go is the adjoint of g_o = g_c without a deformation network
int sdf_bwd_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, bool deform, bool color, int m_color,
                const float* d_sdf, const float* d_go, hipStream_t st) {
    if (int e = train_attrs()) return e;
    const Tabs tb = make_tabs();
    ScopedTimer tm(KID_SDF_BWD_X3, src.M, st);
    const dim3 grid(L.Mp / 128), block(XR_THREADS);
    const u32x4* pk = reinterpret_cast<const u32x4*>(packed_r);
    const float* gmain = deform ? ws + L.off[WS_JU] : d_go;
    const int Mmain = deform ? L.Mp : src.M;                      // WS_JU has a row for every workspace row, d_go only M
    const int mc = color ? m_color : 0;
    hipLaunchKernelGGL(k_sdf_tan_x3r, grid, block, XSB_LDS_BYTES, st, src, tb, pk, ws + L.off[WS_XC], gmain, ws + L.off[WS_GCBAR_C], mc, Mmain,
                       ws + L.off[WS_S_ACT], ws + L.off[WS_S_RHO], ws + L.off[WS_S_TAU0], ws + L.off[WS_S_TAU], ws + L.off[WS_S_ZB], L.Mp);
#define ES_LAUNCH_SREV(Cc) hipLaunchKernelGGL(k_sdf_rev_x3r<Cc>, grid, block, XSB_LDS_BYTES, st, tb, pk, weff, ws + L.off[WS_XC], d_sdf, src.M, gmain,            \
        ws + L.off[WS_GCBAR_C], ws + L.off[WS_XCBAR_C], ws + L.off[WS_FEATBAR], mc, ws + L.off[WS_S_ACT], ws + L.off[WS_S_ZB], ws + L.off[WS_S_ADJEPS], \
        ws + L.off[WS_XCBAR], L.Mp)
    if (color) ES_LAUNCH_SREV(true); else ES_LAUNCH_SREV(false);
#undef ES_LAUNCH_SREV
    return hip_last("sdf_bwd_x3r");
}

// reverse sweep of the colour network over the whole 128-point blocks that cover [0, m_color)
int color_bwd_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, bool deform, int m_color, const float* d_rgb,
                  hipStream_t st) {
    if (int e = train_attrs()) return e;
    if (m_color <= 0) return ST_OK;
    const Tabs tb = make_tabs();
    ScopedTimer tm(KID_COLOR_BWD_X3, m_color, st);
    const dim3 grid((m_color + 127) / 128), block(XR_THREADS);
    const u32x4* pk = reinterpret_cast<const u32x4*>(packed_r);
#define ES_LAUNCH_CBWD(D) hipLaunchKernelGGL(k_color_bwd_x3r<D>, grid, block, XCB_LDS_BYTES, st, src, tb, pk, weff, d_rgb, m_color, ws + L.off[WS_RGB], \
        ws + L.off[WS_XC], ws + L.off[WS_V], reinterpret_cast<const u32x4*>(ws + L.off[WS_C_MASK]), ws + L.off[WS_C_Y], ws + L.off[WS_C_Y8],       \
        ws + L.off[WS_FEATBAR], ws + L.off[WS_XCBAR_C], ws + L.off[WS_GCBAR_C], ws + L.off[WS_VBAR_C], L.Mp)
    if (deform) ES_LAUNCH_CBWD(true); else ES_LAUNCH_CBWD(false);
#undef ES_LAUNCH_CBWD
    return hip_last("color_bwd_x3r");
}

}  // namespace es
