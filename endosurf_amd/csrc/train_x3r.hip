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

// ---- host ------------------------------------------------------------------------------------------------------------------------
static int train_attrs() {
    static DeviceOnce attr_done;
    if (attr_done.first()) {
        if (int e = allow_big_lds(k_deform_tan_x3r, XT_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_deform_bwd_x3r, XT_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_deform_bwd_x3r_tail, XT_LDS_BYTES > LEAN_LDS_BYTES ? XT_LDS_BYTES : LEAN_LDS_BYTES)) return e;
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
