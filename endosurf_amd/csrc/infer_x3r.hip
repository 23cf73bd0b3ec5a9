// Opt-in split-precision INFERENCE chain on the register-resident GEMM core (x3r_core.h; formulation and arithmetic in query_x3r.hip /
// query_x3.hip).  The kernels replace the fp32 launches of point_fwd.hip when a no-grad point evaluation is requested with PF_X3 (no
// PF_SAVE); they read and write the same workspace buffers (k_sdf_fwd_x3r and k_color_fwd_x3r are described where they are defined):
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
#include "point_fwd_bodies.h"

namespace es {

constexpr int XI_LDS_BYTES = XR_RING * XR_CHUNK_BYTES + (128 * XR_ENC_LD + 8 * 256 + 3 * 256 + 4) * 4 + XR_TILE_BYTES;      // + the save tiles (RowTile)
static_assert(XI_LDS_BYTES <= 160 * 1024, "LDS carve");

// ---- value + tangent ------------------------------------------------------------------------------------------------------------
// SAVE (training): the layer inputs u_0 (encoding rows) and u_1 .. u_8 go to WS_D_U0 / WS_D_U, row 2 p = value, row 2 p + 1 = tangent
// of point p, row-major -- the X operands of the weight-gradient GEMMs (wgrad.hip), same layout as point_fwd.hip's deform_fwd_tile.
template <bool SAVE>
__device__ __forceinline__ void deform_jvp_x3r_body(const PointSrc& src, const Tabs& tb, const u32x4* __restrict__ chunks, const float* __restrict__ weff,
                                                    float* __restrict__ ws_xc, float* __restrict__ ws_v, u32x4* __restrict__ masks,
                                                    float* __restrict__ U0, float* __restrict__ U, int Mp, const int blk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsr[];
    float* encs = reinterpret_cast<float*>(ldsr + XR_RING * XR_CHUNK_BYTES);      // [128 columns][68]
    float* biasL = encs + 128 * XR_ENC_LD;                                         // [8 layers][256]
    float* w8L = biasL + 8 * 256;                                                  // [3][256] last-layer rows, then its 3 biases
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const bool tan = n >= 16;                                 // tangent column of point (n & 15)
    const int point = blk * 64 + wave * 16 + (n & 15);
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
    const size_t urow = (size_t)point * 2 + (tan ? 1 : 0);     // this lane's row of the 2-rows-per-point stacks
    const size_t rows2 = (size_t)Mp * 2;
    const RowTile rt{w8L + 3 * 256 + 4 + wave * 32 * XR_TILE_LD, 2 * (n & 15) + (tan ? 1 : 0), hi, lane};
    const size_t wrow0 = ((size_t)blk * 64 + wave * 16) * 2;             // first of the wave's 32 consecutive rows
    if (SAVE) {
#pragma unroll
        for (int k = 0; k < 32; k += 4) st4(U0 + urow * 64 + 32 * hi + k, erow[32 * hi + k], erow[32 * hi + k + 1], erow[32 * hi + k + 2], erow[32 * hi + k + 3]);
    }
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
    copy8_acc(P, C);
    const size_t mrow = ((size_t)point) * 2 + hi;
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        const bool skip = l == 4;                          // IDR skip: input of layer 4 = [h(204) | enc(52)] (1/sqrt2 folded into W4)
        u32x4 mk = {0u, 0u, 0u, 0u};
        init(C, l);
        float* Ul = U + ((size_t)(l - 1) * rows2 + wrow0) * 256;              // u_l = this GEMM's operand
        gemm_rs<16, 2, false, (SAVE ? 4 : 0)>(C, ws, [&](int s, int j) -> float {
            const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
            const int f = 32 * b + 8 * q + 4 * hi + i;
            const float z = P[b][4 * q + i];
            const bool m = value_row(z) > 0.f;               // ReLU mask of the VALUE column gates both
            mask_set_now(mk, b, 4 * q + i, m);
            const float h = m ? z : 0.f;
            if (32 * b + 8 * q + 4 + i < 204) return h;
            return (skip && f >= 204) ? erow[f - 204] : h;
        }, NoSide(), [&](int s, const float (&v)[8]) { if (SAVE) rt.put<256>(s, v, Ul); });
        if (!tan && point < Mp) masks[((size_t)(l - 1) * Mp) * 2 + mrow] = mk;
        copy8_acc(P, C);
    }
    {   // x_c = x + W8 relu(z_7) + b8 on the value columns, v = d + W8 (mask_7 . tau_7) on the tangent columns
        u32x4 mk = {0u, 0u, 0u, 0u};
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
        float* U8 = U + ((size_t)7 * rows2 + urow) * 256 + 4 * hi;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float h4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * q + i, f = 32 * b + 8 * q + 4 * hi + i;
                    const float z = P[b][r];
                    const bool m = value_row(z) > 0.f;
                    mask_set_now(mk, b, r, m);
                    const float h = m ? z : 0.f;
                    h4[i] = h;
                    d0 = fmaf(w8L[f], h, d0); d1 = fmaf(w8L[256 + f], h, d1); d2 = fmaf(w8L[512 + f], h, d2);
                }
                if (SAVE) st4(U8 + 32 * b + 8 * q, h4[0], h4[1], h4[2], h4[3]);
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

template <bool SAVE>
__global__ __launch_bounds__(XR_THREADS, 1) void k_deform_jvp_x3r(PointSrc src, Tabs tb, const u32x4* __restrict__ chunks, const float* __restrict__ weff,
                                                                float* __restrict__ ws_xc, float* __restrict__ ws_v, u32x4* __restrict__ masks,
                                                                float* __restrict__ U0, float* __restrict__ U, int Mp) {
    deform_jvp_x3r_body<SAVE>(src, tb, chunks, weff, ws_xc, ws_v, masks, U0, U, Mp, (int)blockIdx.x);
}
// Training batch with a colour-less tail (the 3N auxiliary points of a training step behind the ray samples): the main batch fills the
// chip in whole rounds of this family's blocks, so the tail's few blocks would add a nearly empty round to EVERY launch.  As in the fp32
// family (point_fwd.hip) the tail's dependent SDF + VJP stages ride at the head of the launch with the most and shortest rounds -- this
// one (4 rounds of 64-point blocks at 1024 rays x 64 samples) -- as fp32 tile bodies (blocks [0, n0): tiles t0 ..; their deformation
// stage ran as a small fp32 launch before): the tail costs one short round here instead of one round of every kernel.
__global__ __launch_bounds__(XR_THREADS, 1) void k_deform_jvp_x3r_tail(FwdArgs fa, int n0, int t0, const u32x4* __restrict__ chunks,
                                                                     u32x4* __restrict__ masks) {
    if ((int)blockIdx.x < n0) {
        // (taking the tail's deformation stage into this workgroup as well makes its ~1.0 ms the launch's critical path -- the main
        // part needs 0.78 ms: measured 1.01 ms against 0.77 + 0.165 ms for the deformation stage as a launch of its own)
        sdf_fwd_tile(fa, t0 + (int)blockIdx.x);
        __syncthreads();
        deform_vjp_tile(fa, t0 + (int)blockIdx.x);
        return;
    }
    deform_jvp_x3r_body<true>(fa.src, fa.tb, chunks, fa.weff, fa.ws + fa.L.off[WS_XC], fa.ws + fa.L.off[WS_V], masks, fa.ws + fa.L.off[WS_D_U0],
                              fa.ws + fa.L.off[WS_D_U], fa.L.Mp, (int)blockIdx.x - n0);
}

// ---- reverse sweep ---------------------------------------------------------------------------------------------------------------
// SAVE (training): the adjoints r_7 .. r_0 of the pre-activations go to WS_D_R ([8][Mp][256] row-major): paired with the tangent sweep
// of the backward pass (tau_l, train_x3r.hip) they are the weight gradient of the g_o path.
template <bool SAVE>
__global__ __launch_bounds__(XR_THREADS, 1) void k_deform_vjp_x3r(PointSrc src, Tabs tb, const u32x4* __restrict__ chunks, const float* __restrict__ weff,
                                                                const float* __restrict__ ws_gc, float* __restrict__ ws_go,
                                                                const u32x4* __restrict__ masks, float* __restrict__ R, int Mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsr[];
    float* adj = reinterpret_cast<float*>(ldsr + XR_RING * XR_CHUNK_BYTES);       // [128 points][68]: adjoint of the 52 encoding inputs
    float* w8L = adj + 128 * XR_ENC_LD + 8 * 256;                                  // [3][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int point = blockIdx.x * 128 + wave * 32 + n;
    constexpr bool live = true;                                // Mp is a multiple of 128 (workspace.h): every point of a block is a workspace row
    float* arow = adj + (wave * 32 + n) * XR_ENC_LD;
    // (g_c and, at the end, the point itself are (re)read where they are needed instead of living in registers across the sweep)
    float g[3] = {0.f, 0.f, 0.f};
    if (live) { g[0] = ws_gc[(size_t)point * 3]; g[1] = ws_gc[(size_t)point * 3 + 1]; g[2] = ws_gc[(size_t)point * 3 + 2]; }
    for (int i = tid; i < 3 * 256; i += XR_THREADS) w8L[i] = weff[tb.woff[NET_D * LAYERS + 8] + i];
    const unsigned mrow = (unsigned)(live ? point : 0) * 2u + (unsigned)hi;      // 32-bit lane index against a wave-uniform layer base
    const size_t mstride = (size_t)Mp * 2;
    u32x4 mk = (masks + 7 * mstride)[mrow];
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
    float* Rrow = R + (size_t)(live ? point : 0) * 256 + 4 * hi;       // + l Mp 256: r_l of this lane's point
    const size_t rstride = (size_t)Mp * 256;
    int lsave = 7;
    const RowTile rt{w8L + 3 * 256 + 4 + wave * 32 * XR_TILE_LD, n, hi, lane};
    float* Rwave = R + ((size_t)blockIdx.x * 128 + wave * 32) * 256;       // the wave's 32 rows (Mp is a multiple of 128: all of them exist)
    const auto rsink = [&](int s, const float (&v)[8]) { if (SAVE) rt.put<256>(s, v, Rwave + lsave * rstride); };
    gemm_rs<16, 2, false, (SAVE ? 4 : 0)>(C, ws, [&](int s, int j) -> float {
        const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
        const int f = 32 * b + 8 * q + 4 * hi + i;
        const float v = fmaf(w8L[f], g[0], fmaf(w8L[256 + f], g[1], w8L[512 + f] * g[2]));
        return mask_get(mk, b, 4 * q + i) ? v : 0.f;
    }, NoSide(), rsink);
    copy8_acc(P, C);
    // layers 6 .. 1: r_l = mask_l . (adjoint of h_l),  adjoint of h_{l-1} = W_l^T r_l
#pragma unroll 1
    for (int l = 6; l >= 1; --l) {
        mk = (masks + l * mstride)[mrow];
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
        lsave = l;
        if (l == 3) {                           // 204 outputs of layer 3: 14 k-steps (the rest is zero padding in DR3)
            gemm_rs<14, 2, false, (SAVE ? 4 : 0)>(C, ws, val, NoSide(), rsink);
            if (SAVE) {                         // features 224 .. 255 of r_3 are zero like 204 .. 223
                const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                st_kstep(Rrow + 3 * rstride, 14, z8); st_kstep(Rrow + 3 * rstride, 15, z8);
            }
        } else gemm_rs<16, 2, false, (SAVE ? 4 : 0)>(C, ws, val, NoSide(), rsink);
        copy8_acc(P, C);
    }
    // r_0 = mask_0 . (adjoint of h_0);  adjoint of the encoding += W_0^T r_0 (52 outputs: accumulator group 0 only)
    mk = masks[mrow];
    zero(C);
    lsave = 0;
    gemm_rs<16, 1, false, (SAVE ? 4 : 0)>(C, ws, [&](int s, int j) -> float {
        const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
        return mask_get(mk, b, 4 * q + i) ? P[b][4 * q + i] : 0.f;
    }, NoSide(), rsink);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * b + 8 * (r >> 2) + 4 * hi + (r & 3);
            if (f < 52) arow[f] += C[b][r];
        }
    // the two lane halves of a point wrote disjoint features of its row; same wave, so the LDS writes are ordered before the reads
    if (hi == 0 && live) {   // g_o[j] = g_c[j] + sum_k adj[k] * d enc_k / d x_j   (position part of the encoding, observed-space x)
        float x[3], t, d[3];
        load_point(src, point, x, t, d);
        g[0] = ws_gc[(size_t)point * 3]; g[1] = ws_gc[(size_t)point * 3 + 1]; g[2] = ws_gc[(size_t)point * 3 + 2];
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


// ---- SDF network: value pass + geometry features + reverse sweep ------------------------------------------------------------------
// SDFNetwork (endosurf.py:773-786) on x_c: sdf, the 256 geometry features (colour evaluations only) and the analytic reverse sweep
// g_c = d sdf / d x_c (get_sdf_grad_from_canonical_space, :603-619).  A wave owns 32 points.  The reverse sweep needs
// softplus'(z_l) of every layer: the layer inputs s_1 .. s_7 (= softplus(z_0 .. z_6), the operands this kernel builds anyway) go to
// WS_S_ACT as ROW-MAJOR [8][Mp][256] stacks while they are built and come back through a SIDE stream of the weight pipeline -- 2 direct
// loads of 16 B per lane and k-step (per-lane source addresses: this lane's row) into their own ring of four k-steps, same barriers;
// softplus' = 1 - exp(-100 s) as in the fp32 kernels; z_7 is still in registers when the sweep starts.
// Inference only: the training chain keeps the SDF network on the fp32 kernels (point_fwd.hip / point_bwd.hip).  A saving variant of this
// kernel and a split-precision SDF backward (tangent + reverse sweep as two kernels) existed in rounds 3-4 behind a default-off switch,
// parity-tested and SLOWER than the fp32 kernels (2.24 ms against 1.60 for the backward): the SDF backward moves 6-7 KB per layer and
// point and a lane-per-row kernel issues that traffic as 16-B pieces (2.0-2.5 TB/s) where the fp32 kernels' fragment-ordered tiles
// reach 3 TB/s; the LDS tiles that would coalesce two more streams do not fit next to the weight ring.  Removed in round 4.
constexpr int XS_ENC_LD = 41;       // floats per point row of the encoding / its adjoint (40 used; odd: conflict-free)
constexpr int XS_ZRING_BYTES = XR_RING * 4 * 2048;
constexpr int XS_LDS_BYTES = XR_RING * XR_CHUNK_BYTES + XS_ZRING_BYTES + (128 * XS_ENC_LD + 9 * 256 + 256 + 4) * 4;
static_assert(XS_LDS_BYTES <= 160 * 1024, "LDS carve");
__device__ __forceinline__ float sigmoid100(float z) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-144.26950408889634f * z)); }
template <bool DEFORM, bool COLOR>
__global__ __launch_bounds__(XR_THREADS, 1) void k_sdf_fwd_x3r(PointSrc src, Tabs tb, const u32x4* __restrict__ chunks, const float* __restrict__ weff,
                                                             float* __restrict__ ws_xc, float* __restrict__ ws_sdf, float* __restrict__ ws_feat,
                                                             float* __restrict__ ws_gc, float* __restrict__ ws_go, float* __restrict__ SACT, int Mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsr[];
    unsigned char* zring = ldsr + XR_RING * XR_CHUNK_BYTES;
    float* encs = reinterpret_cast<float*>(zring + XS_ZRING_BYTES);                // [128 points][41]: enc6(x_c), later its adjoint
    float* biasL = encs + 128 * XS_ENC_LD;                                         // [9][256]: layers 0..7, feature rows of layer 8
    float* w8L = biasL + 9 * 256;                                                  // [256] row 0 of layer 8, then its bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int point = blockIdx.x * 128 + wave * 32 + n;                            // a workspace row (Mp is a multiple of 128)
    float* erow = encs + (wave * 32 + n) * XS_ENC_LD;
    float x[3];
    if (DEFORM) {
        const float* xc = ws_xc + (size_t)point * 3;
        x[0] = xc[0]; x[1] = xc[1]; x[2] = xc[2];
    } else {
        float t, d[3];
        load_point(src, point, x, t, d);
        if (hi == 0) { ws_xc[(size_t)point * 3] = x[0]; ws_xc[(size_t)point * 3 + 1] = x[1]; ws_xc[(size_t)point * 3 + 2] = x[2]; }
    }
    for (int i = tid; i < 9 * 256; i += XR_THREADS) {
        const int l = i >> 8, f = i & 255;
        biasL[i] = l < 8 ? weff[tb.boff[NET_S * LAYERS + l] + f] : weff[tb.boff[NET_S * LAYERS + 8] + 1 + f];
    }
    for (int i = tid; i < 256; i += XR_THREADS) w8L[i] = weff[tb.woff[NET_S * LAYERS + 8] + i];
    if (tid == 0) w8L[256] = weff[tb.boff[NET_S * LAYERS + 8]];
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) {
        const int i = 3 * hi + ii;
        const float f = (float)(1 << i);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s, co;
            sincosf(x[c] * f, &s, &co);
            erow[enc_index(3, i, 0, c)] = s;
            erow[enc_index(3, i, 1, c)] = co;
        }
    }
    if (hi == 0) { erow[0] = x[0]; erow[1] = x[1]; erow[2] = x[2]; erow[39] = 0.f; }
    __syncthreads();
    const size_t lstride = (size_t)Mp * 256;
    // logical k-steps: [0, 120) SF0 .. SF7, then (colour: SF8F,) SR7, SR6, SR5, SR4A, SR4M, SR3, SR2, SR1, SR0
    constexpr int KR0 = XR_SDF_FWD_CHUNKS + (COLOR ? 16 : 0);                       // first k-step of the reverse sweep
    WStream ws;
    ws.g = chunks; ws.ring = ldsr; ws.k = 0; ws.wave = wave; ws.lane = lane;
    ws.e0 = XR_SDF_FWD_CHUNKS; ws.b0 = XR_SDF_CHUNK0; ws.b1 = XR_SI_CHUNK0 + (COLOR ? 0 : 16);
    ws.start();

    // per-lane rows of the [..][Mp][256] stacks as a block-uniform base + ONE 32-bit lane offset (64-bit per-lane pointers of four
    // streams were what this kernel spilled)
    const size_t blk0 = (size_t)blockIdx.x * 128;
    const unsigned loff = (unsigned)(wave * 32 + n) * 256u + 4u * (unsigned)hi;     // + layer Mp 256 + 16 s (+ 8): this lane's pieces
    float* Sblk = SACT + blk0 * 256;
    // the side stream: the layer input whose softplus' the reverse GEMM of k-step kk needs
    const auto side = [&](int kk, int t, int) {
        if (t != 1 && t != 4) return;
        const int r = kk - (KR0 + 16);                       // SR7 reads z_7 from registers
        if (r < 0 || r >= 8 * 16) return;
        const int gi = r >> 4, s = r & 15, layer = gi <= 2 ? 6 - gi : 7 - gi, piece = t == 4;     // SR6 SR5 SR4A SR4M SR3 SR2 SR1 SR0
        const float* srcp = (Sblk + (size_t)layer * lstride) + (loff + (unsigned)(16 * s + 8 * piece));      // s_{layer+1} = softplus(z_layer)
        unsigned char* dst = zring + (((kk & (XR_RING - 1)) * 4 + wave) * 2 + piece) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcp, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    const auto enc_val = [&](int s, int j) -> float {
        const int k = 16 * s + xr_kperm(hi, j);
        return k < 39 ? erow[k] : 0.f;
    };
    f32x16 P[8], C[8];
    init8(C, biasL, hi);
    gemm_r<4>(C, ws, enc_val, side);
    const auto act_val = [&](int s, int j) -> float {
        const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
        return softplus100_native(P[b][4 * q + i]);
    };
    copy8_acc(P, C);
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        init8(C, biasL + l * 256, hi);
        float* Sl = (Sblk + (size_t)(l - 1) * lstride) + loff;      // s_l = this GEMM's operand
        gemm_rs<16, 2, false, 2>(C, ws, act_val, side, [&](int s, const float (&v)[8]) { st_kstep(Sl, s, v); });
        if (l == 4) gemm_r<4>(C, ws, enc_val, side);           // NeRF skip: + encoding part (SF4A follows SF4M)
        copy8_acc(P, C);
    }
    // P = z_7
    if (COLOR) {       // geometry features = rows 1 .. 256 of the last layer
        init8(C, biasL + 8 * 256, hi);
        gemm_r<16, 2, false>(C, ws, act_val, side);
        float* fo = (ws_feat + blk0 * 256) + loff;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(fo + 32 * b + 8 * q) = make_float4(C[b][4 * q], C[b][4 * q + 1], C[b][4 * q + 2], C[b][4 * q + 3]);
    }
    {
        float s0 = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) s0 = fmaf(w8L[32 * b + 8 * q + 4 * hi + i], softplus100_native(P[b][4 * q + i]), s0);
        s0 += __shfl_xor(s0, 32);
        if (hi == 0) ws_sdf[point] = s0 + w8L[256];
    }
    // ---- reverse sweep: rho_l = softplus'(z_l) . (adjoint of s_{l+1}),  adjoint of s_l = W_l^T rho_l ----
    const auto zero = [&](f32x16(&A)[8], int nb) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (b < nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) A[b][r] = 0.f;
    };
    zero(C, 8);
    gemm_r<16, 2, false>(C, ws, [&](int s, int j) -> float {              // rho_7 = softplus'(z_7) . W8[0, :]
        const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
        return sigmoid100(P[b][4 * q + i]) * w8L[32 * b + 8 * q + 4 * hi + i];
    }, side);
    copy8_acc(P, C);
    int kb = 0;                                                  // first k-step of the running GEMM (its side-stream slots)
    const auto rho_val = [&](int s, int j) -> float {
        const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
        const float sv = reinterpret_cast<const float*>(zring + ((((kb + s) & (XR_RING - 1)) * 4 + wave) * 2 + (j >> 2)) * 1024)[lane * 4 + i];
        return dphi_from_s(sv) * P[b][4 * q + i];
    };
    // adjoint of the encoding input (accumulator blocks 0, 1): the skip layer's part is PARKED in this point's LDS row (the encoding
    // itself is dead once the forward pass is through) until layer 0's arrives -- kept in accumulators across layers 3 .. 1 it would be
    // a third live array next to P and C (288 of the 256 accumulation registers: the spills of round 3)
    f32x16 E[8];
    const auto park = [&](bool add) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = 32 * b + 8 * (r >> 2) + 4 * hi + (r & 3);
                if (f < 39) erow[f] = add ? erow[f] + E[b][r] : E[b][r];
            }
    };
#pragma unroll 1
    for (int l = 6; l >= 1; --l) {
        if (l == 4) {                                            // encoding part of the skip layer's input adjoint, same operand rho_4
            kb = ws.k;
            zero(E, 2);
            gemm_r<16, 0, true>(E, ws, rho_val, side);
            park(false);
            kb = ws.k;
            zero(C, 8);
            gemm_r<16, 2, true>(C, ws, rho_val, side);
        } else {
            kb = ws.k;
            zero(C, 8);
            gemm_r<16, 2, true>(C, ws, rho_val, side);
        }
        copy8_acc(P, C);
    }
    kb = ws.k;
    zero(E, 2);
    gemm_r<16, 0, true>(E, ws, rho_val, side);           // W_0^T rho_0
    park(true);
    // g_c[j] = sum_k adj[k] * d enc_k / d x_j   (a point's row is written and read by its own wave only: LDS operations of a wave are ordered)
    if (hi == 0) {
        // (x_c is read again here instead of living in three registers across both sweeps: ws_xc holds it -- written by the
        // deformation kernel, or by this lane at the top of the kernel)
        x[0] = ws_xc[(size_t)point * 3]; x[1] = ws_xc[(size_t)point * 3 + 1]; x[2] = ws_xc[(size_t)point * 3 + 2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float gv = erow[j];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float f = (float)(1 << i);
                float s, co;
                sincosf(x[j] * f, &s, &co);
                gv += f * (erow[enc_index(3, i, 0, j)] * co - erow[enc_index(3, i, 1, j)] * s);
            }
            ws_gc[(size_t)point * 3 + j] = gv;
            if (!DEFORM) ws_go[(size_t)point * 3 + j] = gv;      // with a deformation network g_o = J^T g_c is the VJP sweep
        }
    }
}


// ---- colour network -----------------------------------------------------------------------------------------------------------------
// ColorNetwork (endosurf.py:828-842) on [enc10(x_c) 63 | g_c 3 | enc4(d_c) 27 | feat 256] -> sigmoid rgb; d_c = J d / (|J d| + 1e-10)
// (:684-685).  A wave owns 32 points.  The 93-wide small part of the input is evaluated where it is consumed (layer 0 and the skip
// layer): every lane computes exactly the 8 operand elements of its half of a k-step.  The geometry features come back from the
// row-major WS_FEAT as a side stream of the weight pipeline (direct loads with per-lane source addresses: 2 x 16 B per lane and k-step).
constexpr int XC_FRING_BYTES = XR_RING * 4 * 2048;
constexpr int XC_LDS_BYTES = XR_RING * XR_CHUNK_BYTES + XC_FRING_BYTES + (8 * 256 + 3 * 256 + 4) * 4 + XR_TILE_BYTES;      // + the save tiles (RowTile)
constexpr int XC_K_0S = 0, XC_K_0F = 6, XC_K_4S = 6 + 16 + 48 + 16, XC_K_4F = XC_K_4S + 6;     // logical k-steps of CF0S, CF0F, CF4S, CF4F
static_assert(xr_kg(35) == 6 && xr_kg(41) == 6 && XC_K_4F == 92, "colour stream layout");

// SAVE (training; launched over whole 128-point blocks, every point of which is a workspace row): the small part of the input goes to
// WS_C_IN ([Mp][128], 96 written), the layer inputs h_1 .. h_8 to WS_C_H (row-major, the X operands of the weight-gradient GEMMs) and
// the ReLU masks to WS_C_MASK in this family's layout (128 bits per (layer, point, lane half), read by k_color_bwd_x3r).
template <bool DEFORM, bool SAVE>
__global__ __launch_bounds__(XR_THREADS, 1) void k_color_fwd_x3r(PointSrc src, Tabs tb, const u32x4* __restrict__ chunks, const float* __restrict__ weff,
                                                               const float* __restrict__ ws_xc, const float* __restrict__ ws_v,
                                                               const float* __restrict__ ws_gc, const float* __restrict__ ws_feat,
                                                               float* __restrict__ ws_rgb, float* __restrict__ CIN, float* __restrict__ CH,
                                                               u32x4* __restrict__ masks, int Mcp, int Mp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsr[];
    unsigned char* fring = ldsr + XR_RING * XR_CHUNK_BYTES;
    float* biasL = reinterpret_cast<float*>(fring + XC_FRING_BYTES);               // [8][256]
    float* w8L = biasL + 8 * 256;                                                  // [3][256] last-layer rows, then its 3 biases
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int point = blockIdx.x * 128 + wave * 32 + n;
    const bool live = SAVE || point < Mcp;
    const size_t pl = live ? point : 0;
    float xc[3], gc[3], dc[3];
    // (read again before the skip layer instead of living in nine registers across layers 1 .. 3)
    const auto load_small = [&]() {
        float x[3], t, d[3];
        load_point(src, (int)pl, x, t, d);
#pragma unroll
        for (int c = 0; c < 3; ++c) { xc[c] = ws_xc[pl * 3 + c]; gc[c] = ws_gc[pl * 3 + c]; dc[c] = DEFORM ? ws_v[pl * 3 + c] : d[c]; }
        const float inv = 1.f / (sqrtf(dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2]) + 1e-10f);
        dc[0] *= inv; dc[1] *= inv; dc[2] *= inv;
    };
    load_small();
    for (int i = tid; i < 8 * 256; i += XR_THREADS) biasL[i] = weff[tb.boff[NET_C * LAYERS + (i >> 8)] + (i & 255)];
    for (int i = tid; i < 3 * 256; i += XR_THREADS) w8L[i] = weff[tb.woff[NET_C * LAYERS + 8] + i];
    if (tid < 3) w8L[3 * 256 + tid] = weff[tb.boff[NET_C * LAYERS + 8] + tid];
    __syncthreads();
    WStream ws;
    ws.g = chunks; ws.ring = ldsr; ws.k = 0; ws.wave = wave; ws.lane = lane; ws.b0 = XR_C_CHUNK0;
    ws.start();

    // per-lane addresses as a block-uniform base + a 32-bit lane offset (one register instead of a 64-bit pair per stream)
    const size_t blk0 = (size_t)blockIdx.x * 128;
    const unsigned lrow = (unsigned)(wave * 32 + n);       // (dead lanes of the last block read their own row: the buffers have Mp >= 128 gridDim rows)
    const float* fblk = ws_feat + blk0 * 256;
    const auto side = [&](int kk, int t, int) {                    // geometry features of k-step s of CF0F / CF4F
        if (t != 1 && t != 4) return;
        int s = kk - XC_K_0F;
        if (s < 0 || s >= 16) s = kk - XC_K_4F;
        if (s < 0 || s >= 16) return;
        const int piece = t == 4;
        const float* srcp = fblk + (lrow * 256u + (unsigned)(16 * s + 4 * hi + 8 * piece));      // features 32 b + 16 p + 4 hi (+ 8): the k order of the operand
        unsigned char* dst = fring + (((kk & (XR_RING - 1)) * 4 + wave) * 2 + piece) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcp, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    int kb = 0;
    const auto feat_val = [&](int s, int j) -> float {
        return reinterpret_cast<const float*>(fring + ((((kb + s) & (XR_RING - 1)) * 4 + wave) * 2 + (j >> 2)) * 1024)[lane * 4 + (j & 3)];
    };
    // element k of [enc10(x_c) 63 | g_c 3 | enc4(d_c) 27 | 0 0 0]
    const auto small_val = [&](int s, int j) -> float {
        int k = 16 * s + xr_kperm(hi, j);
        if (k >= 93) return 0.f;
        if (k >= 63 && k < 66) return k == 63 ? gc[0] : (k == 64 ? gc[1] : gc[2]);
        const bool dir = k >= 66;
        if (dir) k -= 66;
        const float p0 = dir ? dc[0] : xc[0], p1 = dir ? dc[1] : xc[1], p2 = dir ? dc[2] : xc[2];
        if (k < 3) return k == 0 ? p0 : (k == 1 ? p1 : p2);
        const int i = (k - 3) / 6, rem = (k - 3) - 6 * i, fn = rem >= 3, c = rem - 3 * fn;
        float sn, co;
        sincosf((c == 0 ? p0 : (c == 1 ? p1 : p2)) * (float)(1 << i), &sn, &co);
        return fn ? co : sn;
    };
    f32x16 P[8], C[8];
    u32x4 mk = {0u, 0u, 0u, 0u};
    const auto relu_val = [&](int s, int j) -> float {
        const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
        const float z = P[b][4 * q + i];
        if (SAVE) mask_set(mk, b, 4 * q + i, z > 0.f);
        return fmaxf(z, 0.f);
    };
    const unsigned mrow = (unsigned)pl * 2u + (unsigned)hi;       // 32-bit lane index against a wave-uniform layer base
    const size_t mstride = (size_t)Mp * 2;
    const size_t hstride = (size_t)Mp * 256;
    const RowTile rt{w8L + 3 * 256 + 4 + wave * 32 * XR_TILE_LD, n, hi, lane};
    const size_t wrow0 = (size_t)blockIdx.x * 128 + wave * 32;      // the wave's 32 rows (SAVE: whole blocks of workspace rows)
    init8(C, biasL, hi);
    gemm_rs<6, 2, false, (SAVE ? 4 : 0)>(C, ws, small_val, side, [&](int s, const float (&v)[8]) { if (SAVE) rt.put<128>(s, v, CIN + wrow0 * 128); });
    kb = ws.k;
    gemm_r<16, 2, true>(C, ws, feat_val, side);
    copy8_acc(P, C);
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        init8(C, biasL + l * 256, hi);
        float* Hl = CH + (size_t)(l - 1) * hstride + wrow0 * 256;
        mk = u32x4{0u, 0u, 0u, 0u};
        gemm_rs<16, 2, false, (SAVE ? 4 : 0)>(C, ws, relu_val, side, [&](int s, const float (&v)[8]) { if (SAVE) rt.put<256>(s, v, Hl); });
        if (SAVE) (masks + (l - 1) * mstride)[mrow] = mk;
        if (l == 4) {       // skip layer: input = [h(256) | small(93) | feat(256)] / sqrt2
            load_small();
            gemm_r<6>(C, ws, small_val, side);
            kb = ws.k;
            gemm_r<16, 2, true>(C, ws, feat_val, side);
        }
        copy8_acc(P, C);
    }
    {
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
        mk = u32x4{0u, 0u, 0u, 0u};
        float* H8 = CH + (size_t)7 * hstride + blk0 * 256 + (lrow * 256u + 4u * (unsigned)hi);      // h_8 of this lane's point (SAVE: every lane is live)
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float h4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * q + i, f = 32 * b + 8 * q + 4 * hi + i;
                    const float h = fmaxf(P[b][r], 0.f);
                    if (SAVE) mask_set(mk, b, r, P[b][r] > 0.f);
                    h4[i] = h;
                    d0 = fmaf(w8L[f], h, d0); d1 = fmaf(w8L[256 + f], h, d1); d2 = fmaf(w8L[512 + f], h, d2);
                }
                if (SAVE) st4(H8 + 32 * b + 8 * q, h4[0], h4[1], h4[2], h4[3]);
            }
        if (SAVE) (masks + 7 * mstride)[mrow] = mk;
        d0 += __shfl_xor(d0, 32); d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32);
        if (hi == 0 && live) {
            float* o = ws_rgb + (size_t)point * 3;
            o[0] = 1.f / (1.f + expf(-(d0 + w8L[768])));
            o[1] = 1.f / (1.f + expf(-(d1 + w8L[769])));
            o[2] = 1.f / (1.f + expf(-(d2 + w8L[770])));
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------------------
static int infer_attrs() {
    static DeviceOnce attr_done;
    if (attr_done.first()) {
        if (int e = allow_big_lds(k_deform_jvp_x3r<false>, XI_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_deform_jvp_x3r<true>, XI_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_deform_jvp_x3r_tail, XI_LDS_BYTES > LEAN_LDS_BYTES ? XI_LDS_BYTES : LEAN_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_deform_vjp_x3r<false>, XI_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_deform_vjp_x3r<true>, XI_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_sdf_fwd_x3r<true, true>, XS_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_sdf_fwd_x3r<true, false>, XS_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_sdf_fwd_x3r<false, true>, XS_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_sdf_fwd_x3r<false, false>, XS_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_color_fwd_x3r<true, false>, XC_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_color_fwd_x3r<false, false>, XC_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_color_fwd_x3r<true, true>, XC_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_color_fwd_x3r<false, true>, XC_LDS_BYTES)) return e;
        attr_done.done();
    }
    return ST_OK;
}
// packed_r = the k-step-ordered split weights of pack_x3r (query_x3r.hip)
// save: the workspace was laid out with PF_SAVE and the kernel keeps what the weight-gradient GEMMs need (training)
int deform_jvp_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, bool save, hipStream_t st) {
    if (int e = infer_attrs()) return e;
    const Tabs tb = make_tabs();
    ScopedTimer tm(KID_DEFORM_FWD_X3, src.M, st);
    const dim3 grid(L.Mp / 64), block(XR_THREADS);
    const u32x4* pk = reinterpret_cast<const u32x4*>(packed_r);
    u32x4* mk = reinterpret_cast<u32x4*>(ws + L.off[WS_D_MASK]);
    if (save) hipLaunchKernelGGL(k_deform_jvp_x3r<true>, grid, block, XI_LDS_BYTES, st, src, tb, pk, weff, ws + L.off[WS_XC], ws + L.off[WS_V], mk,
                                 ws + L.off[WS_D_U0], ws + L.off[WS_D_U], L.Mp);
    else hipLaunchKernelGGL(k_deform_jvp_x3r<false>, grid, block, XI_LDS_BYTES, st, src, tb, pk, weff, ws + L.off[WS_XC], ws + L.off[WS_V], mk,
                            (float*)nullptr, (float*)nullptr, L.Mp);
    return hip_last("deform_jvp_x3r");
}
// the training batch's main part [0, m_main) through this family's value + tangent kernel, its colour-less tail's SDF + VJP stages as fp32
// tile bodies at the head of the same launch (k_deform_jvp_x3r_tail); m_main is a multiple of 128, the tail's deformation stage has run
int deform_jvp_x3r_with_tail(const FwdArgs& fa, const void* packed_r, int m_main, hipStream_t st) {
    if (int e = infer_attrs()) return e;
    ScopedTimer tm(KID_DEFORM_FWD_X3, fa.src.M, st);
    const int n0 = (fa.L.Mp - m_main) / TM;
    hipLaunchKernelGGL(k_deform_jvp_x3r_tail, dim3(n0 + m_main / 64), dim3(XR_THREADS), XI_LDS_BYTES > LEAN_LDS_BYTES ? XI_LDS_BYTES : LEAN_LDS_BYTES, st, fa, n0,
                       m_main / TM, reinterpret_cast<const u32x4*>(packed_r), reinterpret_cast<u32x4*>(fa.ws + fa.L.off[WS_D_MASK]));
    return hip_last("deform_jvp_x3r_with_tail");
}
// m_rows: the points [0, m_rows) (a multiple of 128; default all workspace rows)
int deform_vjp_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, bool save, hipStream_t st, int m_rows) {
    if (int e = infer_attrs()) return e;
    const Tabs tb = make_tabs();
    ScopedTimer tm(KID_DEFORM_VJP_X3, m_rows > 0 ? m_rows : src.M, st);
    const dim3 grid(((m_rows > 0 ? m_rows : L.Mp) + 127) / 128), block(XR_THREADS);
    const u32x4* pk = reinterpret_cast<const u32x4*>(packed_r);
    const u32x4* mk = reinterpret_cast<const u32x4*>(ws + L.off[WS_D_MASK]);
    if (save) hipLaunchKernelGGL(k_deform_vjp_x3r<true>, grid, block, XI_LDS_BYTES, st, src, tb, pk, weff, ws + L.off[WS_GC], ws + L.off[WS_GO], mk,
                                 ws + L.off[WS_D_R], L.Mp);
    else hipLaunchKernelGGL(k_deform_vjp_x3r<false>, grid, block, XI_LDS_BYTES, st, src, tb, pk, weff, ws + L.off[WS_GC], ws + L.off[WS_GO], mk,
                            (float*)nullptr, L.Mp);
    return hip_last("deform_vjp_x3r");
}

// all Mp points get sdf / g_c (/ g_o); the geometry features (color) are written for every point as well
int sdf_fwd_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, bool deform, bool color, hipStream_t st) {
    if (int e = infer_attrs()) return e;
    const Tabs tb = make_tabs();
    ScopedTimer tm(KID_SDF_FWD_X3, src.M, st);
    const dim3 grid(L.Mp / 128), block(XR_THREADS);
    const u32x4* pk = reinterpret_cast<const u32x4*>(packed_r);
    float* xc = ws + L.off[WS_XC]; float* sdf = ws + L.off[WS_SDF]; float* feat = ws + L.off[WS_FEAT]; float* gc = ws + L.off[WS_GC];
    float* go = ws + L.off[WS_GO]; float* sact = ws + L.off[WS_S_ACT];
#define ES_LAUNCH_SDF_X3R(D, Cc) hipLaunchKernelGGL((k_sdf_fwd_x3r<D, Cc>), grid, block, XS_LDS_BYTES, st, src, tb, pk, weff, xc, sdf, feat, gc, go, sact, L.Mp)
    if (deform) { if (color) ES_LAUNCH_SDF_X3R(true, true); else ES_LAUNCH_SDF_X3R(true, false); }
    else { if (color) ES_LAUNCH_SDF_X3R(false, true); else ES_LAUNCH_SDF_X3R(false, false); }
#undef ES_LAUNCH_SDF_X3R
    return hip_last("sdf_fwd_x3r");
}

// colour of the points [0, Mcp); save (training): whole 128-point blocks, + WS_C_IN / WS_C_H / WS_C_MASK
int color_fwd_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, bool deform, int Mcp, bool save, hipStream_t st) {
    if (int e = infer_attrs()) return e;
    if (Mcp <= 0) return ST_OK;
    const Tabs tb = make_tabs();
    ScopedTimer tm(KID_COLOR_FWD_X3, Mcp, st);
    const dim3 grid((Mcp + 127) / 128), block(XR_THREADS);
    const u32x4* pk = reinterpret_cast<const u32x4*>(packed_r);
    float* cin = ws + L.off[WS_C_IN]; float* ch = ws + L.off[WS_C_H]; u32x4* mk = reinterpret_cast<u32x4*>(ws + L.off[WS_C_MASK]);
#define ES_LAUNCH_COLOR_X3R(D, S) hipLaunchKernelGGL((k_color_fwd_x3r<D, S>), grid, block, XC_LDS_BYTES, st, src, tb, pk, weff, ws + L.off[WS_XC], \
        ws + L.off[WS_V], ws + L.off[WS_GC], ws + L.off[WS_FEAT], ws + L.off[WS_RGB], cin, ch, mk, Mcp, L.Mp)
    if (deform) { if (save) ES_LAUNCH_COLOR_X3R(true, true); else ES_LAUNCH_COLOR_X3R(true, false); }
    else { if (save) ES_LAUNCH_COLOR_X3R(false, true); else ES_LAUNCH_COLOR_X3R(false, false); }
#undef ES_LAUNCH_COLOR_X3R
    return hip_last("color_fwd_x3r");
}

}  // namespace es
