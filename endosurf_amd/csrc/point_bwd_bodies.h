// The tile bodies of the fp32 backward chain kernels (point_bwd.hip describes them): colour reverse sweep, SDF tangent + reverse sweep,
// deformation tangent sweep, deformation reverse sweep.  A header for the same reason as point_fwd_bodies.h.
#pragma once
#include <type_traits>

#include "chain_common.h"
#include "encode.h"
#include "tabs.h"
#include "workspace.h"

namespace es {

#ifdef ES_PROFILE_BWD        // dev builds only: cycle stamps of block 0 / thread 0 inside sdf_bwd_tile (tools/dev/bwd_profile.py)
extern __device__ long long b_prof[256];
#define B_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) b_prof[i] = __builtin_readcyclecounter(); } while (0)
#else
#define B_STAMP(i) do {} while (0)
#endif

struct BwdArgs {
    PointSrc src;
    Tabs tb;
    const float4* packed;
    const float* weff;
    float* ws;
    WsLayout L;
    int flags;
    int M_color;       // points [0, M_color) go through the colour network (multiple of 64 unless == M)
    const float* d_sdf;   // [M]
    const float* d_go;    // [M][3]
    const float* d_rgb;   // [M][3] (colour only)
};
__device__ __forceinline__ float* wsb(const BwdArgs& a, int buf) { return a.ws + a.L.off[buf]; }

// d enc/dx contraction: sum_k adj[k] * d enc_k / d x_j for a 3-D encoding with L frequencies starting at row kbase
template <int L>
__device__ __forceinline__ float enc3_adjoint(const float* At, int kbase, int j, int row, float x) {
    float g = At[swz(kbase + j, row)];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const float f = (float)(1 << i);
        float s, co;
        sincosf(x * f, &s, &co);
        g += f * (At[swz(kbase + enc_index(3, i, 0, j), row)] * co - At[swz(kbase + enc_index(3, i, 1, j), row)] * s);
    }
    return g;
}

// same contraction with the adjoint row in global memory (adj[idx], idx relative to the encoding's first element)
template <int L>
__device__ __forceinline__ float enc3_adjoint_g(const float* __restrict__ adj, int j, float x) {
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

// -------------------------------------------------------------------------------------------------------------
// lean carve (activation tile + 3.75 KB): the adjoint of the input's small part accumulates in HBM (WS_C_SBAR), not in LDS
constexpr int CBWD_LDS_BYTES = (MAIN_FLOATS + 960) * 4;   // 69 376 B
__device__ __forceinline__ void color_bwd_tile(const BwdArgs& a, const int tile) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    float* scr = lds + MAIN_FLOATS;
    float* y8 = scr;           // [3][64]
    float* px = scr + 192;     // [3][64] x_c
    float* pd = scr + 384;     // [3][64] d_c
    float* tx = scr + 576;     // [3][64]
    float* td = scr + 768;     // [3][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = tile * TM;
    const size_t grow0 = (size_t)row0;
    const bool deform = a.flags & PF_DEFORM;
    const size_t Mp = (size_t)a.L.Mp;
    float dray[3] = {0.f, 0.f, 1.f};

    if (tid < 64) {
        const size_t gp = grow0 + tid;
        const bool valid = row0 + tid < a.M_color;
        float x[3], t;
        load_point(a.src, row0 + tid, x, t, dray);
        const float* rgb = wsb(a, WS_RGB) + gp * 3;
        float* Y8 = wsb(a, WS_C_Y8) + gp * 4;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float c = rgb[i];
            const float y = valid ? a.d_rgb[3 * (size_t)(row0 + tid) + i] * c * (1.f - c) : 0.f;   // sigmoid'
            y8[i * 64 + tid] = y; Y8[i] = y;
        }
        Y8[3] = 0.f;
        const float* xc = wsb(a, WS_XC) + gp * 3;
        px[tid] = xc[0]; px[64 + tid] = xc[1]; px[128 + tid] = xc[2];
        float v0 = dray[0], v1 = dray[1], v2 = dray[2];
        if (deform) {
            const float* v = wsb(a, WS_V) + gp * 3;
            v0 = v[0]; v1 = v[1]; v2 = v[2];
        }
        const float inv = 1.f / (sqrtf(v0 * v0 + v1 * v1 + v2 * v2) + 1e-10f);
        pd[tid] = v0 * inv; pd[64 + tid] = v1 * inv; pd[128 + tid] = v2 * inv;
    }
    __syncthreads();
    const unsigned long long* CM = reinterpret_cast<const unsigned long long*>(wsb(a, WS_C_MASK));   // this thread's own words
    const size_t nt64 = Mp / 64;
    float* CY = wsb(a, WS_C_Y);
    {   // ybar_7 = relu'(y_7) * (U8^T ybar_8)
        const float* U8 = a.weff + a.tb.woff[NET_C * LAYERS + 8];
        const unsigned long long bits = CM[((size_t)7 * nt64 + tile) * 256 + tid];
        for_quads_noacc<2, 2>(0, 2 * wave, lane, [&](int row, int col) {
            const float u0 = U8[col], u1 = U8[256 + col], u2 = U8[512 + col];
            const int qi = ((row >> 5) * 2 + ((col >> 5) & 1)) * 4 + ((row & 31) >> 3);
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float hb = y8[row + i] * u0 + y8[64 + row + i] * u1 + y8[128 + row + i] * u2;
                v[i] = ((bits >> (4 * qi + i)) & 1ull) ? hb : 0.f;
            }
            lds_store_quad(mainT, col, row, v);
            g_store_quad(CY + (size_t)7 * Mp * 256, grow0, 256, row, col, v);
        });
    }
    __syncthreads();
    const QuadOff<2> qo = quad_offsets<2>(0, 2 * wave, lane);
    auto epi = [&](f32x16(&acc)[2][2], int l, unsigned long long bits) {   // acc = hbar_l ; ybar_{l-1} = relu'(.) * hbar_l
        float* Yl = CY + (size_t)(l - 1) * Mp * 256;
        const unsigned blo = (unsigned)bits, bhi = (unsigned)(bits >> 32);      // quads 0..7 | 8..15
        for_quads_off(acc, qo, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int off, int ni) {
            const int qi = (((row >> 5) * 2 + ni) << 2) + ((row >> 3) & 3);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = keep_if_bit(v[i], qi < 8 ? blo : bhi, 4 * (qi & 7) + i);
            lds_store_quad_at(mainT, off, v);
            g_store_quad(Yl, grow0, 256, row, col, v);
        });
    };
    float* FB = wsb(a, WS_FEATBAR);
    float* SB = wsb(a, WS_C_SBAR);
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
        const unsigned long long bits = CM[((size_t)(l - 1) * nt64 + tile) * 256 + tid];      // in flight during the GEMMs
        if (l == 4) {   // skip layer: adjoint also flows to the network input [small(93) | feat(256)]
            {
                f32x16 accF[2][2];
                acc_zero(accF);
                gemm_seg<32, 2, 2>(accF, mainT, a.packed + a.tb.segoff[CR4F], 0, 2 * wave, lane);
                for_quads(accF, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4]) { g_store_quad(FB, grow0, 256, row, col, v); });
            }
            {
                f32x16 accS[2][1];
                acc_zero(accS);
                gemm_seg<32, 2, 1>(accS, mainT, a.packed + a.tb.segoff[CR4S], 0, wave, lane);
                for_quads(accS, 0, wave, lane, [&](int row, int col, float(&v)[4]) { if (col < 96) g_store_quad(SB, grow0, 128, row, col, v); });
            }
        }
        f32x16 acc[2][2];
        acc_zero(acc);
        const int seg = l < 4 ? CR1 + (l - 1) : (l == 4 ? (int)CR4H : CR5 + (l - 5));
        gemm_seg<32, 2, 2>(acc, mainT, a.packed + a.tb.segoff[seg], 0, 2 * wave, lane);
        __syncthreads();
        epi(acc, l, bits);
        __syncthreads();
    }
    {   // layer 0: adjoint of the network input
        f32x16 accF[2][2];
        acc_zero(accF);
        gemm_seg<32, 2, 2>(accF, mainT, a.packed + a.tb.segoff[CR0F], 0, 2 * wave, lane);
        for_quads(accF, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4]) {
            float p[4];
            g_load_quad(FB, grow0, 256, row, col, p);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += p[i];
            g_store_quad(FB, grow0, 256, row, col, v);
        });
        f32x16 accS[2][1];
        acc_zero(accS);
        gemm_seg<32, 2, 1>(accS, mainT, a.packed + a.tb.segoff[CR0S], 0, wave, lane);
        for_quads(accS, 0, wave, lane, [&](int row, int col, float(&v)[4]) {
            if (col < 96) {
                float p[4];
                g_load_quad(SB, grow0, 128, row, col, p);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += p[i];
                g_store_quad(SB, grow0, 128, row, col, v);
            }
        });
    }
    __syncthreads();
    if (tid < 192) {
        const int j = tid >> 6, row = tid & 63;
        const float* sbr = SB + (grow0 + row) * 128;
        tx[j * 64 + row] = enc3_adjoint_g<10>(sbr, j, px[j * 64 + row]);
        td[j * 64 + row] = enc3_adjoint_g<4>(sbr + 66, j, pd[j * 64 + row]);
    }
    __syncthreads();
    if (tid < 64) {
        const size_t gp = grow0 + tid;
        float* xb = wsb(a, WS_XCBAR_C) + gp * 3;
        float* gb = wsb(a, WS_GCBAR_C) + gp * 3;
        float* Vb = wsb(a, WS_VBAR_C) + gp * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) { xb[j] = tx[j * 64 + tid]; gb[j] = SB[gp * 128 + 63 + j]; }
        {   // d_c = v/(|v| + eps), v = J d (the view direction itself without a deformation network)  ->  vbar: seeds the J d row of the
            // deformation backward; public as ES_WS_VBAR (the adjoint of the view direction, up to J^T: EndoSurfNet.forward's input gradient)
            const float* vv = wsb(a, WS_V) + gp * 3;
            const float v[3] = {deform ? vv[0] : dray[0], deform ? vv[1] : dray[1], deform ? vv[2] : dray[2]};
            const float db[3] = {td[tid], td[64 + tid], td[128 + tid]};
            const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const float den = n + 1e-10f;
            const float dot = v[0] * db[0] + v[1] * db[1] + v[2] * db[2];
            const float k2 = n > 0.f ? dot / (n * den * den) : 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) Vb[i] = db[i] / den - v[i] * k2;
        }
    }
}

// -------------------------------------------------------------------------------------------------------------
// lean carve: activation tile | 40-row auxiliary tile | 640 floats of per-row data  => two workgroups per CU
constexpr int SBWD_LDS_BYTES = (MAIN_FLOATS + 40 * TM + 640) * 4;   // 78 336 B
__device__ __forceinline__ void sdf_bwd_tile(const BwdArgs& a, const int tile) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    float* aux = lds + MAIN_FLOATS;
    float* scr = aux + 40 * TM;
    float* px = scr;           // [3][64] x_c
    float* gb = scr + 192;     // [3][64] gbar_c
    float* sb = scr + 384;     // [64] sdfbar
    float* tx = scr + 448;     // [3][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = tile * TM;
    const size_t grow0 = (size_t)row0;
    B_STAMP(200);
    const bool deform = a.flags & PF_DEFORM, color = (a.flags & PF_COLOR) && row0 < a.M_color;
    const size_t Mp = (size_t)a.L.Mp;

    if (tid < 64) {
        const size_t gp = grow0 + tid;
        const bool valid = row0 + tid < a.src.M;
        float go[3] = {0.f, 0.f, 0.f};
        if (valid) { go[0] = a.d_go[3 * (size_t)(row0 + tid)]; go[1] = a.d_go[3 * (size_t)(row0 + tid) + 1]; go[2] = a.d_go[3 * (size_t)(row0 + tid) + 2]; }
        sb[tid] = valid ? a.d_sdf[row0 + tid] : 0.f;
        const float* xc = wsb(a, WS_XC) + gp * 3;
        px[tid] = xc[0]; px[64 + tid] = xc[1]; px[128 + tid] = xc[2];
        // gbar_c = J gbar_o (+ the colour network's): J gbar_o comes from the deformation tangent sweep (deform_tan_tile)
        const float* ju = deform ? wsb(a, WS_JU) + gp * 3 : go;
#pragma unroll
        for (int i = 0; i < 3; ++i) gb[i * 64 + tid] = ju[i] + (color ? wsb(a, WS_GCBAR_C)[gp * 3 + i] : 0.f);
    }
    __syncthreads();
    {   // tau_0 = (d enc6 / d x_c) gbar_c
        const int row = tid & 63, part = tid >> 6;
        for (int item = part; item < 18; item += 4) {
            const int c = item % 3, i = item / 3;
            const float f = (float)(1 << i);
            float s, co;
            sincosf(px[c * 64 + row] * f, &s, &co);
            const float g = gb[c * 64 + row];
            aux[swz(enc_index(3, i, 0, c), row)] = f * co * g;
            aux[swz(enc_index(3, i, 1, c), row)] = -f * s * g;
        }
        if (part == 3) {
#pragma unroll
            for (int c = 0; c < 3; ++c) aux[swz(c, row)] = gb[c * 64 + row];
            aux[swz(39, row)] = 0.f;
        }
    }
    __syncthreads();
    {
        float* T0 = wsb(a, WS_S_TAU0);
        const int r = tid >> 2, c4 = tid & 3;
        for (int k = c4; k < 40; k += 4) T0[(grow0 + r) * 64 + k] = aux[swz(k, r)];
    }
    const float* SACT = wsb(a, WS_S_ACT);
    const float* RHO = wsb(a, WS_S_RHO);
    float* TAU = wsb(a, WS_S_TAU);
    float* ZB = wsb(a, WS_S_ZB);
    // ---- (i) forward tangent sweep ----
    // (epilogue operands are loaded in the epilogue: the co-resident workgroup's MFMAs cover the HBM latency)
    auto epi_t = [&](f32x16(&acc)[2][2], int l) {   // acc = pi_l
        auto half = [&](auto NI) {
            float S[8][4], Rr[8][4];
            prefetch_half_f<decltype(NI)::value>(S, SACT + (size_t)l * Mp * 256, grow0, 2 * wave, lane);
            prefetch_half_f<decltype(NI)::value>(Rr, RHO + (size_t)l * Mp * 256, grow0, 2 * wave, lane);
            for_quads_half<decltype(NI)::value>(acc, 2 * wave, lane, [&](int row, int col, float(&v)[4], int b8) {
                float z2[4], dphi[4];
                softplus100_grad_from_s4(S[b8], dphi);
#pragma unroll
                for (int h = 0; h < 2; ++h) {      // packed pairs
                    const f32x2p d = {dphi[2 * h], dphi[2 * h + 1]}, r = {Rr[b8][2 * h], Rr[b8][2 * h + 1]}, t = {v[2 * h], v[2 * h + 1]};
                    const f32x2p zz = 100.f * (1.f - d) * r * t;            // softplus'' / softplus' = 100 (1 - softplus')
                    const f32x2p tt = d * t;                                // tau_{l+1}
                    z2[2 * h] = zz[0]; z2[2 * h + 1] = zz[1]; v[2 * h] = tt[0]; v[2 * h + 1] = tt[1];
                }
                lds_store_quad(mainT, col, row, v);
                g_store_quad_f(TAU + (size_t)l * Mp * 256, grow0, row, col, v);
                g_store_quad_f(ZB + (size_t)l * Mp * 256, grow0, row, col, z2);
            });
        };
        half(std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 1>{});
    };
    {
        f32x16 acc[2][2];
        acc_zero(acc);
        gemm_seg<5, 2, 2>(acc, aux, a.packed + a.tb.segoff[SF0], 0, 2 * wave, lane);
        epi_t(acc, 0);
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        f32x16 acc[2][2];
        acc_zero(acc);
        const int seg = l <= 4 ? SF0 + l : SF0 + l + 1;
        B_STAMP(4 * l);
        gemm_seg<32, 2, 2>(acc, mainT, a.packed + a.tb.segoff[seg], 0, 2 * wave, lane);
        if (l == 4) gemm_seg<5, 2, 2>(acc, aux, a.packed + a.tb.segoff[SF4A], 0, 2 * wave, lane);
        B_STAMP(4 * l + 1);
        __syncthreads();
        B_STAMP(4 * l + 2);
        epi_t(acc, l);
        B_STAMP(4 * l + 3);
        __syncthreads();
    }
    // ---- (ii) reverse sweep of the value pass, seeded with zbar_8 = [sdfbar | featbar] ----
    auto epi_b = [&](f32x16(&acc)[2][2], int l) {   // acc = sbar_l; zbar_{l-1} = phi'(z_{l-1}) sbar_l + second-order term
        auto half = [&](auto NI) {
            float S[8][4], Z2[8][4];
            prefetch_half_f<decltype(NI)::value>(S, SACT + (size_t)(l - 1) * Mp * 256, grow0, 2 * wave, lane);
            prefetch_half_f<decltype(NI)::value>(Z2, ZB + (size_t)(l - 1) * Mp * 256, grow0, 2 * wave, lane);
            for_quads_half<decltype(NI)::value>(acc, 2 * wave, lane, [&](int row, int col, float(&v)[4], int b8) {
                float dphi[4];
                softplus100_grad_from_s4(S[b8], dphi);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x2p d = {dphi[2 * h], dphi[2 * h + 1]}, t = {v[2 * h], v[2 * h + 1]}, z = {Z2[b8][2 * h], Z2[b8][2 * h + 1]};
                    const f32x2p o = d * t + z;
                    v[2 * h] = o[0]; v[2 * h + 1] = o[1];
                }
                lds_store_quad(mainT, col, row, v);
                g_store_quad_f(ZB + (size_t)(l - 1) * Mp * 256, grow0, row, col, v);
            });
        };
        half(std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 1>{});
    };
    {
        f32x16 acc[2][2];
        acc_zero(acc);
        if (color) {
            load_tile_256(mainT, wsb(a, WS_FEATBAR), grow0, 256, tid);
            __syncthreads();
            gemm_seg<32, 2, 2>(acc, mainT, a.packed + a.tb.segoff[SR8F], 0, 2 * wave, lane);
            __syncthreads();
        }
        const float* w8 = a.weff + a.tb.woff[NET_S * LAYERS + 8];
        for_quads(acc, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4]) {
            const float w = w8[col];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += sb[row + i] * w;
            float s[4], z2[4];
            g_load_quad_f(SACT + (size_t)7 * Mp * 256, grow0, row, col, s);
            g_load_quad_f(ZB + (size_t)7 * Mp * 256, grow0, row, col, z2);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = softplus100_grad_from_s(s[i]) * v[i] + z2[i];
            lds_store_quad(mainT, col, row, v);
            g_store_quad_f(ZB + (size_t)7 * Mp * 256, grow0, row, col, v);
        });
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
        f32x16 acc[2][2];
        acc_zero(acc);
        const int seg = l <= 4 ? SR0 + l : SR0 + l + 1;
        B_STAMP(100 + 4 * l);
        gemm_seg<32, 2, 2>(acc, mainT, a.packed + a.tb.segoff[seg], 0, 2 * wave, lane);
        f32x16 accA[1][1];
        if (l == 4) {
            acc_zero(accA);
            gemm_seg<32, 1, 1>(accA, mainT, a.packed + a.tb.segoff[SR4A], wave >> 1, wave & 1, lane);
        }
        B_STAMP(100 + 4 * l + 1);
        __syncthreads();
        B_STAMP(100 + 4 * l + 2);
        epi_b(acc, l);
        if (l == 4)
            for_quads(accA, wave >> 1, wave & 1, lane, [&](int row, int col, float(&v)[4]) { if (col < 40) lds_store_quad(aux, col, row, v); });
        B_STAMP(100 + 4 * l + 3);
        __syncthreads();
    }
    {
        f32x16 accA[1][1];
        acc_zero(accA);
        gemm_seg<32, 1, 1>(accA, mainT, a.packed + a.tb.segoff[SR0], wave >> 1, wave & 1, lane);
        for_quads(accA, wave >> 1, wave & 1, lane, [&](int row, int col, float(&v)[4]) { if (col < 40) lds_add_quad(aux, col, row, v); });
    }
    __syncthreads();
    if (tid < 192) {
        const int j = tid >> 6, row = tid & 63;
        const float x = px[j * 64 + row];
        float g = enc3_adjoint<6>(aux, 0, j, row, x);
        // second-order encoding term: sum_k adj_eps[k] * d2 enc_k / dx_j^2 * gbar_c[j]
        const float* AE = wsb(a, WS_S_ADJEPS) + (grow0 + row) * 64;
        float h = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float f = (float)(1 << i);
            float s, co;
            sincosf(x * f, &s, &co);
            h -= f * f * (AE[enc_index(3, i, 0, j)] * s + AE[enc_index(3, i, 1, j)] * co);
        }
        tx[j * 64 + row] = g + h * gb[j * 64 + row];
    }
    __syncthreads();
    if (tid < 64) {
        const size_t gp = grow0 + tid;
        float* xb = wsb(a, WS_XCBAR) + gp * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) xb[j] = tx[j * 64 + tid] + (color ? wsb(a, WS_XCBAR_C)[gp * 3 + j] : 0.f);
    }
    B_STAMP(201);
}

// -------------------------------------------------------------------------------------------------------------
// Deformation network, forward tangent sweep along gbar_o (the adjoint of g_o = J^T g_c with respect to g_c is J gbar_o):
//   tau_0 = E(x) gbar_o,  tau_{l+1} = M_l (W_l tau_l),  J gbar_o = gbar_o + W_8 tau_8.
// Tile = 64 points, one row per point.  tau_0..tau_8 are kept: (tau_l, r_l) with the r_l of the VJP sweep is the weight
// gradient of the g_o path.  Runs before the SDF backward, which consumes J gbar_o.
__device__ __forceinline__ void deform_tan_tile(const BwdArgs& a, const int tile) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    float* aux = lds + MAIN_FLOATS;          // 56 rows: tau_0 (52 valid)
    float* scr = aux + AUX56_FLOATS;
    float* ub = scr;                         // [3][64] gbar_o
    float* px = mainT;                       // [3][64] x, only until the first epilogue overwrites the tile
    float* red = aux;                        // [4][3][64]: tau_0 is dead after layer 3's epilogue
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = tile * TM;
    const size_t grow0 = (size_t)row0;
    const size_t Mp = (size_t)a.L.Mp;
    const unsigned* MK = reinterpret_cast<const unsigned*>(wsb(a, WS_D_MASK));
    const size_t nt32 = Mp / 32;
    const int hi = lane >> 5;
    float* T = wsb(a, WS_D_T);

    if (tid < 64) {
        const bool valid = row0 + tid < a.src.M;
        float x[3], t, d[3];
        load_point(a.src, row0 + tid, x, t, d);
        px[tid] = x[0]; px[64 + tid] = x[1]; px[128 + tid] = x[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) ub[i * 64 + tid] = valid ? a.d_go[3 * (size_t)(row0 + tid) + i] : 0.f;
    }
    zero_rows(aux, 0, 56, tid);
    __syncthreads();
    {
        const int row = tid & 63, part = tid >> 6;
        for (int item = part; item < 18; item += 4) {
            const int c = item % 3, i = item / 3;
            const float f = (float)(1 << i);
            float s, co;
            sincosf(px[c * 64 + row] * f, &s, &co);
            const float g = ub[c * 64 + row];
            aux[swz(enc_index(3, i, 0, c), row)] = f * co * g;
            aux[swz(enc_index(3, i, 1, c), row)] = -f * s * g;
        }
        if (part == 3) {
#pragma unroll
            for (int c = 0; c < 3; ++c) aux[swz(c, row)] = ub[c * 64 + row];
        }
    }
    __syncthreads();
    {
        float* T0 = wsb(a, WS_D_T0);
        const int r = tid >> 2, c4 = tid & 3;
        for (int k = c4; k < 56; k += 4) T0[(grow0 + r) * 64 + k] = aux[swz(k, r)];
    }
    // (the skip layer's per-lane test is a compile-time instantiation behind a wave-uniform branch: see deform_fwd_tile)
    const QuadOff<2> qo = quad_offsets<2>(0, 2 * wave, lane);
    auto epi_impl = [&](f32x16(&acc)[2][2], int l, const MaskWords& mk, auto SKIP) {
        float* Tl = T + (size_t)l * Mp * 256;
        for_quads_off(acc, qo, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int off, int ni) {
            const int qi = (((row >> 5) * 2 + ni) << 2) + ((row >> 3) & 3);
            if (decltype(SKIP)::value && col >= 204) {
                lds_load_quad(aux, col - 204, row, v);          // IDR skip: next input = [h(204) | enc(52)]
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = mask_keep(v[i], mk, qi, i, hi);
            }
            lds_store_quad_at(mainT, off, v);
            g_store_quad(Tl, grow0, 256, row, col, v);
        });
    };
    auto epi = [&](f32x16(&acc)[2][2], int l, const MaskWords& mk) {
        if (l == 3 && wave == 3) epi_impl(acc, l, mk, std::true_type{});
        else epi_impl(acc, l, mk, std::false_type{});
    };
    {
        const MaskWords mk = load_mask_words(MK, tile, wave, lane);
        f32x16 acc[2][2];
        acc_zero(acc);
        gemm_seg<7, 2, 2>(acc, aux, a.packed + a.tb.segoff[DF0], 0, 2 * wave, lane);
        epi(acc, 0, mk);
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        const MaskWords mk = load_mask_words(MK + (size_t)l * nt32 * 256, tile, wave, lane);          // in flight during the GEMM
        f32x16 acc[2][2];
        acc_zero(acc);
        gemm_seg<32, 2, 2>(acc, mainT, a.packed + a.tb.segoff[DF0 + l], 0, 2 * wave, lane);
        __syncthreads();
        epi(acc, l, mk);
        __syncthreads();
    }
    smalln_partial<3>(mainT, a.weff + a.tb.woff[NET_D * LAYERS + 8], 256, red, tid);
    __syncthreads();
    if (tid < 192) {
        const int i = tid >> 6, row = tid & 63;
        wsb(a, WS_JU)[(grow0 + row) * 3 + i] = smalln_reduce<3>(red, i, row) + ub[i * 64 + row];
    }
}

// -------------------------------------------------------------------------------------------------------------
// Deformation network, reverse sweep of the value row (seed xbar_c) and of the J d row (seed vbar from the colour network).
// Tile = 32 points = 64 rows (row 2p = value, 2p + 1 = tangent).  LDS: activation tile + 768 B => two workgroups per CU.
// HALF: 16 points = 32 rows per workgroup (the stand-alone launch of a training batch's tail runs at one tile's latency: see deform_fwd_tile).
constexpr int DBWD_LDS_BYTES = (MAIN_FLOATS + 192) * 4;
template <bool HALF = false>
__device__ __forceinline__ void deform_bwd_tile(const BwdArgs& a, const int tile) {
    constexpr int PTS = HALF ? 16 : 32, RTC = HALF ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    float* a8 = lds + MAIN_FLOATS;   // [3][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pt0 = tile * PTS;
    const size_t grow0 = (size_t)pt0 * 2;
    const size_t rows2 = (size_t)a.L.Mp * 2;
    const bool color = a.flags & PF_COLOR;
    // this thread's mask word of layer l (its own word of the forward tile; a half tile owns 16 bits of the 32-point tile's word)
    auto mask_word = [&](const unsigned* MK, size_t nt32, int l) -> unsigned {
        if constexpr (HALF) return MK[((size_t)l * nt32 + (tile >> 1)) * 256 + tid] >> (16 * (tile & 1));
        else return MK[((size_t)l * nt32 + tile) * 256 + tid];
    };

    if (tid < 2 * PTS) {
        const int p = tid >> 1, c = tid & 1;
        const size_t gp = (size_t)(pt0 + p);
        const bool has_v = color && pt0 + p < a.M_color;
        float* A8 = wsb(a, WS_D_A8) + (grow0 + tid) * 4;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float v = c == 0 ? wsb(a, WS_XCBAR)[gp * 3 + i] : (has_v ? wsb(a, WS_VBAR_C)[gp * 3 + i] : 0.f);
            a8[i * 64 + tid] = v; A8[i] = v;
        }
        A8[3] = 0.f;
    }
    __syncthreads();
    const unsigned* MK = reinterpret_cast<const unsigned*>(wsb(a, WS_D_MASK));
    const size_t nt32 = (size_t)a.L.Mp / 32;
    float* DA = wsb(a, WS_D_A);
    {   // abar_7 = mask_7 * (W8^T abar_8)
        const float* W8 = a.weff + a.tb.woff[NET_D * LAYERS + 8];
        const unsigned bits = mask_word(MK, nt32, 7);
        for_quads_noacc<RTC, 2>(0, 2 * wave, lane, [&](int row, int col) {
            const float w0 = W8[col], w1 = W8[256 + col], w2 = W8[512 + col];
            const int qi = ((row >> 5) * 2 + ((col >> 5) & 1)) * 4 + ((row & 31) >> 3);
            const bool m0 = (bits >> (2 * qi)) & 1u, m2 = (bits >> (2 * qi + 1)) & 1u;   // value rows
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (i < 2 ? m0 : m2) ? a8[row + i] * w0 + a8[64 + row + i] * w1 + a8[128 + row + i] * w2 : 0.f;
            lds_store_quad(mainT, col, row, v);
            g_store_quad(DA + (size_t)7 * rows2 * 256, grow0, 256, row, col, v);
        });
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
        const unsigned bits = mask_word(MK, nt32, l - 1);        // in flight during the GEMM
        f32x16 acc[RTC][2];
        acc_zero(acc);
        if (l == 3) gemm_seg<26, RTC, 2>(acc, mainT, a.packed + a.tb.segoff[DR3], 0, 2 * wave, lane);
        else gemm_seg<32, RTC, 2>(acc, mainT, a.packed + a.tb.segoff[DR0 + l], 0, 2 * wave, lane);
        __syncthreads();
        float* Al = DA + (size_t)(l - 1) * rows2 * 256;
        for_quads_qi(acc, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int qi) {
            // layer 3 has 204 outputs: the skip's encoding part (cols >= 204) carries no parameter gradient
            const bool dead = l == 4 && col >= 204;
            const bool m0 = !dead && ((bits >> (2 * qi)) & 1u), m2 = !dead && ((bits >> (2 * qi + 1)) & 1u);
            v[0] = m0 ? v[0] : 0.f; v[1] = m0 ? v[1] : 0.f; v[2] = m2 ? v[2] : 0.f; v[3] = m2 ? v[3] : 0.f;
            lds_store_quad(mainT, col, row, v);
            g_store_quad(Al, grow0, 256, row, col, v);
        });
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------------------------
}  // namespace es
