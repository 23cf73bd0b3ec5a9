// The tile bodies of the fp32 forward chain kernels (point_fwd.hip describes them): deformation value + tangent, deformation VJP, SDF value
// + reverse sweep, colour.  A header so that the split-precision family's launches can host the fp32 bodies of a batch's colour-less
// tail (infer_x3r.hip / train_x3r.hip: two-segment launches across the families).
#pragma once
#include <type_traits>
#include "chain_common.h"
#include "encode.h"
#include "tabs.h"
#include "workspace.h"

namespace es {

struct FwdArgs {
    PointSrc src;
    Tabs tb;
    const float4* packed;
    const float* weff;
    float* ws;
    WsLayout L;
    int flags;
    int M_color;       // points [0, M_color) go through the colour network (multiple of 64 unless == M)
};

__device__ __forceinline__ float* wsb(const FwdArgs& a, int buf) { return a.ws + a.L.off[buf]; }

// -------------------------------------------------------------------------------------------------------------
// deformation network, value + forward-mode tangent along the ray direction d:  x_c = x + MLP(x, t) and v = J d.
// Tile = 32 points = 64 rows (row 2p = value, row 2p + 1 = tangent).  The layer outputs u_1..u_8 are always streamed out:
// their value rows are the ReLU masks of the VJP sweep below (inference writes only those rows).
// HALF: 16 points = 32 rows per workgroup (row tile 0 of the same LDS layout; the GEMMs issue half the MFMAs) for the stand-alone launch of a
// training batch's colour-less tail: 96 tiles run at ONE tile's latency whatever their number, so half the height is ~half the time
// (0.16 -> 0.09 ms).  A half tile writes its 16 mask bits per thread into its half of the 32-point tile's word.
template <bool HALF = false>
__device__ __forceinline__ void deform_fwd_tile(const FwdArgs& a, const int tile) {
    constexpr int PTS = HALF ? 16 : 32, RTC = HALF ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    float* aux = lds + MAIN_FLOATS;
    float* scr = aux + AUX56_FLOATS;
    float* px = scr;         // [3][32]
    float* pd = scr + 96;    // [3][32] ray direction
    float* pt = scr + 192;   // [32]
    float* red = aux;        // [4][3][64]: the encoding rows are dead after layer 3's epilogue
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pt0 = tile * PTS;
    const size_t grow0 = (size_t)pt0 * 2;
    const bool save = a.flags & PF_SAVE;
    const size_t rows2 = (size_t)a.L.Mp * 2;

    if (tid < PTS) {
        float x[3], t, d[3];
        load_point(a.src, pt0 + tid, x, t, d);
        px[tid] = x[0]; px[32 + tid] = x[1]; px[64 + tid] = x[2]; pt[tid] = t;
        pd[tid] = d[0]; pd[32 + tid] = d[1]; pd[64 + tid] = d[2];
    }
    zero_rows(aux, 0, 56, tid);
    __syncthreads();
    {   // encoding rows: value row 2p, tangent row 2p+1 = (d enc / d x) d; the time part has no tangent
        const int p = tid & 31;
        for (int item = tid >> 5; item < 25 && p < PTS; item += 8) {
            if (item < 18) {
                const int c = item % 3, i = item / 3;
                const float f = (float)(1 << i);
                float s, co;
                sincosf(px[c * 32 + p] * f, &s, &co);
                const float dc = pd[c * 32 + p];
                aux[swz(enc_index(3, i, 0, c), 2 * p)] = s;
                aux[swz(enc_index(3, i, 1, c), 2 * p)] = co;
                aux[swz(enc_index(3, i, 0, c), 2 * p + 1)] = f * co * dc;
                aux[swz(enc_index(3, i, 1, c), 2 * p + 1)] = -f * s * dc;
            } else if (item < 24) {
                const int i = item - 18;
                float s, co;
                sincosf(pt[p] * (float)(1 << i), &s, &co);
                aux[swz(39 + enc_index(1, i, 0, 0), 2 * p)] = s;
                aux[swz(39 + enc_index(1, i, 1, 0), 2 * p)] = co;
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) { aux[swz(c, 2 * p)] = px[c * 32 + p]; aux[swz(c, 2 * p + 1)] = pd[c * 32 + p]; }
                aux[swz(39, 2 * p)] = pt[p];
            }
        }
    }
    __syncthreads();
    if (save) {   // u_0 rows for the weight-gradient GEMM: [2Mp][64], 56 columns written
        float* U0 = wsb(a, WS_D_U0);
        const int r = tid >> 2, c4 = tid & 3;
        if (r < 2 * PTS)
            for (int k = c4; k < 56; k += 4) U0[(grow0 + r) * 64 + k] = aux[swz(k, r)];
    }

    float* U = wsb(a, WS_D_U);
    unsigned* MK = reinterpret_cast<unsigned*>(wsb(a, WS_D_MASK));
    const size_t nt32 = (size_t)a.L.Mp / 32;
    // Epilogue.  Every non-MFMA instruction of a wave adds to its MFMA time, so the regular layers carry no per-lane
    // branch: the skip layer's copy (l == 3, columns >= 204: wave 3 only) is its own instantiation behind a wave-uniform test -- inside
    // one lambda it cost every quad of every layer an exec-mask branch and the phi moves behind it (708 -> ~300 instructions per layer
    // and wave) --, the bias is requested a layer ahead (two registers) and ``save`` is uniform.
    const QuadOff<RTC> qo = quad_offsets<RTC>(0, 2 * wave, lane);
    auto bias2 = [&](float(&b)[2], int l) {
        const float* bias = a.weff + a.tb.boff[NET_D * LAYERS + l] + 64 * wave + (lane & 31);
        b[0] = bias[0]; b[1] = bias[32];
    };
    auto epi_impl = [&](f32x16(&acc)[RTC][2], int l, const float(&bc)[2], auto SKIP, auto SAVE) {
        float* Ul = U + (size_t)l * rows2 * 256;
        unsigned bits = 0;
        for_quads_off(acc, qo, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int off, int ni) {       // rows: value, tangent, value, tangent
            const int qi = (((row >> 5) * 2 + ni) << 2) + ((row >> 3) & 3);
            if (decltype(SKIP)::value && col >= 204) {
                lds_load_quad(aux, col - 204, row, v);          // IDR skip: next input = [h(204) | enc(52)]
            } else {
                const float a0 = v[0] + bc[ni], a2 = v[2] + bc[ni];
                const bool m0 = a0 > 0.f, m2 = a2 > 0.f;         // the ReLU mask of a value row gates its tangent
                v[0] = m0 ? a0 : 0.f; v[1] = m0 ? v[1] : 0.f; v[2] = m2 ? a2 : 0.f; v[3] = m2 ? v[3] : 0.f;
                bits |= (m0 ? 1u : 0u) << (2 * qi) | (m2 ? 1u : 0u) << (2 * qi + 1);
            }
            lds_store_quad_at(mainT, off, v);
            if (decltype(SAVE)::value) g_store_quad(Ul, grow0, 256, row, col, v);
        });
        // the masks of the VJP / tangent / reverse sweeps: one word per thread and 32-point tile (a half tile owns 16 bits of it)
        if constexpr (HALF) reinterpret_cast<unsigned short*>(MK)[(((size_t)l * nt32 + (tile >> 1)) * 256 + tid) * 2 + (tile & 1)] = (unsigned short)bits;
        else MK[((size_t)l * nt32 + tile) * 256 + tid] = bits;
    };
    auto epi = [&](f32x16(&acc)[RTC][2], int l, const float(&bc)[2], auto SKIP) {
        if (save) epi_impl(acc, l, bc, SKIP, std::true_type{});
        else epi_impl(acc, l, bc, SKIP, std::false_type{});
    };
    float bc[2], bn[2];
    bias2(bn, 0);
    {
        f32x16 acc[RTC][2];
        acc_zero(acc);
        bc[0] = bn[0]; bc[1] = bn[1];
        bias2(bn, 1);
        gemm_seg<7, RTC, 2>(acc, aux, a.packed + a.tb.segoff[DF0], 0, 2 * wave, lane);
        epi(acc, 0, bc, std::false_type{});
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        f32x16 acc[RTC][2];
        acc_zero(acc);
        bc[0] = bn[0]; bc[1] = bn[1];
        if (l < 7) bias2(bn, l + 1);
        gemm_seg<32, RTC, 2>(acc, mainT, a.packed + a.tb.segoff[DF0 + l], 0, 2 * wave, lane);
        __syncthreads();
        if (l == 3 && wave == 3) epi(acc, l, bc, std::true_type{});
        else epi(acc, l, bc, std::false_type{});
        __syncthreads();
    }
    smalln_partial<3>(mainT, a.weff + a.tb.woff[NET_D * LAYERS + 8], 256, red, tid);
    __syncthreads();
    if (tid < 192) {
        const int i = tid >> 6, row = tid & 63, p = row >> 1, c = row & 1;
        const float val = smalln_reduce<3>(red, i, row);
        const size_t gp = (size_t)(pt0 + p);
        if (p >= PTS) {
        } else if (c == 0) wsb(a, WS_XC)[gp * 3 + i] = px[i * 32 + p] + val + a.weff[a.tb.boff[NET_D * LAYERS + 8] + i];
        else wsb(a, WS_V)[gp * 3 + i] = val + pd[i * 32 + p];          // J d = d + (d Delta x / d x) d
    }
}

// -------------------------------------------------------------------------------------------------------------
// deformation network, reverse (VJP) sweep for the covector g_c:  g_o = J^T g_c = g_c + E(x)^T W_0^T M_0 W_1^T ... M_7 W_8^T g_c
// (get_sdf_grad_from_observed_space, endosurf.py:581-601, is exactly this product).  Tile = 64 points, one row per point;
// masks M_l from the value rows of u_{l+1}.  With PF_SAVE the adjoints r_0..r_7 are kept: paired with the tangent sweep of
// the backward pass they give this path's weight gradient.
// HALF: ``tile_in`` counts 32-row half tiles; the workgroup runs row tile rt = tile_in & 1 of the 64-row tile tile_in >> 1 (same LDS and
// workspace layout, half the MFMAs): a stand-alone piece of a colour-less tail is 16 - 32 workgroups at ONE tile's latency, so half the
// height is ~half the time.
template <bool HALF = false>
__device__ __forceinline__ void deform_vjp_tile(const FwdArgs& a, const int tile_in) {
    constexpr int RTC = HALF ? 1 : 2;
    const int tile = HALF ? (tile_in >> 1) : tile_in;
    const int RT0 = HALF ? (tile_in & 1) : 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    float* aux = lds + MAIN_FLOATS;          // adjoint of the 52 encoding inputs (skip part + layer 0), rows 52..55 zero
    float* scr = aux + AUX56_FLOATS;
    float* g8 = scr;                         // [3][64] g_c
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = tile * TM;
    const size_t grow0 = (size_t)row0;
    const bool save = a.flags & PF_SAVE;
    const size_t Mp = (size_t)a.L.Mp;
    const unsigned* MK = reinterpret_cast<const unsigned*>(wsb(a, WS_D_MASK));
    const size_t nt32 = Mp / 32;
    const int hi = lane >> 5;
    float* R = wsb(a, WS_D_R);

    const QuadOff<RTC> qo = quad_offsets<RTC>(RT0, 2 * wave, lane);
    if (tid < 64) {
        const float* gc = wsb(a, WS_GC) + (grow0 + tid) * 3;
        g8[tid] = gc[0]; g8[64 + tid] = gc[1]; g8[128 + tid] = gc[2];
    }
    zero_rows(aux, 0, 56, tid);
    __syncthreads();
    {   // r_7 = mask_7 * (W8^T g_c)
        const float* W8 = a.weff + a.tb.woff[NET_D * LAYERS + 8];
        const MaskWords mk = load_mask_words(MK + (size_t)7 * nt32 * 256, tile, wave, lane);
        for_quads_noacc<RTC, 2>(RT0, 2 * wave, lane, [&](int row, int col) {
            const float w0 = W8[col], w1 = W8[256 + col], w2 = W8[512 + col];
            const int qi = ((row >> 5) * 2 + ((col >> 5) & 1)) * 4 + ((row & 31) >> 3);
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = mask_bit(mk, qi, i, hi) ? g8[row + i] * w0 + g8[64 + row + i] * w1 + g8[128 + row + i] * w2 : 0.f;
            lds_store_quad(mainT, col, row, v);
            if (save) g_store_quad(R + (size_t)7 * Mp * 256, grow0, 256, row, col, v);
        });
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
        const MaskWords mk = load_mask_words(MK + (size_t)(l - 1) * nt32 * 256, tile, wave, lane);     // in flight during the GEMM
        f32x16 acc[RTC][2];
        acc_zero(acc);
        if (l == 3) gemm_seg<26, RTC, 2>(acc, mainT, a.packed + a.tb.segoff[DR3], RT0, 2 * wave, lane);
        else gemm_seg<32, RTC, 2>(acc, mainT, a.packed + a.tb.segoff[DR0 + l], RT0, 2 * wave, lane);
        __syncthreads();
        float* Rl = R + (size_t)(l - 1) * Mp * 256;
        // (the skip layer's per-lane test and ``save`` are compile-time in the epilogue: see deform_fwd_tile)
        auto epi = [&](auto SKIP, auto SAVE) {
            for_quads_off(acc, qo, RT0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int off, int ni) {
                const int qi = (((row >> 5) * 2 + ni) << 2) + ((row >> 3) & 3);      // (global row tile: the mask words are per 64-row tile)
                if (decltype(SKIP)::value && col >= 204) {
                    lds_store_quad(aux, col - 204, row, v);          // skip: adjoint of the encoding part of layer 4's input
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = 0.f;          // layer 3 has 204 outputs
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = mask_keep(v[i], mk, qi, i, hi);
                }
                lds_store_quad_at(mainT, off, v);
                if (decltype(SAVE)::value) g_store_quad(Rl, grow0, 256, row, col, v);
            });
        };
        if (l == 4 && wave == 3) { if (save) epi(std::true_type{}, std::true_type{}); else epi(std::true_type{}, std::false_type{}); }
        else { if (save) epi(std::false_type{}, std::true_type{}); else epi(std::false_type{}, std::false_type{}); }
        __syncthreads();
    }
    {   // adjoint of the encoding input: += W_0^T r_0
        // (a half tile: waves 0 / 1 take its one row tile, waves 2 / 3 have none)
        if (!HALF || (wave >> 1) == 0) {
            const int art = HALF ? RT0 : (wave >> 1);
            f32x16 accA[1][1];
            acc_zero(accA);
            gemm_seg<32, 1, 1>(accA, mainT, a.packed + a.tb.segoff[DR0], art, wave & 1, lane);
            for_quads(accA, art, wave & 1, lane, [&](int row, int col, float(&v)[4]) { if (col < 52) lds_add_quad(aux, col, row, v); });
        }
    }
    __syncthreads();
    if (tid < 192) {   // g_o[j] = g_c[j] + sum_k adj[k] * d enc_k / d x_j   (position part of the encoding, observed-space x)
        const int j = tid >> 6, row = tid & 63;
        float x[3], t, d[3];
        load_point(a.src, row0 + row, x, t, d);
        float g = g8[j * 64 + row] + aux[swz(j, row)];
        float c2 = 0.f;      // the encoding's curvature against the same adjoint (WS_CURV: second derivative of the query w.r.t. the point)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float f = (float)(1 << i);
            float s, co;
            sincosf(x[j] * f, &s, &co);
            const float as = aux[swz(enc_index(3, i, 0, j), row)], ac = aux[swz(enc_index(3, i, 1, j), row)];
            g += f * (as * co - ac * s);
            c2 += (f * f) * (as * s + ac * co);
        }
        if (!HALF || (row >> 5) == RT0) {
            wsb(a, WS_GO)[(grow0 + row) * 3 + j] = g;
            wsb(a, WS_CURV)[(grow0 + row) * 3 + j] = c2;
        }
    } else {   // the adjoint of the TIME input under the same covector: <g_c, d x_c / d t> (raw t + its six sin / cos pairs: rows 39..51)
        const int row = tid & 63;
        float x[3], t, d[3];
        load_point(a.src, row0 + row, x, t, d);
        float g = aux[swz(39, row)];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float f = (float)(1 << i);
            float s, co;
            sincosf(t * f, &s, &co);
            g += f * (aux[swz(39 + enc_index(1, i, 0, 0), row)] * co - aux[swz(39 + enc_index(1, i, 1, 0), row)] * s);
        }
        if (!HALF || (row >> 5) == RT0) wsb(a, WS_TBAR)[grow0 + row] = g;
    }
}

// -------------------------------------------------------------------------------------------------------------
// SDF network: value pass + reverse sweep.  Tile = 64 points.  HALF: one 32-row tile of a 64-row tile (see deform_vjp_tile): the
// per-row stages still run on all 64 rows of the LDS tile (the other half's rows hold whatever its inputs hold and are never stored).
template <bool HALF = false>
__device__ __forceinline__ void sdf_fwd_tile(const FwdArgs& a, const int tile_in) {
    constexpr int RTC = HALF ? 1 : 2;
    const int tile = HALF ? (tile_in >> 1) : tile_in;
    const int RT0 = HALF ? (tile_in & 1) : 0;
    auto own = [&](int row) { return !HALF || (row >> 5) == RT0; };
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    float* aux = lds + MAIN_FLOATS;        // 56 rows: enc6(x_c) (40 used), later the adjoint of the encoding (40 used)
    float* scr = aux + AUX56_FLOATS;
    float* px = scr;          // [3][64]
    float* red = aux;         // [4][1][64]  (encoding rows are dead when the last layer runs)
    float* gcv = mainT;       // [3][64]     (activation tile is dead after the last reverse GEMM)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = tile * TM;
    const size_t grow0 = (size_t)row0;
    const bool save = a.flags & PF_SAVE, deform = a.flags & PF_DEFORM, color = (a.flags & PF_COLOR) && row0 < a.M_color;
    const size_t Mp = (size_t)a.L.Mp;

    if (tid < 64) {
        if (deform) {
            const float* xc = wsb(a, WS_XC) + (grow0 + tid) * 3;
            px[tid] = xc[0]; px[64 + tid] = xc[1]; px[128 + tid] = xc[2];
        } else {
            float x[3], t, d[3];
            load_point(a.src, row0 + tid, x, t, d);
            px[tid] = x[0]; px[64 + tid] = x[1]; px[128 + tid] = x[2];
            if (own(tid)) {
                float* xc = wsb(a, WS_XC) + (grow0 + tid) * 3;
                xc[0] = x[0]; xc[1] = x[1]; xc[2] = x[2];
            }
        }
    }
    __syncthreads();
    encode3<6>(aux, 0, px, tid);
    zero_rows(aux, 39, 40, tid);
    __syncthreads();
    if (save) {
        float* S0 = wsb(a, WS_S_S0);
        const int r = tid >> 2, c4 = tid & 3;
        if (own(r))
            for (int k = c4; k < 40; k += 4) S0[(grow0 + r) * 64 + k] = aux[swz(k, r)];
    }
    float* SACT = wsb(a, WS_S_ACT);
    // (bias requested a layer ahead: inside the epilogue the 16 quads reloaded it after every store -- 16 loads, their waits and nops)
    const QuadOff<RTC> qo = quad_offsets<RTC>(RT0, 2 * wave, lane);
    auto bias2 = [&](float(&b)[2], int l) {
        const float* bias = a.weff + a.tb.boff[NET_S * LAYERS + l] + 64 * wave + (lane & 31);
        b[0] = bias[0]; b[1] = bias[32];
    };
    auto epi = [&](f32x16(&acc)[RTC][2], int l, const float(&bc)[2]) {
        float* Sl = SACT + (size_t)l * Mp * 256;
        for_quads_off(acc, qo, RT0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int off, int ni) {
            add_bias4(v, bc[ni]);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = softplus100(v[i]);
            lds_store_quad_at(mainT, off, v);
            g_store_quad_f(Sl, grow0, row, col, v);
        });
    };
    float bc[2], bn[2];
    bias2(bn, 0);
    {
        f32x16 acc[RTC][2];
        acc_zero(acc);
        bc[0] = bn[0]; bc[1] = bn[1];
        bias2(bn, 1);
        gemm_seg<5, RTC, 2>(acc, aux, a.packed + a.tb.segoff[SF0], RT0, 2 * wave, lane);
        epi(acc, 0, bc);
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        f32x16 acc[RTC][2];
        acc_zero(acc);
        bc[0] = bn[0]; bc[1] = bn[1];
        if (l < 7) bias2(bn, l + 1);
        const int seg = l <= 4 ? SF0 + l : SF0 + l + 1;
        gemm_seg<32, RTC, 2>(acc, mainT, a.packed + a.tb.segoff[seg], RT0, 2 * wave, lane);
        if (l == 4) gemm_seg<5, RTC, 2>(acc, aux, a.packed + a.tb.segoff[SF4A], RT0, 2 * wave, lane);
        __syncthreads();
        epi(acc, l, bc);
        __syncthreads();
    }
    // last layer: 256 geometry features (MFMA) + sdf (row 0, VALU)
    if (color) {
        f32x16 acc[RTC][2];
        acc_zero(acc);
        gemm_seg<32, RTC, 2>(acc, mainT, a.packed + a.tb.segoff[SF8F], RT0, 2 * wave, lane);
        const float* bias = a.weff + a.tb.boff[NET_S * LAYERS + 8] + 1;
        float* feat = wsb(a, WS_FEAT);
        for_quads(acc, RT0, 2 * wave, lane, [&](int row, int col, float(&v)[4]) {
            const float b = bias[col];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += b;
            g_store_quad(feat, grow0, 256, row, col, v);
        });
    }
    smalln_partial<1>(mainT, a.weff + a.tb.woff[NET_S * LAYERS + 8], 256, red, tid);
    __syncthreads();
    if (tid < 64 && own(tid)) wsb(a, WS_SDF)[grow0 + tid] = smalln_reduce<1>(red, 0, tid) + a.weff[a.tb.boff[NET_S * LAYERS + 8]];

    // ---- reverse sweep: rho_l = d sdf / d z_l ----
    float* RHO = wsb(a, WS_S_RHO);
    {   // rho_7 = softplus'(z_7) * W8[0,:]   (mainT still holds s_8 = softplus(z_7))
        const float* w8 = a.weff + a.tb.woff[NET_S * LAYERS + 8];
        const float w8c[2] = {w8[64 * wave + (lane & 31)], w8[64 * wave + 32 + (lane & 31)]};
        auto rho7 = [&](auto SAVE) {
            f32x16 dummy[RTC][2];
            for_quads_off(dummy, qo, RT0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int off, int ni) {
                const float4 t = *reinterpret_cast<const float4*>(mainT + off);
                const float sv[4] = {t.x, t.y, t.z, t.w};
                softplus100_grad_from_s4(sv, v);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] *= w8c[ni];
                lds_store_quad_at(mainT, off, v);
                if (decltype(SAVE)::value) g_store_quad_f(RHO + (size_t)7 * Mp * 256, grow0, row, col, v);
            });
        };
        if (save) rho7(std::true_type{}); else rho7(std::false_type{});
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
        const float* Sl = SACT + (size_t)(l - 1) * Mp * 256;                                      // s_l
        f32x16 acc[RTC][2];
        acc_zero(acc);
        const int seg = l <= 4 ? SR0 + l : SR0 + l + 1;
        gemm_seg<32, RTC, 2>(acc, mainT, a.packed + a.tb.segoff[seg], RT0, 2 * wave, lane);       // adjoint of s_l
        // (the 40-column adjoint of the encoding: one [32 x 32] block per wave of a full tile; waves 0 / 1 of a half tile take its row tile)
        const bool has_a = !HALF || (wave >> 1) == 0;
        const int art = HALF ? RT0 : (wave >> 1);
        f32x16 accA[1][1];
        if (l == 4 && has_a) {
            acc_zero(accA);
            gemm_seg<32, 1, 1>(accA, mainT, a.packed + a.tb.segoff[SR4A], art, wave & 1, lane);   // adjoint of the skip's encoding part
        }
        float S[RTC * 8][4];                                     // all quads requested at once: one memory round trip
        prefetch_quads_f<RTC, 2>(S, Sl, grow0, RT0, 2 * wave, lane);
        __syncthreads();
        auto repi = [&](auto SAVE) {
            for_quads_off(acc, qo, RT0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int off, int ni) {
                const int qi = ((((row >> 5) - RT0) * 2 + ni) << 2) + ((row >> 3) & 3);
                float dphi[4];
                softplus100_grad_from_s4(S[qi], dphi);
#pragma unroll
                for (int h = 0; h < 2; ++h) {      // packed pairs
                    const f32x2p o = f32x2p{v[2 * h], v[2 * h + 1]} * f32x2p{dphi[2 * h], dphi[2 * h + 1]};
                    v[2 * h] = o[0]; v[2 * h + 1] = o[1];
                }
                lds_store_quad_at(mainT, off, v);
                if (decltype(SAVE)::value) g_store_quad_f(RHO + (size_t)(l - 1) * Mp * 256, grow0, row, col, v);
            });
        };
        if (save) repi(std::true_type{}); else repi(std::false_type{});
        if (l == 4 && has_a)
            for_quads(accA, art, wave & 1, lane, [&](int row, int col, float(&v)[4]) { if (col < 40) lds_store_quad(aux, col, row, v); });
        __syncthreads();
    }
    if (!HALF || (wave >> 1) == 0) {
        const int art = HALF ? RT0 : (wave >> 1);
        f32x16 accA[1][1];
        acc_zero(accA);
        gemm_seg<32, 1, 1>(accA, mainT, a.packed + a.tb.segoff[SR0], art, wave & 1, lane);
        for_quads(accA, art, wave & 1, lane, [&](int row, int col, float(&v)[4]) { if (col < 40) lds_add_quad(aux, col, row, v); });
    }
    __syncthreads();
    if (save) {
        float* AE = wsb(a, WS_S_ADJEPS);
        const int r = tid >> 2, c4 = tid & 3;
        if (own(r))
            for (int k = c4; k < 40; k += 4) AE[(grow0 + r) * 64 + k] = aux[swz(k, r)];
    }
    if (tid < 192) {   // g_c[j] = sum_k adj_eps[k] * d enc_k / d x_j
        const int j = tid >> 6, row = tid & 63;
        const float x = px[j * 64 + row];
        float g = aux[swz(j, row)];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float f = (float)(1 << i);
            float s, co;
            sincosf(x * f, &s, &co);
            g += f * (aux[swz(enc_index(3, i, 0, j), row)] * co - aux[swz(enc_index(3, i, 1, j), row)] * s);
        }
        gcv[j * 64 + row] = g;
    }
    __syncthreads();
    if (tid < 64 && own(tid)) {
        const size_t gp = grow0 + tid;
        const float g0 = gcv[tid], g1 = gcv[64 + tid], g2 = gcv[128 + tid];
        float* gc = wsb(a, WS_GC) + gp * 3;
        gc[0] = g0; gc[1] = g1; gc[2] = g2;
        if (!deform) {      // with a deformation network g_o = J^T g_c is the VJP sweep (deform_vjp_tile)
            float* go = wsb(a, WS_GO) + gp * 3;
            go[0] = g0; go[1] = g1; go[2] = g2;
        }
    }
}

// -------------------------------------------------------------------------------------------------------------
// colour network.  Tile = 64 points.  Lean LDS carve (activation tile + 5 KB): the 93-wide small part of the input is
// staged through the activation tile itself (and re-staged from HBM at the skip layer), so two workgroups fit per CU.
constexpr int CFWD_LDS_BYTES = (MAIN_FLOATS + 1344) * 4;   // 70 912 B
__device__ __forceinline__ void color_fwd_tile(const FwdArgs& a, const int tile) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    float* scr = lds + MAIN_FLOATS;
    float* px = scr;          // [3][64] x_c
    float* pd = scr + 192;    // [3][64] d_c
    float* pg = scr + 384;    // [3][64] g_c
    float* red = scr + 576;   // [4][3][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = tile * TM;
    const size_t grow0 = (size_t)row0;
    const bool save = a.flags & PF_SAVE, deform = a.flags & PF_DEFORM;
    const size_t Mp = (size_t)a.L.Mp;
    const float* feat = wsb(a, WS_FEAT);
    float* CIN = wsb(a, WS_C_IN);

    if (tid < 64) {
        const size_t gp = grow0 + tid;
        float x[3], t, d[3];
        load_point(a.src, row0 + tid, x, t, d);
        const float* xc = wsb(a, WS_XC) + gp * 3;
        const float* gc = wsb(a, WS_GC) + gp * 3;
        px[tid] = xc[0]; px[64 + tid] = xc[1]; px[128 + tid] = xc[2];
        pg[tid] = gc[0]; pg[64 + tid] = gc[1]; pg[128 + tid] = gc[2];
        float v0 = d[0], v1 = d[1], v2 = d[2];
        if (deform) {
            const float* v = wsb(a, WS_V) + gp * 3;      // d_c = J d / (|J d| + 1e-10)   endosurf.py:684-685
            v0 = v[0]; v1 = v[1]; v2 = v[2];
        }
        const float inv = (a.flags & PF_RAW_DIR) ? 1.f : 1.f / (sqrtf(v0 * v0 + v1 * v1 + v2 * v2) + 1e-10f);
        pd[tid] = v0 * inv; pd[64 + tid] = v1 * inv; pd[128 + tid] = v2 * inv;
    }
    __syncthreads();
    // small part [enc10(x_c) 63 | g_c 3 | enc4(d_c) 27 | 0 0 0] into rows 0..95 of the activation tile
    encode3<10>(mainT, 0, px, tid);
    if (tid < 192) mainT[swz(63 + (tid >> 6), tid & 63)] = pg[tid];
    encode3<4>(mainT, 66, pd, tid);
    zero_rows(mainT, 93, 96, tid);
    __syncthreads();
    {
        const int r = tid >> 2, c4 = tid & 3;
        for (int k = c4; k < 96; k += 4) CIN[(grow0 + r) * 128 + k] = mainT[swz(k, r)];
    }
    float* CH = wsb(a, WS_C_H);
    // lean epilogue (see deform_fwd_tile): bias requested a layer ahead, ``save`` compile-time, pinned LDS offsets, the mask bits of a
    // quad gathered in 32-bit halves (quads 0..7 | 8..15) instead of 64-bit shifts
    const QuadOff<2> qo = quad_offsets<2>(0, 2 * wave, lane);
    auto bias2 = [&](float(&b)[2], int l) {
        const float* bias = a.weff + a.tb.boff[NET_C * LAYERS + l] + 64 * wave + (lane & 31);
        b[0] = bias[0]; b[1] = bias[32];
    };
    auto epi_impl = [&](f32x16(&acc)[2][2], int l, const float(&bc)[2], auto SAVE) {
        float* Hl = CH + (size_t)l * Mp * 256;
        unsigned blo = 0, bhi = 0;
        for_quads_off(acc, qo, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int off, int ni) {
            const int qi = (((row >> 5) * 2 + ni) << 2) + ((row >> 3) & 3);
            add_bias4(v, bc[ni]);
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                m |= (v[i] > 0.f ? 1u : 0u) << (4 * (qi & 7) + i);
                v[i] = relu1(v[i]);
            }
            if (qi < 8) blo |= m; else bhi |= m;
            lds_store_quad_at(mainT, off, v);
            if (decltype(SAVE)::value) g_store_quad(Hl, grow0, 256, row, col, v);
        });
        if (decltype(SAVE)::value)      // the ReLU masks of the backward sweep: 2 words per thread instead of 64 activations
            reinterpret_cast<unsigned long long*>(wsb(a, WS_C_MASK))[((size_t)l * (Mp / 64) + tile) * 256 + tid] = (unsigned long long)bhi << 32 | blo;
    };
    auto epi = [&](f32x16(&acc)[2][2], int l, const float(&bc)[2]) {
        if (save) epi_impl(acc, l, bc, std::true_type{});
        else epi_impl(acc, l, bc, std::false_type{});
    };
    float bc[2], bn[2];
    bias2(bn, 0);
    {
        f32x16 acc[2][2];
        acc_zero(acc);
        gemm_seg<12, 2, 2>(acc, mainT, a.packed + a.tb.segoff[CF0S], 0, 2 * wave, lane);
        __syncthreads();
        load_tile_256(mainT, feat, grow0, 256, tid);
        __syncthreads();
        gemm_seg<32, 2, 2>(acc, mainT, a.packed + a.tb.segoff[CF0F], 0, 2 * wave, lane);
        __syncthreads();
        bc[0] = bn[0]; bc[1] = bn[1];
        bias2(bn, 1);
        epi(acc, 0, bc);
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        f32x16 acc[2][2];
        acc_zero(acc);
        bc[0] = bn[0]; bc[1] = bn[1];
        if (l < 7) bias2(bn, l + 1);
        if (l != 4) {
            const int seg = l < 4 ? CF1 + (l - 1) : CF5 + (l - 5);
            gemm_seg<32, 2, 2>(acc, mainT, a.packed + a.tb.segoff[seg], 0, 2 * wave, lane);
        } else {   // skip layer: input = [h(256) | small(93) | feat(256)] / sqrt2, staged through the tile one part at a time
            gemm_seg<32, 2, 2>(acc, mainT, a.packed + a.tb.segoff[CF4H], 0, 2 * wave, lane);
            __syncthreads();
            load_tile<96>(mainT, CIN, grow0, 128, tid);
            __syncthreads();
            gemm_seg<12, 2, 2>(acc, mainT, a.packed + a.tb.segoff[CF4S], 0, 2 * wave, lane);
            __syncthreads();
            load_tile_256(mainT, feat, grow0, 256, tid);
            __syncthreads();
            gemm_seg<32, 2, 2>(acc, mainT, a.packed + a.tb.segoff[CF4F], 0, 2 * wave, lane);
        }
        __syncthreads();
        epi(acc, l, bc);
        __syncthreads();
    }
    smalln_partial<3>(mainT, a.weff + a.tb.woff[NET_C * LAYERS + 8], 256, red, tid);
    __syncthreads();
    if (tid < 192) {
        const int i = tid >> 6, row = tid & 63;
        const float y = smalln_reduce<3>(red, i, row) + a.weff[a.tb.boff[NET_C * LAYERS + 8] + i];
        wsb(a, WS_RGB)[(grow0 + row) * 3 + i] = 1.f / (1.f + expf(-y));
    }
}

// -------------------------------------------------------------------------------------------------------------
}  // namespace es
