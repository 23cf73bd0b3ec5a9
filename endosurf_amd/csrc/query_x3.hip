// Opt-in SPLIT-PRECISION variant of the fused no-grad SDF query (query.hip: sdf(x + deform(x, t)), reference
// EndoSurfNet.get_sdf_from_observed_space, endosurf.py:570-579).
//
// fp32 MFMA runs at 1/16 of the bf16 matrix rate on gfx950.  Every fp32 operand is therefore split into three bf16 planes
//      x = x_h + x_m + x_l,   x_h = bf16(x), x_m = bf16(x - x_h), x_l = bf16(x - x_h - x_m)      (round to nearest even)
// which is exact for a 24-bit significand, and the product of two fp32 numbers is formed from the six partial products whose
// weight is >= 2^-16 of the leading one:
//      x w ~= x_h w_h + (x_h w_m + x_m w_h) + (x_h w_l + x_m w_m + x_l w_h)          (dropped terms <= 2^-23 |x w|)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 6 bf16 MFMAs replace 8 fp32 MFMAs of the same tile at 1/2 the cycles each,
// i.e. 2.67x the fp32 matrix rate at fp32-class accuracy (the tests hold this path to the SAME budgets as the fp32 path).
// NOT the default: the fp32 kernels remain the product path and the headline benchmark; this mode is requested explicitly
// (ES_SPLIT_BF16=1 / render_cfg["split_precision"]) and reported as a separate bench line.
//
// Formulation (transposed w.r.t. the fp32 chain kernels so that the epilogue's registers ARE the next layer's operand):
//      Y^T [256 features][64 points] = W [256][K] . X^T [K][64 points]
//   * MFMA A operand = weight fragments (row = output feature, 8 consecutive k per lane), pre-split and pre-packed by
//     k_pack_x3, streamed global -> VGPR three k-steps ahead;
//   * MFMA B operand = activations in LDS as three bf16 planes [k/8][point][8]: one conflict-free ds_read_b128 per fragment;
//   * D: lane (lo, hi) holds, for point lo, the features 8q + 4hi .. +3 of its 32-feature block: after bias + activation they
//     are split and stored as ONE ds_write_b64 per plane -- exactly half of a [k/8][point] unit of the next layer's operand.
// One workgroup = 512 threads = 8 waves (2 per SIMD) owns 64 points; wave w computes features 32w .. 32w+31 for all of them.
#include "chain_common.h"
#include "launch.h"
#include "tabs.h"
#include "timing.h"

namespace es {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int X3_SEGS[] = {DF0, DF1, DF2, DF3, DF4, DF5, DF6, DF7, SF0, SF1, SF2, SF3, SF4M, SF4A, SF5, SF6, SF7};
constexpr int X3_COUNT = sizeof(X3_SEGS) / sizeof(int);
constexpr int x3_kg(int i) { return cdiv(SEGS[X3_SEGS[i]].kreal, 16); }
// segment i: [8 feature blocks][kg][3 planes][64 lanes] units of 16 B
constexpr size_t x3_off16(int i) {
    size_t off = 0;
    for (int k = 0; k < i; ++k) off += (size_t)8 * x3_kg(k) * 3 * 64;
    return off;
}
constexpr size_t X3_UNITS = x3_off16(X3_COUNT);
struct X3Tabs { unsigned off[X3_COUNT]; };
static X3Tabs make_x3_tabs() {
    X3Tabs t;
    for (int i = 0; i < X3_COUNT; ++i) t.off[i] = (unsigned)x3_off16(i);
    return t;
}
constexpr int x3_index(int seg) {
    for (int i = 0; i < X3_COUNT; ++i)
        if (X3_SEGS[i] == seg) return i;
    return -1;
}
constexpr int X3_DF0 = x3_index(DF0), X3_SF0 = x3_index(SF0), X3_SF4A = x3_index(SF4A);
static_assert(X3_DF0 == 0 && X3_SF0 == 8 && X3_SF4A == X3_SF0 + 5 && x3_index(SF5) == X3_SF0 + 6, "segment order the kernel indexes by");

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {      // round-to-nearest-even pack of two floats
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// (x0, x1) -> three packed bf16 pairs with x = h + m + l exactly (24-bit significand)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(x0, x1);
    float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    r0 -= __uint_as_float(m << 16); r1 -= __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(r0, r1);
}

// ---- weight packing ---------------------------------------------------------------------------------------------------
// one thread per (segment, feature block, k-group, lane): lane l of the fragment holds W[32 fb + (l & 31)][16 kg + 8 (l >> 5) + j],
// j = 0..7, as three bf16x8 planes.  Forward orientation of the 17 query segments only (skip scale folded in).
struct X3PackArgs { int seg[X3_COUNT], net[X3_COUNT], layer[X3_COUNT], row0[X3_COUNT], col0[X3_COUNT], kreal[X3_COUNT], nreal[X3_COUNT],
                    skip[X3_COUNT], K[X3_COUNT], woff[X3_COUNT]; unsigned off[X3_COUNT + 1]; };
__global__ __launch_bounds__(256) void k_pack_x3(const float* __restrict__ weff, u32x4* __restrict__ out, X3PackArgs a, int first_net) {
    const unsigned idx = blockIdx.x * 256 + threadIdx.x;           // (fragment, lane), fragment = (fb * kg + g) of a segment
    if (idx >= a.off[X3_COUNT] / 3) return;
    int si = 0;
#pragma unroll 1
    for (int i = 1; i < X3_COUNT; ++i)
        if (idx >= a.off[i] / 3) si = i;
    if (a.net[si] < first_net) return;
    const unsigned rel = idx - a.off[si] / 3;
    const int lane = rel & 63;
    const int kgn = (a.kreal[si] + 15) / 16;
    const int g = (rel >> 6) % kgn, fb = (rel >> 6) / kgn;
    const float* W = weff + a.woff[si];
    const float sc = a.skip[si] ? INV_SQRT2 : 1.f;
    const int n = 32 * fb + (lane & 31);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * g + 8 * (lane >> 5) + j;
        v[j] = (k < a.kreal[si] && n < a.nreal[si]) ? sc * W[(size_t)(a.row0[si] + n) * a.K[si] + a.col0[si] + k] : 0.f;
    }
    u32x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned hh, mm, ll;
        split_pair(v[2 * j], v[2 * j + 1], hh, mm, ll);
        h[j] = hh; m[j] = mm; l[j] = ll;
    }
    u32x4* o = out + a.off[si] + (size_t)((fb * kgn + g) * 3) * 64 + lane;
    o[0] = h; o[64] = m; o[128] = l;
}

// ---- LDS operand planes -------------------------------------------------------------------------------------------------
// plane p of a K-wide operand: [K/8][PTS points] units of 16 B (8 consecutive k of one point).
// Tile shapes: PTS = 64 with 8 waves (wave = 32 features x 64 points; the one the host launches), 16 waves (32 x 32 per wave, four
// waves per SIMD) or 4 waves (64 x 64 per wave, one per SIMD) -- 127 KB of LDS, one workgroup per CU -- and PTS = 32 (4 waves,
// wave = 64 features x 32 points, 64 KB: TWO workgroups per CU whose GEMM and epilogue phases interleave on the SIMDs, at twice the
// weight traffic from L2).  Measured (163 840 points of a training step): 1.90 ms vs 2.70 ms -- the 384 KB of split weights per layer
// and tile make the short tile L2-bound (62 B/clk/CU at full MFMA rate), so the host launches PTS = 64 (-DX3_PTS=32 builds the other).
#ifdef X3_PROFILE        // dev builds only (-DX3_PROFILE): cycle stamps of block 0 / wave 0 at the phase boundaries of every layer
__device__ long long x3_prof[512];
#define X3_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) x3_prof[i] = __builtin_readcyclecounter(); } while (0)
#else
#define X3_STAMP(i) do {} while (0)
#endif
#ifndef X3_PTS
#define X3_PTS 64
#endif
// Cycle stamps (-DX3_PROFILE, tools/x3_profile.py) of one 256-wide softplus layer, 8 waves: GEMM 14.5 k cycles for wave 0 + 3.9 k waiting
// at the barrier for its SIMD partner (12.3 k of MFMA issue for the pair: the GEMM phase is ~66 % efficient), epilogue 4.7 k + 2.4 k
// waiting for the partner's: 25.6 k per layer, 48 % of it MFMA.  16 waves: the GEMM phase grows to 22.5 k (twice the weight-fragment
// requests: L1 delivers 64 B/clk hit or miss) -> 2.15 ms instead of 1.89 ms per training step; 4 waves: 26.6 k (one wave per SIMD cannot
// cover its own operand latencies) -> 2.43 ms.  Requesting the bias values before the GEMM and a rolled (truly prefetching) weight
// pipeline change nothing (the partner wave already hides those latencies); softplus' exp / log are 0.2 of the 1.9 ms.
#ifndef X3_WAVES
#define X3_WAVES (X3_PTS == 64 ? 8 : 4)
#endif
template <int PTS>
struct X3Cfg {
    static constexpr int WAVES = X3_WAVES;              // PTS = 64: 8 or 16;  PTS = 32: 4
    static constexpr int BLOCKS = 8 * (PTS / 32) / WAVES;   // 32 x 32 output blocks per wave
    static constexpr int PB = BLOCKS >= 2 && PTS == 64 ? 2 : 1;     // point blocks of a wave tile
    static constexpr int FB = BLOCKS / PB;                          // feature blocks of a wave tile
    static constexpr int FGROUPS = 8 / FB;              // waves along the feature axis
    static constexpr int THREADS = WAVES * 64;
    static constexpr int MAIN_PLANE = 32 * PTS * 16;
    static constexpr int ENC_PLANE = 8 * PTS * 16;
    static constexpr int NPARTS = THREADS / PTS < 8 ? THREADS / PTS : 8;      // thread groups (of PTS threads) of the per-point VALU stages
    static constexpr int VTHREADS = NPARTS * PTS;
    static constexpr int LDS_BYTES = 3 * MAIN_PLANE + 3 * ENC_PLANE + (4 * PTS + 8 * 3 * PTS) * 4;
    static_assert(BLOCKS >= 1 && FB * PB == BLOCKS && VTHREADS <= THREADS && NPARTS >= 3, "tile shape");
    __device__ static int fb0(int wave) { return (wave % FGROUPS) * FB; }
    __device__ static int pb0(int wave) { return (wave / FGROUPS) * PB; }
};

template <int PTS>
__device__ __forceinline__ void put_x3(unsigned char* planes, int plane_bytes, int k, int p, float v) {      // one element
    const unsigned h = cvt_pk_bf16(v, 0.f);
    const float r1 = v - __uint_as_float(h << 16);
    const unsigned m = cvt_pk_bf16(r1, 0.f);
    const unsigned l = cvt_pk_bf16(r1 - __uint_as_float(m << 16), 0.f);
    const int o = ((k >> 3) * PTS + p) * 16 + (k & 7) * 2;
    *reinterpret_cast<unsigned short*>(planes + o) = (unsigned short)h;
    *reinterpret_cast<unsigned short*>(planes + plane_bytes + o) = (unsigned short)m;
    *reinterpret_cast<unsigned short*>(planes + 2 * plane_bytes + o) = (unsigned short)l;
}
template <int PTS>
__device__ __forceinline__ float get_x3(const unsigned char* planes, int plane_bytes, int k, int p) {
    const int o = ((k >> 3) * PTS + p) * 16 + (k & 7) * 2;
    const unsigned h = *reinterpret_cast<const unsigned short*>(planes + o), m = *reinterpret_cast<const unsigned short*>(planes + plane_bytes + o),
                   l = *reinterpret_cast<const unsigned short*>(planes + 2 * plane_bytes + o);
    return __uint_as_float(h << 16) + (__uint_as_float(m << 16) + __uint_as_float(l << 16));
}

// acc[fi][pb] += W[features 32 (FB wave + fi) ..][0 .. 16 KG) . X^T[0 .. 16 KG)[points 32 pb ..]     (six partial products per tile)
template <int KG, int PTS>
__device__ __forceinline__ void gemm_x3(f32x16 (&acc)[X3Cfg<PTS>::FB][X3Cfg<PTS>::PB], const u32x4* __restrict__ W, const unsigned char* X,
                                        int plane_bytes, int wave, int lane) {
    constexpr int FB = X3Cfg<PTS>::FB, PB = X3Cfg<PTS>::PB;
    const u32x4* wl = W + (size_t)X3Cfg<PTS>::fb0(wave) * KG * 3 * 64 + lane;
    const unsigned char* xb = X + ((lane >> 5) * PTS + 32 * X3Cfg<PTS>::pb0(wave) + (lane & 31)) * 16;
    constexpr int PF = 3;                       // weight fragments in flight: three k-steps ahead (L2 latency)
    u32x4 a[PF + 1][FB][3], b[2][PB][3];
    auto load_a = [&](u32x4(&d)[FB][3], int g) {
#pragma unroll
        for (int fi = 0; fi < FB; ++fi)
#pragma unroll
            for (int p = 0; p < 3; ++p) d[fi][p] = wl[(size_t)((fi * KG + g) * 3 + p) * 64];
    };
    auto load_b = [&](u32x4(&d)[PB][3], int g) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int p = 0; p < 3; ++p) d[pb][p] = *reinterpret_cast<const u32x4*>(xb + p * plane_bytes + (2 * g) * PTS * 16 + pb * 32 * 16);
    };
#pragma unroll
    for (int s = 0; s < PF && s < KG; ++s) load_a(a[s], s);
    load_b(b[0], 0);
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        if (g + PF < KG) load_a(a[(g + PF) % (PF + 1)], g + PF);
        if (g + 1 < KG) load_b(b[(g + 1) & 1], g + 1);
        const u32x4(&aa)[FB][3] = a[g % (PF + 1)];
        const u32x4(&bb)[PB][3] = b[g & 1];
        // smallest terms first: (l,h) (m,m) (h,l) | (m,h) (h,m) | (h,h); six dependent MFMAs per accumulator in a row (alternating the
        // accumulators per term was measured and is slower in this loop: 1.87 -> 2.11 ms per training step)
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int fi = 0; fi < FB; ++fi)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    acc[fi][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aa[fi][TA[t]]),
                                                                         __builtin_bit_cast(bf16x8, bb[pb][TB[t]]), acc[fi][pb], 0, 0, 0);
    }
}

// epilogue visitor: f(f0, p, v[4]) with v = features f0 .. f0+3 (f0 = 32 (FB wave + fi) + 8 q + 4 hi) of point p = 32 pb + lo
template <int PTS, class F>
__device__ __forceinline__ void for_quads_x3(f32x16 (&acc)[X3Cfg<PTS>::FB][X3Cfg<PTS>::PB], int wave, int lane, F&& f) {
    constexpr int FB = X3Cfg<PTS>::FB, PB = X3Cfg<PTS>::PB;
    const int lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int fi = 0; fi < FB; ++fi)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4] = {acc[fi][pb][4 * q + 0], acc[fi][pb][4 * q + 1], acc[fi][pb][4 * q + 2], acc[fi][pb][4 * q + 3]};
                f(32 * (X3Cfg<PTS>::fb0(wave) + fi) + 8 * q + 4 * hi, 32 * (X3Cfg<PTS>::pb0(wave) + pb) + lo, v);
            }
}
// store features f0..f0+3 of point p into the three main planes (half of one [k/8][point] unit: ds_write_b64)
template <int PTS>
__device__ __forceinline__ void store_quad_x3(unsigned char* X, int f0, int p, const float (&v)[4]) {
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair(v[0], v[1], h0, m0, l0);
    split_pair(v[2], v[3], h1, m1, l1);
    const int o = ((f0 >> 3) * PTS + p) * 16 + (f0 & 7) * 2;
    *reinterpret_cast<u32x2*>(X + o) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(X + X3Cfg<PTS>::MAIN_PLANE + o) = u32x2{m0, m1};
    *reinterpret_cast<u32x2*>(X + 2 * X3Cfg<PTS>::MAIN_PLANE + o) = u32x2{l0, l1};
}
template <int FB, int PB>
__device__ __forceinline__ void accx_zero(f32x16 (&acc)[FB][PB]) {
#pragma unroll
    for (int i = 0; i < FB; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// out[i][p] = sum_k Wrows[i][k] x[p][k] over the 256-wide main planes: 8 thread groups x 32 k each
template <int NOUT, int PTS>
__device__ __forceinline__ void smalln_x3(const unsigned char* X, const float* __restrict__ Wrows, float* red, int tid) {
    if (tid >= X3Cfg<PTS>::VTHREADS) return;
    const int p = tid % PTS, part = tid / PTS;
    float s[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) s[i] = 0.f;
    constexpr int KP = 256 / X3Cfg<PTS>::NPARTS;
#pragma unroll 4
    for (int kk = 0; kk < KP; ++kk) {
        const int k = KP * part + kk;
        const float x = get_x3<PTS>(X, X3Cfg<PTS>::MAIN_PLANE, k, p);
#pragma unroll
        for (int i = 0; i < NOUT; ++i) s[i] = fmaf(Wrows[i * 256 + k], x, s[i]);
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) red[(part * NOUT + i) * PTS + p] = s[i];
}
template <int NOUT, int PTS>
__device__ __forceinline__ float smalln_x3_reduce(const float* red, int i, int p) {
    float s = 0.f;
#pragma unroll
    for (int part = 0; part < X3Cfg<PTS>::NPARTS; ++part) s += red[(part * NOUT + i) * PTS + p];
    return s;
}

template <int L, int PTS>
__device__ __forceinline__ void encode3_x3(unsigned char* E, int kbase, const float* px, int tid) {
    if (tid >= X3Cfg<PTS>::VTHREADS) return;
    const int p = tid % PTS, part = tid / PTS;
    for (int item = part; item < 3 * L; item += X3Cfg<PTS>::NPARTS) {
        const int c = item % 3, i = item / 3;
        float s, co;
        sincosf(px[c * PTS + p] * (float)(1 << i), &s, &co);
        put_x3<PTS>(E, X3Cfg<PTS>::ENC_PLANE, kbase + enc_index(3, i, 0, c), p, s);
        put_x3<PTS>(E, X3Cfg<PTS>::ENC_PLANE, kbase + enc_index(3, i, 1, c), p, co);
    }
    if (part == X3Cfg<PTS>::NPARTS - 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) put_x3<PTS>(E, X3Cfg<PTS>::ENC_PLANE, kbase + c, p, px[c * PTS + p]);
    }
}
template <int L, int PTS>
__device__ __forceinline__ void encode1_x3(unsigned char* E, int kbase, const float* pt, int tid) {
    if (tid >= X3Cfg<PTS>::VTHREADS) return;
    const int p = tid % PTS, part = tid / PTS;
    for (int i = part; i < L; i += X3Cfg<PTS>::NPARTS) {
        float s, co;
        sincosf(pt[p] * (float)(1 << i), &s, &co);
        put_x3<PTS>(E, X3Cfg<PTS>::ENC_PLANE, kbase + enc_index(1, i, 0, 0), p, s);
        put_x3<PTS>(E, X3Cfg<PTS>::ENC_PLANE, kbase + enc_index(1, i, 1, 0), p, co);
    }
    if (part == X3Cfg<PTS>::NPARTS - 2) put_x3<PTS>(E, X3Cfg<PTS>::ENC_PLANE, kbase, p, pt[p]);
}
template <int PTS>
__device__ __forceinline__ void zero_enc_x3(unsigned char* E, int tid) {       // all three encoding planes (padding k must read as 0)
    for (int i = tid; i < 3 * X3Cfg<PTS>::ENC_PLANE / 16; i += X3Cfg<PTS>::THREADS) reinterpret_cast<u32x4*>(E)[i] = u32x4{0u, 0u, 0u, 0u};
}

template <bool DEFORM, int PTS>
__global__ __launch_bounds__(X3Cfg<PTS>::THREADS, 1) void k_query_sdf_x3(PointSrc src, Tabs tb, X3Tabs xt, const u32x4* __restrict__ packed,
                                                                       const float* __restrict__ weff, float* __restrict__ sdf_out, int ld_out,
                                                                       const int* __restrict__ ray_done) {
    using Cfg = X3Cfg<PTS>;
    constexpr int FB = Cfg::FB, PB = Cfg::PB, MAIN_PLANE = Cfg::MAIN_PLANE, ENC_PLANE = Cfg::ENC_PLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    unsigned char* X = lds3;                                  // main activation planes
    unsigned char* E = lds3 + 3 * MAIN_PLANE;                 // encoding planes
    float* scr = reinterpret_cast<float*>(E + 3 * ENC_PLANE);
    float* px = scr;               // [3][PTS]
    float* pt = scr + 3 * PTS;     // [PTS]
    float* red = scr + 4 * PTS;    // [8][<=3][PTS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * PTS;
    if (ray_done != nullptr) {      // block-wise ray marching: a tile whose rays already have their first sign change is skipped
        const int r_first = row0 / src.n_per_ray, r_last = min(row0 + PTS - 1, src.M - 1) / src.n_per_ray;
        bool all_done = true;
        for (int r = r_first; r <= r_last; ++r) all_done = all_done && ray_done[r] != 0;
        if (all_done) return;       // workgroup-uniform
    }
    X3_STAMP(0);
    if (tid < PTS) {
        float x[3], t, d[3];
        load_point(src, row0 + tid, x, t, d);
        px[tid] = x[0]; px[PTS + tid] = x[1]; px[2 * PTS + tid] = x[2]; pt[tid] = t;
    }
    zero_enc_x3<PTS>(E, tid);
    __syncthreads();

    auto W = [&](int seg) { return packed + xt.off[seg]; };
    auto bias4 = [&](const float* bias, int f0) { return make_float4(bias[f0], bias[f0 + 1], bias[f0 + 2], bias[f0 + 3]); };   // dword aligned only
    if (DEFORM) {
        // ---- deformation MLP, value only: x_c = x + MLP([enc6(x), enc6(t)]) ----
        encode3_x3<6, PTS>(E, 0, px, tid);
        encode1_x3<6, PTS>(E, 39, pt, tid);
        __syncthreads();
        {
            f32x16 acc[FB][PB];
            accx_zero(acc);
            gemm_x3<4, PTS>(acc, W(X3_DF0), E, ENC_PLANE, wave, lane);
            const float* bias = weff + tb.boff[NET_D * LAYERS + 0];
            for_quads_x3<PTS>(acc, wave, lane, [&](int f0, int p, float(&v)[4]) {
                const float4 b = bias4(bias, f0);
                v[0] = fmaxf(v[0] + b.x, 0.f); v[1] = fmaxf(v[1] + b.y, 0.f); v[2] = fmaxf(v[2] + b.z, 0.f); v[3] = fmaxf(v[3] + b.w, 0.f);
                store_quad_x3<PTS>(X, f0, p, v);
            });
        }
        __syncthreads();
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            f32x16 acc[FB][PB];
            accx_zero(acc);
            gemm_x3<16, PTS>(acc, W(X3_DF0 + l), X, MAIN_PLANE, wave, lane);
            __syncthreads();
            const float* bias = weff + tb.boff[NET_D * LAYERS + l];
            for_quads_x3<PTS>(acc, wave, lane, [&](int f0, int p, float(&v)[4]) {
                if (l == 3 && f0 >= 204) {       // IDR skip: next input = [h(204) | enc(52)] (1/sqrt2 folded into W4); 204 % 4 == 0
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = get_x3<PTS>(E, ENC_PLANE, f0 - 204 + i, p);
                } else {
                    // layer 3 has 204 outputs: its bias vector is 204 long and a quad never straddles the boundary
                    const float4 b = bias4(bias, f0);
                    v[0] = fmaxf(v[0] + b.x, 0.f); v[1] = fmaxf(v[1] + b.y, 0.f); v[2] = fmaxf(v[2] + b.z, 0.f); v[3] = fmaxf(v[3] + b.w, 0.f);
                }
                store_quad_x3<PTS>(X, f0, p, v);
            });
            __syncthreads();
        }
        smalln_x3<3, PTS>(X, weff + tb.woff[NET_D * LAYERS + 8], red, tid);
        __syncthreads();
        if (tid < 3 * PTS) {
            const int i = tid / PTS, p = tid % PTS;
            px[i * PTS + p] += smalln_x3_reduce<3, PTS>(red, i, p) + weff[tb.boff[NET_D * LAYERS + 8] + i];
        }
        __syncthreads();
        zero_enc_x3<PTS>(E, tid);
        __syncthreads();
    }

    // ---- SDF MLP on x_c, output column 0 only ----
    encode3_x3<6, PTS>(E, 0, px, tid);
    __syncthreads();
    {
        f32x16 acc[FB][PB];
        accx_zero(acc);
        gemm_x3<3, PTS>(acc, W(X3_SF0), E, ENC_PLANE, wave, lane);
        const float* bias = weff + tb.boff[NET_S * LAYERS + 0];
        for_quads_x3<PTS>(acc, wave, lane, [&](int f0, int p, float(&v)[4]) {
            const float4 b = bias4(bias, f0);
            v[0] = softplus100(v[0] + b.x); v[1] = softplus100(v[1] + b.y); v[2] = softplus100(v[2] + b.z); v[3] = softplus100(v[3] + b.w);
            store_quad_x3<PTS>(X, f0, p, v);
        });
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        f32x16 acc[FB][PB];
        accx_zero(acc);
        // X3_SEGS order: ..., SF3, SF4M, SF4A, SF5, ...
        const int si = X3_SF0 + (l <= 4 ? l : l + 1);
        X3_STAMP(100 + 4 * l);
        gemm_x3<16, PTS>(acc, W(si), X, MAIN_PLANE, wave, lane);
        if (l == 4) gemm_x3<3, PTS>(acc, W(X3_SF4A), E, ENC_PLANE, wave, lane);   // NeRF skip: + encoding part
        X3_STAMP(101 + 4 * l);
        __syncthreads();
        X3_STAMP(102 + 4 * l);
        const float* bias = weff + tb.boff[NET_S * LAYERS + l];
        for_quads_x3<PTS>(acc, wave, lane, [&](int f0, int p, float(&v)[4]) {
            const float4 b = bias4(bias, f0);
            v[0] = softplus100(v[0] + b.x); v[1] = softplus100(v[1] + b.y); v[2] = softplus100(v[2] + b.z); v[3] = softplus100(v[3] + b.w);
            store_quad_x3<PTS>(X, f0, p, v);
        });
        X3_STAMP(103 + 4 * l);
        __syncthreads();
    }
    X3_STAMP(140);
    smalln_x3<1, PTS>(X, weff + tb.woff[NET_S * LAYERS + 8], red, tid);
    __syncthreads();
    if (tid < PTS && row0 + tid < src.M) {
        const int i = row0 + tid;
        const size_t o = ld_out > 0 ? (size_t)(i / src.n_per_ray) * ld_out + (i % src.n_per_ray) : (size_t)i;   // [ray][ld_out] or flat
        sdf_out[o] = smalln_x3_reduce<1, PTS>(red, 0, tid) + weff[tb.boff[NET_S * LAYERS + 8]];
    }
}

#ifdef X3_PROFILE
extern "C" int es_debug_x3_profile(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(x3_prof), sizeof(long long) * (n < 512 ? n : 512));
}
#endif
size_t packed_x3_bytes() { return X3_UNITS * 16; }

int pack_x3(const float* weff, void* packed_x3, int use_deform, hipStream_t st) {
    if (int e = init_tables()) return e;
    X3PackArgs a;
    const Tabs tb = make_tabs();
    for (int i = 0; i < X3_COUNT; ++i) {
        const SegDesc& s = SEGS[X3_SEGS[i]];
        a.seg[i] = X3_SEGS[i]; a.net[i] = s.net; a.layer[i] = s.layer; a.row0[i] = s.row0; a.col0[i] = s.col0; a.kreal[i] = s.kreal;
        a.nreal[i] = s.nreal; a.skip[i] = s.skip_scale; a.K[i] = LAYER_K[s.net][s.layer]; a.woff[i] = tb.woff[s.net * LAYERS + s.layer];
        a.off[i] = (unsigned)x3_off16(i);
    }
    a.off[X3_COUNT] = (unsigned)X3_UNITS;
    const unsigned n = (unsigned)(X3_UNITS / 3);
    hipLaunchKernelGGL(k_pack_x3, dim3((n + 255) / 256), dim3(256), 0, st, weff, reinterpret_cast<u32x4*>(packed_x3), a, use_deform ? 0 : 1);
    return hip_last("pack_x3");
}

int query_sdf_x3(const PointSrc& src, const void* packed_x3, const float* weff, float* sdf_out, int use_deform, hipStream_t st, int ld_out,
                 const int* ray_done) {
    constexpr int PTS = X3_PTS;
    using Cfg = X3Cfg<PTS>;
    static DeviceOnce attr_done;
    if (attr_done.first()) {
        if (int e = allow_big_lds(k_query_sdf_x3<true, PTS>, Cfg::LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_query_sdf_x3<false, PTS>, Cfg::LDS_BYTES)) return e;
        attr_done.done();
    }
    if (src.M <= 0) return ST_OK;
    const Tabs tb = make_tabs();
    const X3Tabs xt = make_x3_tabs();
    const dim3 grid((src.M + PTS - 1) / PTS), block(Cfg::THREADS);
    const u32x4* pk = reinterpret_cast<const u32x4*>(packed_x3);
    ScopedTimer tm(ray_done ? KID_QUERY_EXIT : KID_QUERY_X3, src.M, st);
    if (use_deform) hipLaunchKernelGGL((k_query_sdf_x3<true, PTS>), grid, block, Cfg::LDS_BYTES, st, src, tb, xt, pk, weff, sdf_out, ld_out, ray_done);
    else hipLaunchKernelGGL((k_query_sdf_x3<false, PTS>), grid, block, Cfg::LDS_BYTES, st, src, tb, xt, pk, weff, sdf_out, ld_out, ray_done);
    return hip_last("query_sdf_x3");
}

}  // namespace es
