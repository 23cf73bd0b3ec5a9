// Shared device helpers of the OPT-IN split-precision kernels (query_x3.hip, infer_x3.hip): exact 3-way bf16 splitting, the LDS
// operand planes, the bf16-MFMA GEMM over them and the per-point VALU stages.  See query_x3.hip for the arithmetic and the layout.
#pragma once
#include "chain_common.h"

namespace es {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

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

// softplus(beta = 100) on the raw v_exp_f32 / v_log_f32 (base 2): max(z, 0) + 0.01 ln 2 log2(1 + 2^(-100 log2(e) |z|)).  The argument of
// the logarithm is in (1, 2], so none of the range handling of the library forms is needed: 6 VALU instructions instead of ~25,
// absolute error < 1e-9 (chain_common.h softplus100 is the form the fp32 kernels use).
__device__ __forceinline__ float softplus100_native(float z) {
    const float e = __builtin_amdgcn_exp2f(-144.26950408889634f * fabsf(z));
    return fmaf(0.006931471805599453f, __builtin_amdgcn_logf(1.f + e), fmaxf(z, 0.f));
}

#ifndef X3_PTS
#define X3_PTS 64
#endif
// Cycle stamps (-DX3_PROFILE, tools/dev/x3_profile.py) of one 256-wide softplus layer, 8 waves: GEMM 14.5 k cycles for wave 0 + 3.9 k waiting
// at the barrier for its SIMD partner (12.3 k of MFMA issue for the pair: the GEMM phase is ~66 % efficient), epilogue 4.7 k + 2.4 k
// waiting for the partner's: 25.6 k per layer, 48 % of it MFMA.  16 waves: the GEMM phase grows to 22.5 k (twice the weight-fragment
// requests: L1 delivers 64 B/clk hit or miss) -> 2.15 ms instead of 1.89 ms per training step; 4 waves: 26.6 k (one wave per SIMD cannot
// cover its own operand latencies) -> 2.43 ms.  Requesting the bias values before the GEMM and a rolled (truly prefetching) weight
// pipeline change nothing (the partner wave already hides those latencies); softplus' exp / log are 0.2 of the 1.9 ms.
#ifndef X3_WAVES
#define X3_WAVES 8
#endif
template <int PTS>
struct X3Cfg {
    static constexpr int WAVES = PTS == 64 ? X3_WAVES : 4;      // PTS = 64: 8 (or 16);  PTS = 32: 4
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

}  // namespace es
