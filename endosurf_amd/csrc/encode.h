// Frequency encodings written straight into the k-major LDS tiles (reference src/renderer/encoder.py:40-54:
// [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], each block ``D`` wide).
#pragma once
#include "chain_common.h"

namespace es {

// 3-D input held as px[c*64 + row]; writes 3*(1+2L) rows of the tile starting at k = kbase. One value row per tile row.
template <int L>
__device__ __forceinline__ void encode3(float* At, int kbase, const float* px, int tid) {
    const int row = tid & 63, part = tid >> 6;
    for (int item = part; item < 3 * L; item += 4) {
        const int c = item % 3, i = item / 3;
        float s, co;
        sincosf(px[c * 64 + row] * (float)(1 << i), &s, &co);
        At[swz(kbase + enc_index(3, i, 0, c), row)] = s;
        At[swz(kbase + enc_index(3, i, 1, c), row)] = co;
    }
    if (part == 3) {
#pragma unroll
        for (int c = 0; c < 3; ++c) At[swz(kbase + c, row)] = px[c * 64 + row];
    }
}

// 1-D input pt[row]; writes (1+2L) rows starting at kbase
template <int L>
__device__ __forceinline__ void encode1(float* At, int kbase, const float* pt, int tid) {
    const int row = tid & 63, part = tid >> 6;
    for (int i = part; i < L; i += 4) {
        float s, co;
        sincosf(pt[row] * (float)(1 << i), &s, &co);
        At[swz(kbase + enc_index(1, i, 0, 0), row)] = s;
        At[swz(kbase + enc_index(1, i, 1, 0), row)] = co;
    }
    if (part == 2) At[swz(kbase, row)] = pt[row];
}

__device__ __forceinline__ void zero_rows(float* At, int k0, int k1, int tid) {
    for (int i = tid; i < (k1 - k0) * 64; i += NTHREADS) At[swz(k0 + (i >> 6), i & 63)] = 0.f;
}

}  // namespace es
