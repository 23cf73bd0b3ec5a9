// Split-precision SDF query, REGISTER-RESIDENT formulation (opt-in like query_x3.hip, same arithmetic: three exact bf16 planes per fp32
// operand, six partial products on v_mfma_f32_32x32x16_bf16, fp32 accumulation; reference EndoSurfNet.get_sdf_from_observed_space,
// endosurf.py:570-579).
//
// query_x3.hip keeps the activations of a 64-point tile in LDS and lets every wave read all of them: at the bf16 matrix rate that kernel
// is bound by operand delivery -- 128 B/clk of LDS for the activation planes plus 64 B/clk of L1 for the weight fragments is exactly what
// its 8 waves ask for, and the GEMM phase reaches 2/3 of the MFMA issue rate, the layer 48 %.  Here the roles are swapped:
//   * a wave owns 32 POINTS and ALL 256 features of them.  The accumulators of layer l (8 blocks of 32 features x 32 points, 128
//     registers) are, after bias + activation + splitting, the B operands of layer l+1 -- lane (point n, half hi) holds exactly the 8
//     consecutive-in-our-order k values of its half of a 16-wide k-step.  Activations never leave the register file;
//   * the k order inside a k-step is a free choice as long as both operands agree, so the weights are packed in the order the
//     accumulator layout produces:  k-step s = (block b = s / 2, p = s % 2), lane half hi, element j  <->  feature
//     32 b + 16 p + (j < 4 ? 4 hi + j : 8 + 4 hi + j - 4);
//   * the weights (393 KB per 256 x 256 layer in split form) are the only LDS traffic: 24 KB per k-step (8 feature blocks x 3 planes x
//     64 lanes x 16 B), streamed global -> LDS by direct loads (global_load_lds_dwordx4, no staging registers) into a ring of 4
//     k-steps, ONE workgroup barrier per TWO k-steps, shared by the 4 waves (one per SIMD) of the workgroup: 64 B/clk of LDS reads,
//     16 B/clk of L2 reads per CU.  (A direct load costs ~45 issue cycles per 1 KB piece on a wave that has no partner to hide them;
//     the register-staged alternative -- 6 global loads + 6 ds_write_b128 per wave and k-step -- was measured at twice that.)
//   * the epilogue of layer l is spread over the k-steps of layer l+1: while the 48 MFMAs of k-step s run, the VALU builds the operand
//     of k-step s+1 from 8 accumulator values per lane.
// One workgroup = 256 threads = 4 waves = 128 points.  Both accumulator sets (previous layer / this layer) live in registers: ~400 of
// the 512 per lane that a one-wave-per-SIMD kernel may use.
#include "chain_common.h"
#include "launch.h"
#include "x3r_core.h"
#include "tabs.h"
#include "timing.h"

namespace es {

#ifdef XR_PROFILE
__device__ long long xr_prof[512];
#endif
constexpr int XR_PTS = 128;
constexpr int XR_LDS_BYTES = XR_RING * XR_CHUNK_BYTES + (XR_PTS * XR_ENC_LD + 16 * 256 + 4 * 256 + 4) * 4;
static_assert(XR_LDS_BYTES <= 160 * 1024, "LDS carve");

// ---- weight packing: chunk c = (segment, k-step g): [8 feature blocks][3 planes][64 lanes] x 16 B ------------------------------
struct XrPackArgs { int row0[XR_COUNT], col0[XR_COUNT], kreal[XR_COUNT], nreal[XR_COUNT], skip[XR_COUNT], K[XR_COUNT], woff[XR_COUNT], net[XR_COUNT],
                    dir[XR_COUNT], chunk0[XR_COUNT + 1]; };
__global__ __launch_bounds__(256) void k_pack_x3r(const float* __restrict__ weff, u32x4* __restrict__ out, XrPackArgs a, int first_net) {
    const unsigned idx = blockIdx.x * 256 + threadIdx.x;            // (chunk, feature block, lane)
    if (idx >= (unsigned)(XR_CHUNKS + XR_PAD_CHUNKS) * 8 * 64) return;
    const int lane = idx & 63, fb = (idx >> 6) & 7, c = idx >> 9;
    u32x4* o = out + ((size_t)c * XR_CHUNK_UNITS + fb * 3) * 64 + lane;
    if (c >= XR_CHUNKS) { o[0] = o[64] = o[128] = u32x4{0u, 0u, 0u, 0u}; return; }
    int si = 0;
#pragma unroll 1
    for (int i = 1; i < XR_COUNT; ++i)
        if (c >= a.chunk0[i]) si = i;
    if (a.net[si] < first_net) return;
    const int g = c - a.chunk0[si];
    const float* W = weff + a.woff[si];
    const float sc = a.skip[si] ? INV_SQRT2 : 1.f;
    const int n = 32 * fb + (lane & 31);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * g + xr_kperm(lane >> 5, j);
        // dir 0: B[k][n] = W[row0 + n][col0 + k] (y = x W^T);  dir 1: B[k][n] = W[row0 + k][col0 + n] (input adjoint = output adjoint x W)
        const size_t wi = a.dir[si] == 0 ? (size_t)(a.row0[si] + n) * a.K[si] + a.col0[si] + k : (size_t)(a.row0[si] + k) * a.K[si] + a.col0[si] + n;
        v[j] = (k < a.kreal[si] && n < a.nreal[si]) ? sc * W[wi] : 0.f;
    }
    u32x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned hh, mm, ll;
        split_pair(v[2 * j], v[2 * j + 1], hh, mm, ll);
        h[j] = hh; m[j] = mm; l[j] = ll;
    }
    o[0] = h; o[64] = m; o[128] = l;
}

template <bool DEFORM>
__global__ __launch_bounds__(XR_THREADS, 1) void k_query_sdf_x3r(PointSrc src, Tabs tb, const u32x4* __restrict__ chunks,
                                                               const float* __restrict__ weff, float* __restrict__ sdf_out, int ld_out,
                                                               const int* __restrict__ ray_done) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsr[];
    float* encs = reinterpret_cast<float*>(ldsr + XR_RING * XR_CHUNK_BYTES);      // [128 points][68]
    float* biasL = encs + XR_PTS * XR_ENC_LD;                               // [16 layers][256]: deform 0..7, sdf 0..7
    float* w8L = biasL + 16 * 256;                                          // [4][256]: deform last-layer rows 0..2, sdf last-layer row 0
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const int row0 = blockIdx.x * XR_PTS;
    if (ray_done != nullptr) {      // block-wise ray marching: a tile whose rays already have their first sign change is skipped
        const int r_first = row0 / src.n_per_ray, r_last = min(row0 + XR_PTS - 1, src.M - 1) / src.n_per_ray;
        bool all_done = true;
        for (int r = r_first; r <= r_last; ++r) all_done = all_done && ray_done[r] != 0;
        if (all_done) return;       // workgroup-uniform
    }
    XR_STAMP(0);
    const int prow = wave * 32 + n;              // this lane's point inside the tile (both lane halves hold the same point)
    float* erow = encs + prow * XR_ENC_LD;
    float x[3], t, dd[3];
    load_point(src, row0 + prow, x, t, dd);

    // biases and last-layer rows into LDS
    for (int i = tid; i < 16 * 256; i += XR_THREADS) {
        const int l = i >> 8, f = i & 255, net = l < 8 ? NET_D : NET_S, ll = l & 7;
        const int nout = (net == NET_D && ll == 3) ? 204 : 256;
        biasL[i] = f < nout ? weff[tb.boff[net * LAYERS + ll] + f] : 0.f;
    }
    for (int i = tid; i < 4 * 256; i += XR_THREADS)
        w8L[i] = i < 768 ? weff[tb.woff[NET_D * LAYERS + 8] + i] : weff[tb.woff[NET_S * LAYERS + 8] + (i - 768)];
    if (tid < 4) w8L[4 * 256 + tid] = tid < 3 ? weff[tb.boff[NET_D * LAYERS + 8] + tid] : weff[tb.boff[NET_S * LAYERS + 8]];   // last-layer biases
    const float* b8 = w8L + 4 * 256;

    // encoding of this lane's point: the two lane halves share the frequencies
    auto encode_x = [&](const float (&p)[3]) {
#pragma unroll
        for (int ii = 0; ii < 3; ++ii) {
            const int i = 3 * hi + ii;
            const float f = (float)(1 << i);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s, co;
                sincosf(p[c] * f, &s, &co);
                erow[enc_index(3, i, 0, c)] = s;
                erow[enc_index(3, i, 1, c)] = co;
            }
        }
        if (hi == 0) { erow[0] = p[0]; erow[1] = p[1]; erow[2] = p[2]; }
    };
    WStream ws;
    ws.g = chunks; ws.ring = ldsr; ws.k = DEFORM ? 0 : XR_SDF_CHUNK0; ws.wave = wave; ws.lane = lane;

    const auto enc_val = [&](int s, int j) -> float { return erow[16 * s + xr_kperm(hi, j)]; };
    f32x16 P[8], C[8];
    if (DEFORM) {
        encode_x(x);
#pragma unroll
        for (int ii = 0; ii < 3; ++ii) {
            const int i = 3 * hi + ii;
            float s, co;
            sincosf(t * (float)(1 << i), &s, &co);
            erow[39 + enc_index(1, i, 0, 0)] = s;
            erow[39 + enc_index(1, i, 1, 0)] = co;
        }
        if (hi == 0) erow[39] = t;
        else {
#pragma unroll
            for (int k = 52; k < 64; ++k) erow[k] = 0.f;
        }
        __syncthreads();                                   // biasL / w8L visible (the encoding rows are private to the wave)
        ws.start();
        init8(C, biasL, hi);
        gemm_r<4>(C, ws, enc_val);
        copy8(P, C);
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            const bool skip = l == 4;                      // IDR skip: input of layer 4 = [h(204) | enc(52)] (1/sqrt2 folded into W4)
            init8(C, biasL + l * 256, hi);
            gemm_r<16>(C, ws, [&](int s, int j) -> float {
                const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
                const int f = 32 * b + 8 * q + 4 * hi + i;
                const float h = fmaxf(P[b][4 * q + i], 0.f);
                if (32 * b + 8 * q + 4 + i < 204) return h;           // compile-time: this register is a hidden feature for both halves
                return (skip && f >= 204) ? erow[f - 204] : h;
            });
            copy8(P, C);
        }
        {   // x_c = x + W8 relu(z_7) + b8
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 32 * b + 8 * (r >> 2) + 4 * hi + (r & 3);
                    const float h = fmaxf(P[b][r], 0.f);
                    d0 = fmaf(w8L[f], h, d0); d1 = fmaf(w8L[256 + f], h, d1); d2 = fmaf(w8L[512 + f], h, d2);
                }
            d0 += __shfl_xor(d0, 32); d1 += __shfl_xor(d1, 32); d2 += __shfl_xor(d2, 32);
            x[0] += d0 + b8[0]; x[1] += d1 + b8[1]; x[2] += d2 + b8[2];
        }
    }
    // ---- SDF MLP on x_c, output column 0 only ----
    encode_x(x);
    if (hi == 1) {
#pragma unroll
        for (int k = 39; k < 64; ++k) erow[k] = 0.f;
    }
    if (!DEFORM) {
        __syncthreads();
        ws.start();
    }
    XR_STAMP(1);
    init8(C, biasL + 8 * 256, hi);
    gemm_r<4>(C, ws, enc_val);
    copy8(P, C);
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        XR_STAMP(10 + l);
        init8(C, biasL + (8 + l) * 256, hi);
        gemm_r<16>(C, ws, [&](int s, int j) -> float {
            const int b = s >> 1, q = 2 * (s & 1) + (j >> 2), i = j & 3;
            return softplus100_native(P[b][4 * q + i]);
        });
        if (l == 4) gemm_r<4>(C, ws, enc_val);             // NeRF skip: + encoding part (chunks of SF4A follow those of SF4M)
        copy8(P, C);
    }
    XR_STAMP(18);
    {
        float s0 = 0.f;
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = 32 * b + 8 * (r >> 2) + 4 * hi + (r & 3);
                s0 = fmaf(w8L[768 + f], softplus100_native(P[b][r]), s0);
            }
        s0 += __shfl_xor(s0, 32);
        const int i = row0 + prow;
        if (hi == 0 && i < src.M) {
            const size_t o = ld_out > 0 ? (size_t)(i / src.n_per_ray) * ld_out + (i % src.n_per_ray) : (size_t)i;   // [ray][ld_out] or flat
            sdf_out[o] = s0 + b8[3];
        }
    }
    XR_STAMP(19);
}

#ifdef XR_PROFILE
extern "C" int es_debug_xr_profile(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(xr_prof), sizeof(long long) * (n < 512 ? n : 512));
}
extern "C" int es_debug_xr_reset() {
    static long long z[512];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(xr_prof), z, sizeof(z));
}
#endif

size_t packed_x3r_bytes() { return (size_t)(XR_CHUNKS + XR_PAD_CHUNKS) * XR_CHUNK_BYTES; }

int pack_x3r(const float* weff, void* packed, int use_deform, hipStream_t st) {
    XrPackArgs a;
    const Tabs tb = make_tabs();
    for (int i = 0; i < XR_COUNT; ++i) {
        const SegDesc& s = SEGS[XR_SEGS[i]];
        a.row0[i] = s.row0; a.col0[i] = s.col0; a.kreal[i] = s.kreal; a.nreal[i] = s.nreal; a.skip[i] = s.skip_scale; a.net[i] = s.net; a.dir[i] = s.dir;
        a.K[i] = LAYER_K[s.net][s.layer]; a.woff[i] = tb.woff[s.net * LAYERS + s.layer]; a.chunk0[i] = xr_chunk0(i);
    }
    a.chunk0[XR_COUNT] = XR_CHUNKS;
    const unsigned nthr = (unsigned)(XR_CHUNKS + XR_PAD_CHUNKS) * 8 * 64;
    hipLaunchKernelGGL(k_pack_x3r, dim3((nthr + 255) / 256), dim3(256), 0, st, weff, reinterpret_cast<u32x4*>(packed), a, use_deform ? 0 : 1);
    return hip_last("pack_x3r");
}

int query_sdf_x3r(const PointSrc& src, const void* packed, const float* weff, float* sdf_out, int use_deform, hipStream_t st, int ld_out,
                  const int* ray_done) {
    static DeviceOnce attr_done;
    if (attr_done.first()) {
        if (int e = allow_big_lds(k_query_sdf_x3r<true>, XR_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_query_sdf_x3r<false>, XR_LDS_BYTES)) return e;
        attr_done.done();
    }
    if (src.M <= 0) return ST_OK;
    const Tabs tb = make_tabs();
    const dim3 grid((src.M + XR_PTS - 1) / XR_PTS), block(XR_THREADS);
    const u32x4* pk = reinterpret_cast<const u32x4*>(packed);
    ScopedTimer tm(ray_done ? KID_QUERY_EXIT : KID_QUERY_X3, src.M, st);
    if (use_deform) hipLaunchKernelGGL((k_query_sdf_x3r<true>), grid, block, XR_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done);
    else hipLaunchKernelGGL((k_query_sdf_x3r<false>), grid, block, XR_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done);
    return hip_last("query_sdf_x3r");
}

}  // namespace es
