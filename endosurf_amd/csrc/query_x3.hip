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
#include "x3_common.h"
#include "tabs.h"
#include "timing.h"

namespace es {

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
template <bool DEFORM, int PTS>
__global__ __launch_bounds__(X3Cfg<PTS>::THREADS, (PTS == 32 ? 2 : 1)) void k_query_sdf_x3(PointSrc src, Tabs tb, X3Tabs xt, const u32x4* __restrict__ packed,
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
// the register-resident formulation (query_x3r.hip) keeps its own chunk-ordered copy of the split weights behind this kernel's
size_t packed_x3r_bytes();
int pack_x3r(const float* weff, void* packed, int use_deform, hipStream_t st);
int query_sdf_x3r(const PointSrc& src, const void* packed, const float* weff, float* sdf_out, int use_deform, hipStream_t st, int ld_out,
                  const int* ray_done);
constexpr int X3_SMALL_MAX = 8192;      // batches up to here run the 32-point LDS-resident tiles
static bool use_x3r() { return true; }      // (large batches: the register-resident kernel of query_x3r.hip; round 3's A/B switch is gone)
size_t packed_x3_bytes() { return X3_UNITS * 16 + packed_x3r_bytes(); }
const void* packed_x3r_part(const void* packed_x3) { return static_cast<const unsigned char*>(packed_x3) + X3_UNITS * 16; }

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
    // the LDS-resident kernel's fragment order: small batches (32-point tiles, below) and the A/B runs of dev builds (ES_X3R=0)
    hipLaunchKernelGGL(k_pack_x3, dim3((n + 255) / 256), dim3(256), 0, st, weff, reinterpret_cast<u32x4*>(packed_x3), a, use_deform ? 0 : 1);
    if (int e = hip_last("pack_x3")) return e;
    return pack_x3r(weff, static_cast<unsigned char*>(packed_x3) + X3_UNITS * 16, use_deform, st);
}

int query_sdf_x3(const PointSrc& src, const void* packed_x3, const float* weff, float* sdf_out, int use_deform, hipStream_t st, int ld_out,
                 const int* ray_done) {
    static DeviceOnce attr_done;
    if (attr_done.first()) {
        if (int e = allow_big_lds(k_query_sdf_x3<true, 64>, X3Cfg<64>::LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_query_sdf_x3<false, 64>, X3Cfg<64>::LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_query_sdf_x3<true, 32>, X3Cfg<32>::LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_query_sdf_x3<false, 32>, X3Cfg<32>::LDS_BYTES)) return e;
        attr_done.done();
    }
    if (src.M <= 0) return ST_OK;
    // Small batches (the 8 192-point up-sampling queries of a training step): the register-resident kernel's 128-point blocks would
    // leave most of the chip idle (64 blocks) for the latency of a whole 17-layer chain; the LDS-resident formulation with 32-point tiles
    // (4 waves, 62 KB of LDS: two workgroups per CU) spreads them over 256 workgroups.  Same arithmetic, same tests.
    if (src.M <= X3_SMALL_MAX && ld_out == 0 && ray_done == nullptr) {
        using Cfg = X3Cfg<32>;
        const Tabs tb = make_tabs();
        const X3Tabs xt = make_x3_tabs();
        const dim3 grid((src.M + 31) / 32), block(Cfg::THREADS);
        const u32x4* pk = reinterpret_cast<const u32x4*>(packed_x3);
        ScopedTimer tm(KID_QUERY_X3, src.M, st);
        if (use_deform) hipLaunchKernelGGL((k_query_sdf_x3<true, 32>), grid, block, Cfg::LDS_BYTES, st, src, tb, xt, pk, weff, sdf_out, ld_out, ray_done);
        else hipLaunchKernelGGL((k_query_sdf_x3<false, 32>), grid, block, Cfg::LDS_BYTES, st, src, tb, xt, pk, weff, sdf_out, ld_out, ray_done);
        return hip_last("query_sdf_x3[32]");
    }
    if (use_x3r()) return query_sdf_x3r(src, static_cast<const unsigned char*>(packed_x3) + X3_UNITS * 16, weff, sdf_out, use_deform, st, ld_out, ray_done);
    constexpr int PTS = 64;
    using Cfg = X3Cfg<PTS>;
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
