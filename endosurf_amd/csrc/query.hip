// K2: fused no-grad SDF query  sdf(x + deform(x, t))  — the reference's
// EndoSurfNet.get_sdf_from_observed_space (endosurf.py:570-579: DeformNetwork.forward :724-738 then
// SDFNetwork.sdf :788-791), used by hierarchical up-sampling (:92, :281), ray marching (:375), the secant
// refinement (:435) and mesh extraction (:493).  One launch runs both 9-layer MLPs per 64-point tile with the
// activations resident in LDS; only 16 B/point are read and 4 B/point written.
#include <cstdlib>

#include "chain_common.h"
#include "encode.h"
#include "launch.h"
#include "tabs.h"
#include "timing.h"

namespace es {

#ifdef ES_PROFILE_QUERY      // dev builds only: cycle stamps of block 0 / thread 0 at the phase boundaries (tools/dev/q_profile.py)
__device__ long long q_prof[192];      // [0, 64): block 0 (first round) | [64, 128): the last block (last round) | [128, 192): their wall clocks (100 MHz)
#define Q_STAMP(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) { \
        const int b_ = blockIdx.x == 0 ? 0 : 64; q_prof[b_ + (i)] = __builtin_readcyclecounter(); \
        if ((i) == 0 || (i) == 7) q_prof[128 + (b_ >> 1) + (i)] = wall_clock64(); } } while (0)
extern "C" int es_debug_q_profile(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(q_prof), sizeof(long long) * (n < 192 ? n : 192)); }
__device__ long long q_times[2 * 4096];      // wall clock (100 MHz) at the start and the end of every block of the last launch (<= 4096 blocks)
#define Q_BLOCK_TIME(e) do { if (threadIdx.x == 0 && blockIdx.x < 4096) q_times[2 * blockIdx.x + (e)] = wall_clock64(); } while (0)
extern "C" int es_debug_q_times(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(q_times), sizeof(long long) * (n < 8192 ? n : 8192)); }
#else
#define Q_STAMP(i) do {} while (0)
#define Q_BLOCK_TIME(e) do {} while (0)
#endif

// HALF: latency-bound small batches (secant iterations, 8-sample up-sampling queries) use 32-point tiles: the LDS tile keeps
// its 64-row layout but only row-tile 0 carries points, so every layer issues half the MFMAs and twice as many workgroups
// share the batch.
template <bool DEFORM, bool HALF>
__global__ __launch_bounds__(NTHREADS, 2) void k_query_sdf(PointSrc src, Tabs tb, const float4* __restrict__ packed,
                                                        const float* __restrict__ weff, float* __restrict__ sdf_out, int ld_out,
                                                        const int* __restrict__ ray_done) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    float* aux = lds + MAIN_FLOATS;
    float* scr = aux + AUX56_FLOATS;
    float* px = scr;          // [3][64]
    float* pt = scr + 192;    // [64]
    float* red = aux;         // [4][<=3][64]: aliases the encoding rows, dead by the time the tiny last layers run
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int RTC = HALF ? 1 : 2;
    constexpr int PTS = HALF ? 32 : 64;
    const int row0 = blockIdx.x * PTS;
    if (ray_done != nullptr) {      // block-wise ray marching: a tile whose rays already have their first sign change is skipped
        const int r_first = row0 / src.n_per_ray, r_last = min(row0 + PTS - 1, src.M - 1) / src.n_per_ray;
        bool all_done = true;
        for (int r = r_first; r <= r_last; ++r) all_done = all_done && ray_done[r] != 0;
        if (all_done) return;       // workgroup-uniform
    }

    const QuadOff<RTC> qo = quad_offsets<RTC>(0, 2 * wave, lane);
    auto bias2 = [&](float(&b)[2], const float* __restrict__ bias) { b[0] = bias[64 * wave + (lane & 31)]; b[1] = bias[64 * wave + 32 + (lane & 31)]; };
    Q_STAMP(0);
    Q_BLOCK_TIME(0);
    if (tid < 64) {
        float x[3], t, d[3];
        load_point(src, tid < PTS ? row0 + tid : src.M, x, t, d);
        px[tid] = x[0]; px[64 + tid] = x[1]; px[128 + tid] = x[2]; pt[tid] = t;
    }
    __syncthreads();

    if (DEFORM) {
        // ---- deformation MLP, value only: x_c = x + MLP([enc6(x), enc6(t)]) ----
        encode3<6>(aux, 0, px, tid);
        encode1<6>(aux, 39, pt, tid);
        zero_rows(aux, 52, 56, tid);
        __syncthreads();
        Q_STAMP(1);
        float bc[2], bn[2];   // this layer's and the NEXT layer's bias of this lane's two columns (requested a layer ahead)
        bias2(bn, weff + tb.boff[NET_D * LAYERS + 0]);
        {
            f32x16 acc[RTC][2];
            acc_zero(acc);
            bc[0] = bn[0]; bc[1] = bn[1];
            bias2(bn, weff + tb.boff[NET_D * LAYERS + 1]);
            gemm_seg<7, RTC, 2>(acc, aux, packed + tb.segoff[DF0], 0, 2 * wave, lane);
            for_quads_off(acc, qo, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int off, int ni) {
                add_bias4(v, bc[ni]);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = relu1(v[i]);
                lds_store_quad_at(mainT, off, v);
            });
        }
        __syncthreads();
        Q_STAMP(2);
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            f32x16 acc[RTC][2];
            acc_zero(acc);
            bc[0] = bn[0]; bc[1] = bn[1];
            if (l < 7) bias2(bn, weff + tb.boff[NET_D * LAYERS + l + 1]);
            Q_STAMP(10 + l);
            gemm_seg<32, RTC, 2>(acc, mainT, packed + tb.segoff[DF0 + l], 0, 2 * wave, lane);
            Q_STAMP(20 + l);
            __syncthreads();
            if (l == 3 && wave == 3) {      // (wave-uniform: the skip's address arithmetic stays out of every other epilogue)
                for_quads_off(acc, qo, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int off, int ni) {
                    if (col >= 204) {
                        lds_load_quad(aux, col - 204, row, v);   // IDR skip: next input = [h(204) | enc(52)] (1/sqrt2 folded into W4)
                    } else {
                        add_bias4(v, bc[ni]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = relu1(v[i]);
                    }
                    lds_store_quad_at(mainT, off, v);
                });
            } else {
                for_quads_off(acc, qo, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int off, int ni) {
                    add_bias4(v, bc[ni]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = relu1(v[i]);
                    lds_store_quad_at(mainT, off, v);
                });
            }
            __syncthreads();
        }
        Q_STAMP(3);
        smalln_partial<3>(mainT, weff + tb.woff[NET_D * LAYERS + 8], 256, red, tid);
        __syncthreads();
        if (tid < 192) {
            const int i = tid >> 6, row = tid & 63;
            px[i * 64 + row] += smalln_reduce<3>(red, i, row) + weff[tb.boff[NET_D * LAYERS + 8] + i];
        }
        __syncthreads();
    }

    // ---- SDF MLP on x_c, output column 0 only ----
    Q_STAMP(4);
    encode3<6>(aux, 0, px, tid);
    zero_rows(aux, 39, 40, tid);
    __syncthreads();
    float bc[2], bn[2];
    bias2(bn, weff + tb.boff[NET_S * LAYERS + 0]);
    {
        f32x16 acc[RTC][2];
        acc_zero(acc);
        bc[0] = bn[0]; bc[1] = bn[1];
        bias2(bn, weff + tb.boff[NET_S * LAYERS + 1]);
        gemm_seg<5, RTC, 2>(acc, aux, packed + tb.segoff[SF0], 0, 2 * wave, lane);
        for_quads_off(acc, qo, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int off, int ni) {
            add_bias4(v, bc[ni]);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = softplus100(v[i]);
            lds_store_quad_at(mainT, off, v);
        });
    }
    __syncthreads();
    Q_STAMP(5);
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        f32x16 acc[RTC][2];
        acc_zero(acc);
        bc[0] = bn[0]; bc[1] = bn[1];
        if (l < 7) bias2(bn, weff + tb.boff[NET_S * LAYERS + l + 1]);
        Q_STAMP(30 + l);
        const int seg = l <= 4 ? SF0 + l : SF0 + l + 1;
        gemm_seg<32, RTC, 2>(acc, mainT, packed + tb.segoff[seg], 0, 2 * wave, lane);
        if (l == 4) gemm_seg<5, RTC, 2>(acc, aux, packed + tb.segoff[SF4A], 0, 2 * wave, lane);   // NeRF skip: + enc part
        __syncthreads();
        for_quads_off(acc, qo, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int off, int ni) {
            add_bias4(v, bc[ni]);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = softplus100(v[i]);
            lds_store_quad_at(mainT, off, v);
        });
        __syncthreads();
    }
    Q_STAMP(6);
    smalln_partial<1>(mainT, weff + tb.woff[NET_S * LAYERS + 8], 256, red, tid);
    __syncthreads();
    Q_STAMP(7);
    Q_BLOCK_TIME(1);
    if (tid < PTS && row0 + tid < src.M) {
        const int i = row0 + tid;
        const size_t o = ld_out > 0 ? (size_t)(i / src.n_per_ray) * ld_out + (i % src.n_per_ray) : (size_t)i;   // [ray][ld_out] or flat
        sdf_out[o] = smalln_reduce<1>(red, 0, tid) + weff[tb.boff[NET_S * LAYERS + 8]];
    }
}

int query_sdf16(const PointSrc& src, const float* packed, const float* weff, float* sdf_out, int use_deform, hipStream_t st);

// tile_points: 0 = chosen by the batch size (<= 9 216 points: the 16-point tiles of query16.hip -- the up-sampling queries of 1 024 rays and
// the secant iterations are latency-bound on their own; <= 16 384: 32-point tiles, fewer than one 64-point tile per CU otherwise; above: 64),
// or 16 / 32 / 64 as the caller says.  A caller whose query SHARES the GPU with another chain of small launches asks for shorter tiles than
// the batch size alone suggests: the 32 768-point coarse query of a training step takes 0.48 ms on 512 64-point tiles when it runs alone,
// but next to the secant iterations those hold every workgroup slot for half a millisecond, while 1 024 32-point tiles turn the slots over
// twice as often (racing chains 1.42 -> 1.35 ms; 16-point tiles: 1.37 ms).
int query_sdf(const PointSrc& src, const float* packed, const float* weff, float* sdf_out, int use_deform, hipStream_t st,
              int ld_out, const int* ray_done, int tile_points) {
    int tp = tile_points ? tile_points : (src.M <= 9216 ? 16 : (src.M <= 16384 ? 32 : 64));
    if (tp == 16 && !(ld_out == 0 && ray_done == nullptr)) tp = 32;      // (the 16-point kernel writes flat outputs only)
    if (src.M > 0 && tp == 16)
        return query_sdf16(src, packed, weff, sdf_out, use_deform, st);   // latency-bound batches
    static DeviceOnce attr_done;
    if (attr_done.first()) {
        if (int e = allow_big_lds(k_query_sdf<true, false>, LEAN_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_query_sdf<false, false>, LEAN_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_query_sdf<true, true>, LEAN_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_query_sdf<false, true>, LEAN_LDS_BYTES)) return e;
        attr_done.done();
    }
    if (src.M <= 0) return ST_OK;
    const Tabs tb = make_tabs();
    const bool half = tp == 32;
    const int pts = half ? 32 : 64;
    const dim3 grid((src.M + pts - 1) / pts), block(NTHREADS);
    const float4* pk = reinterpret_cast<const float4*>(packed);
    ScopedTimer tm(ray_done ? KID_QUERY_EXIT : KID_QUERY, src.M, st);     // launches with early-exit tiles are not counted as work
    if (use_deform) {
        if (half) hipLaunchKernelGGL((k_query_sdf<true, true>), grid, block, LEAN_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done);
        else hipLaunchKernelGGL((k_query_sdf<true, false>), grid, block, LEAN_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done);
    } else {
        if (half) hipLaunchKernelGGL((k_query_sdf<false, true>), grid, block, LEAN_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done);
        else hipLaunchKernelGGL((k_query_sdf<false, false>), grid, block, LEAN_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done);
    }
    return hip_last("query_sdf");
}

}  // namespace es
