// K2 in the TRANSPOSED formulation (round 5): the fused no-grad SDF query  sdf(x + deform(x, t))  of query.hip -- the reference's
// EndoSurfNet.get_sdf_from_observed_space (endosurf.py:570-579: DeformNetwork.forward :724-738, SDFNetwork.sdf :788-791) -- computed as
//   Y^T = W X^T :  MFMA A operand = weights (32 output features x 2 k), B operand = activations (2 k x 32 points),
//                  D = for every lane ONE point (column) and, per register quad, FOUR features (rows).
// The activation tile is therefore ROW-MAJOR in LDS, [point][256 features + 4 pad] floats, and both LDS streams of a layer move 16 bytes
// per lane and instruction: the operand read (one ds_read_b128 per point tile and k-GROUP of 8, where the k-major tile of query.hip
// needs one ds_read_b32 per k-STEP of 2) and the epilogue write (one ds_write_b128 per quad, as before):
//   64 + 16 (+ 8 bias reads) LDS instructions per 256 x 256 layer and wave instead of 256 + 16.
// The same packed weights serve (arch.h: element j of pack slot (lo, hi) = W[32 nt + lo][8 g + 2 j + hi] is an A fragment as well as a
// B fragment); two index permutations make the 16-byte accesses line up with them, and keep every sum in the order query.hip computes it
// (the no-grad queries feed DISCRETE decisions -- ray marching's first sign change -- so the results must not move by a rounding):
//   * within each group of 8 features the tile stores feature f at position pos(f) = 4 (f & 1) + ((f >> 1) & 3): the four floats a lane
//     (., hi) reads at [point][8 g + 4 hi ..] are features 8 g + 2 j + hi, j = 0..3 = the k a lane of that half supplies in k-steps
//     j = 0..3 of the packed float4, i.e. the k pairs (0,1), (2,3), ... are accumulated in the order of the k-major kernel;
//   * a lane loads its weights from pack slot (phi(lo), hi), phi(8 q + 4 h + i) = 8 q + 2 i + h: MFMA row m = 8 q + 4 h + i of an
//     output tile then IS feature phi(m), and the quad (q, hi) a lane holds after the MFMA (rows 8 q + 4 hi + i) is features
//     8 q + 2 i + hi, i = 0..3 = positions 8 q + 4 hi + 0..3 of the next layer's tile: one aligned float4.
// Bias: a lane needs four different values per quad (they vary along the register, not along the lane): each wave keeps the permuted
// bias of its own 64 features of the current layer in the 4-float padding of the encoding tile's rows (256 floats = one layer), reads
// it back as 8 broadcast ds_read_b128 and stages the next layer's values behind its own epilogue (no barrier: a wave only ever touches
// its own 64 entries).  LDS: 64 x 260 + 64 x 60 floats = exactly 80 KiB, two workgroups per CU as before.
// STATUS (round 5): measured and NOT kept.  Bit-identical to query.hip on every shape tried, and the same speed to +-0.3 % (1.893 vs 1.891 ms
// on the 131 072-point launch): the LDS instruction count is not what bounds these kernels (DEAD_ENDS.md, profiles/r05_ab_query_transposed.txt).
// Round 6: moved out of endosurf_amd/csrc (the product sources hold only what ships).  To measure it again: copy this file back into
// endosurf_amd/csrc, declare query_sdf_t in query.hip and route query_sdf() to it (the round-5 hook read ES_QT: 1 = both tile heights,
// 2 = 64-point tiles only, >= 100: the timing-experiment instantiations tools/dev/qt_ab.py drives), build with -DES_DEV_SWITCHES.
#ifdef ES_DEV_SWITCHES
#include <cstdlib>

#include "chain_common.h"
#include "launch.h"
#include "tabs.h"
#include "timing.h"

namespace es {

constexpr int LDM = HID + 4;                 // floats per point row of the main tile (the pad holds x, y, z, t of the point)
constexpr int LDA = 56 + 4;                  // floats per point row of the encoding tile (the pad holds 4 permuted bias values)
constexpr int QT_MAIN_FLOATS = TM * LDM;
constexpr int QT_AUX_FLOATS = TM * LDA;
constexpr int QT_LDS_BYTES = (QT_MAIN_FLOATS + QT_AUX_FLOATS) * 4;      // 81 920 B
static_assert(QT_LDS_BYTES == 80 * 1024, "two workgroups per CU");

// position of feature k inside a tile row / feature stored at position p (inverse)
__device__ __forceinline__ constexpr int ppos(int k) { return (k & ~7) | ((k & 1) << 2) | ((k >> 1) & 3); }
__device__ __forceinline__ constexpr int pfeat(int p) { return (p & ~7) | ((p & 3) << 1) | ((p >> 2) & 1); }

// acc[mi][pi] += W[features of m-tile mt0+mi][0 .. 8 KG) * X^T[0 .. 8 KG)[points of point tile pi]
// Xt: row-major tile with LD floats per point; W: packed segment ([nt][g][slot] float4, KG groups).
template <int KG, int PTC, int LD, int DBG = 0, int PF = 2>
__device__ __forceinline__ void gemm_seg_t(f32x16 (&acc)[2][PTC], const float* Xt, const float4* __restrict__ W, int mt0, int lane, int dbg = 0) {
    static_assert(PF == 2 || PF == 4, "prefetch depth");
    constexpr bool GUARD = (KG % (2 * PF)) != 0;
    const int lo = lane & 31, hi = lane >> 5;
    const float4* wl = W + (dbg ? lane : (((lo & 24) | ((lo & 3) << 1) | ((lo >> 2) & 1)) + 32 * hi));      // slot (phi(lo), hi)
    const float* xb[PTC];
#pragma unroll
    for (int pi = 0; pi < PTC; ++pi) xb[pi] = Xt + (32 * pi + lo) * LD + 4 * hi;
    float4 w0[PF][2], w1[PF][2];
    // The activations of k-group g + 1 are requested BEFORE the 16 MFMAs of group g (one ds_read_b128 per point tile feeds 8 MFMAs),
    // into the other of two register sets; a scheduling barrier keeps the request at the head of the group.  (A read issued at the top
    // of its own group stalls the wave for the LDS round trip once per group: the first cut of this kernel, without the double buffer,
    // ran 5 % SLOWER than the k-major kernel.)
    float4 x0[PTC], x1[PTC];

    auto loadW = [&](float4(&w)[PF][2], int g0) {
#pragma unroll
        for (int gi = 0; gi < PF; ++gi)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                if ((!GUARD || g0 + gi < KG) && (!(DBG & 4) || g0 < 2 * PF)) w[gi][mi] = wl[(size_t)((mt0 + mi) * KG + g0 + gi) * 64];
    };
    auto loadX = [&](float4(&x)[PTC], int g) {
        if ((DBG & 8) && g > 1) return;
#pragma unroll
        for (int pi = 0; pi < PTC; ++pi) x[pi] = *reinterpret_cast<const float4*>(xb[pi] + 8 * g);
    };
    auto group = [&](const float4(&w)[2], const float4(&xc)[PTC], float4(&xn)[PTC], int g) {
        if (g + 1 < KG) loadX(xn, g + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pi = 0; pi < PTC; ++pi)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[mi][pi] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(w[mi], j), f4c(xc[pi], j), acc[mi][pi], 0, 0, 0);
    };
    auto comp = [&](const float4(&w)[PF][2], int g0) {      // g0 is even at every call site: even groups read x0, odd groups x1
#pragma unroll
        for (int gi = 0; gi < PF; ++gi) {
            if (!GUARD || g0 + gi < KG) {
                if ((gi & 1) == 0) group(w[gi], x0, x1, g0 + gi);
                else group(w[gi], x1, x0, g0 + gi);
            }
        }
    };

    if constexpr ((DBG & 128) != 0 && KG % 4 == 0 && KG >= 8) {
        // dev experiment: the SAME 32 weight registers as a ring of four one-group sets; the two loads of group g + 3 are issued one after
        // the 4th and one after the 12th MFMA of group g -- an even trickle (one load per 8 MFMAs, three groups = 3 072 cycles ahead)
        // instead of bursts of four loads once per 32 MFMAs
        float4 w[4][2];
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) w[u][mi] = wl[(size_t)((mt0 + mi) * KG + u) * 64];
        loadX(x0, 0);
#pragma unroll 1
        for (int g0 = 0; g0 < KG; g0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = g0 + u;
                float4(&xc)[PTC] = (u & 1) ? x1 : x0;
                float4(&xn)[PTC] = (u & 1) ? x0 : x1;
                if (g + 1 < KG) loadX(xn, g + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int pi = 0; pi < PTC; ++pi)
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi)
                            acc[mi][pi] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(w[u][mi], j), f4c(xc[pi], j), acc[mi][pi], 0, 0, 0);
                    if (j == 0 || j == 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (g + 3 < KG) w[(u + 3) & 3][j >> 1] = wl[(size_t)((mt0 + (j >> 1)) * KG + g + 3) * 64];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        return;
    }
    loadW(w0, 0);
    loadX(x0, 0);
    if constexpr (KG > 2 * PF) {
#pragma unroll 1
        for (int g0 = 0; g0 < KG; g0 += 2 * PF) {
            if (!GUARD || g0 + PF < KG) loadW(w1, g0 + PF);
            comp(w0, g0);
            if (g0 + 2 * PF < KG) loadW(w0, g0 + 2 * PF);
            if (!GUARD || g0 + PF < KG) comp(w1, g0 + PF);
        }
    } else {
        if (PF < KG) loadW(w1, PF);
        comp(w0, 0);
        if (PF < KG) comp(w1, PF);
    }
}

// The wave's staged bias row as 8 float4 (one per quad (mi, q)): requested right behind the GEMM, so that the LDS round trip is over by
// the time the barrier in front of the epilogue releases the wave (read one by one inside the epilogue, each quad waited for its own).
struct Bias8 { float4 b[8]; };
__device__ __forceinline__ Bias8 load_bias8(const float* bias_row) {
    Bias8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.b[i] = *reinterpret_cast<const float4*>(bias_row + 2 * i * LDA);
    return r;
}
// Visit the accumulators as quads: f(mi, pi, q, v[4], b[4]) -- v[i] = feature 32 (mt0 + mi) + 8 q + 2 i + hi of point 32 pi + lo,
// b[i] = its bias.
template <int PTC, class F>
__device__ __forceinline__ void for_quads_t(f32x16 (&acc)[2][PTC], const Bias8& bias, F&& f) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b4 = bias.b[4 * mi + q];
            const float b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int pi = 0; pi < PTC; ++pi) {
                float v[4] = {acc[mi][pi][4 * q + 0], acc[mi][pi][4 * q + 1], acc[mi][pi][4 * q + 2], acc[mi][pi][4 * q + 3]};
                f(mi, pi, q, v, b);
            }
        }
}
// v[i] += b[i] as two packed adds (the same roundings as four scalar adds)
__device__ __forceinline__ void add_bias4v(float (&v)[4], const float (&b)[4]) {
    f32x2v lo = {v[0], v[1]}, hi = {v[2], v[3]};
    lo += f32x2v{b[0], b[1]}; hi += f32x2v{b[2], b[3]};
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
}

// frequency encodings (reference src/renderer/encoder.py:40-54) into the row-major tile, feature k at position ppos(k).
// The point's coordinates are floats ``c0 ..`` of its main-tile row pad (X(row, c)).
template <int L, class XF>
__device__ __forceinline__ void encode3_t(float* At, int kbase, XF&& X, int tid) {
    const int row = tid & 63, part = tid >> 6;
    float* a = At + row * LDA;
    for (int item = part; item < 3 * L; item += 4) {
        const int c = item % 3, i = item / 3;
        float s, co;
        sincosf(X(row, c) * (float)(1 << i), &s, &co);
        a[ppos(kbase + enc_index(3, i, 0, c))] = s;
        a[ppos(kbase + enc_index(3, i, 1, c))] = co;
    }
    if (part == 3) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a[ppos(kbase + c)] = X(row, c);
    }
}
template <int L, class XF>
__device__ __forceinline__ void encode1_t(float* At, int kbase, XF&& X, int tid) {
    const int row = tid & 63, part = tid >> 6;
    float* a = At + row * LDA;
    for (int i = part; i < L; i += 4) {
        float s, co;
        sincosf(X(row, 3) * (float)(1 << i), &s, &co);
        a[ppos(kbase + enc_index(1, i, 0, 0))] = s;
        a[ppos(kbase + enc_index(1, i, 1, 0))] = co;
    }
    if (part == 2) a[ppos(kbase)] = X(row, 3);
}
__device__ __forceinline__ void zero_feats_t(float* At, int k0, int k1, int tid) {
    for (int i = tid; i < (k1 - k0) * 64; i += NTHREADS) At[(i & 63) * LDA + ppos(k0 + (i >> 6))] = 0.f;
}

// out[i][row] = sum_k Wrows[i][k] * X[row][k], k ascending within each of the four 64-wide parts (the order of smalln_partial):
// partial sums to scr[(part * NOUT + i) * 64 + row]; the caller barriers and reduces with smalln_reduce.
template <int NOUT>
__device__ __forceinline__ void smalln_partial_t(const float* Xt, const float* __restrict__ Wrows, int ldw, float* scr, int tid) {
    const int row = tid & 63;
    const int part = __builtin_amdgcn_readfirstlane(tid >> 6);
    float s[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) s[i] = 0.f;
    const int k0 = part * 64;
    const float* xr = Xt + row * LDM + k0;
#pragma unroll 2
    for (int g = 0; g < 8; ++g) {
        const float4 e = *reinterpret_cast<const float4*>(xr + 8 * g), o = *reinterpret_cast<const float4*>(xr + 8 * g + 4);
        const float a[8] = {e.x, o.x, e.y, o.y, e.z, o.z, e.w, o.w};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < NOUT; ++i) s[i] = fmaf(Wrows[i * ldw + k0 + 8 * g + kk], a[kk], s[i]);
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) scr[(part * NOUT + i) * 64 + row] = s[i];
}

#define QT_SYNC() do { if (!(DBG & 16)) __syncthreads(); } while (0)
__device__ unsigned qt_cu_arrivals[4096];      // dev experiment (DBG & 32): arrivals per CU of the current launch (host clears it)
template <bool DEFORM, bool HALF, int DBG = 0>      // DBG: dev builds only (timing experiments; results are garbage): 2 = no activation math,
                                                    // 4 = weights loaded for the first groups of a segment only, 8 = activations likewise, 16 = no barriers
__global__ __launch_bounds__(NTHREADS, 2) void k_query_sdf_t(PointSrc src, Tabs tb, const float4* __restrict__ packed,
                                                          const float* __restrict__ weff, float* __restrict__ sdf_out, int ld_out,
                                                          const int* __restrict__ ray_done, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    float* aux = lds + QT_MAIN_FLOATS;
    float* red = aux;         // [4][<= 3][64]: aliases the encoding rows (and the bias pads of rows 0..12), dead by the time the tiny last layers run
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int PTC = HALF ? 1 : 2;
    constexpr int PTS = HALF ? 32 : 64;
    const int row0 = blockIdx.x * PTS;
    if (ray_done != nullptr) {      // block-wise ray marching: a tile whose rays already have their first sign change is skipped
        const int r_first = row0 / src.n_per_ray, r_last = min(row0 + PTS - 1, src.M - 1) / src.n_per_ray;
        bool all_done = true;
        for (int r = r_first; r <= r_last; ++r) all_done = all_done && ray_done[r] != 0;
        if (all_done) return;       // workgroup-uniform
    }
    if (DBG & 32) {      // stagger: the SECOND workgroup to arrive on a CU waits ``dbg`` x 1024 cycles before it starts
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned cu = ((((xcc & 0xf) * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xf)) & 4095;
        unsigned* arrival = reinterpret_cast<unsigned*>(lds);      // (no static LDS: it would shift the dynamic base off its 16-byte alignment)
        if (tid == 0) *arrival = atomicAdd(&qt_cu_arrivals[cu], 1u);
        __syncthreads();
        const unsigned arr = *arrival;
        __syncthreads();
        if (arr == 1) {
            const long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < (long long)dbg * 1024) __builtin_amdgcn_s_sleep(32);
        }
    }
    const int lo = lane & 31, hi = lane >> 5;
    auto X = [&](int row, int c) -> float& { return mainT[row * LDM + HID + c]; };
    // this wave's staged bias: entry p = 64 wave + lane of the layer's permuted bias sits in pad float (p & 3) of encoding row p >> 2
    float* bias_slot = aux + (16 * wave + (lane >> 2)) * LDA + 56 + (lane & 3);
    const int bias_feat = pfeat(64 * wave + lane);
    const float* bias_row = aux + (16 * wave + hi) * LDA + 56;        // + (8 mi + 2 q) rows: the float4 of quad (mi, q)
    // epilogue store base: [point 32 pi + lo][position 64 wave + 4 hi] (+ 32 mi + 8 q as an immediate)
    float* est[PTC];
#pragma unroll
    for (int pi = 0; pi < PTC; ++pi) est[pi] = mainT + (32 * pi + lo) * LDM + 64 * wave + 4 * hi;

    if (tid < 64) {
        float x[3], t, d[3];
        load_point(src, tid < PTS ? row0 + tid : src.M, x, t, d);
        X(tid, 0) = x[0]; X(tid, 1) = x[1]; X(tid, 2) = x[2]; X(tid, 3) = t;
    }
    QT_SYNC();

    float bn;          // next layer's bias of this lane's staged entry (requested a layer ahead)
    if (DEFORM) {
        // ---- deformation MLP, value only: x_c = x + MLP([enc6(x), enc6(t)]) ----
        encode3_t<6>(aux, 0, X, tid);
        encode1_t<6>(aux, 39, X, tid);
        zero_feats_t(aux, 52, 56, tid);
        *bias_slot = weff[tb.boff[NET_D * LAYERS + 0] + bias_feat];
        bn = weff[tb.boff[NET_D * LAYERS + 1] + bias_feat];
        QT_SYNC();
        {
            f32x16 acc[2][PTC];
            acc_zero(acc);
            gemm_seg_t<7, PTC, LDA, DBG>(acc, aux, packed + tb.segoff[DF0], 2 * wave, lane, dbg);
            const Bias8 bias = load_bias8(bias_row);
            for_quads_t<PTC>(acc, bias, [&](int mi, int pi, int q, float(&v)[4], const float(&b)[4]) {
                add_bias4v(v, b);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (DBG & 2) ? v[i] : relu1(v[i]);
                *reinterpret_cast<float4*>(est[pi] + 32 * mi + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
            });
            *bias_slot = bn;
        }
        QT_SYNC();
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            f32x16 acc[2][PTC];
            acc_zero(acc);
            if (l < 7) bn = weff[tb.boff[NET_D * LAYERS + l + 1] + bias_feat];
            gemm_seg_t<32, PTC, LDM, DBG, (DBG & 64) ? 4 : 2>(acc, mainT, packed + tb.segoff[DF0 + l], 2 * wave, lane, dbg);
            const Bias8 bias = load_bias8(bias_row);
            QT_SYNC();
            if (l == 3 && wave == 3) {      // (wave-uniform) IDR skip: next input = [h(204) | enc(52)] (1/sqrt2 folded into W4)
                for_quads_t<PTC>(acc, bias, [&](int mi, int pi, int q, float(&v)[4], const float(&b)[4]) {
                    add_bias4v(v, b);
                    const float* ar = aux + (32 * pi + lo) * LDA + 4 * hi;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // feature 192 + 32 mi + 8 q + 2 i + hi >= 204 <=> 32 mi + 8 q + 2 i >= 12 (hi only moves it inside a pair)
                        const int e2 = 32 * mi + 8 * q + 2 * i - 12;          // encoding element e = e2 + hi: position (e2 & ~7) + 4 hi + ((e2 >> 1) & 3)
                        v[i] = e2 >= 0 ? ar[(e2 & ~7) + ((e2 >> 1) & 3)] : relu1(v[i]);
                    }
                    *reinterpret_cast<float4*>(est[pi] + 32 * mi + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
                });
            } else {
                for_quads_t<PTC>(acc, bias, [&](int mi, int pi, int q, float(&v)[4], const float(&b)[4]) {
                    add_bias4v(v, b);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (DBG & 2) ? v[i] : relu1(v[i]);
                    *reinterpret_cast<float4*>(est[pi] + 32 * mi + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
                });
            }
            if (l < 7) *bias_slot = bn;
            QT_SYNC();
        }
        smalln_partial_t<3>(mainT, weff + tb.woff[NET_D * LAYERS + 8], 256, red, tid);
        QT_SYNC();
        if (tid < 192) {
            const int i = tid >> 6, row = tid & 63;
            X(row, i) += smalln_reduce<3>(red, i, row) + weff[tb.boff[NET_D * LAYERS + 8] + i];
        }
        QT_SYNC();
    }

    // ---- SDF MLP on x_c, output column 0 only ----
    encode3_t<6>(aux, 0, X, tid);
    zero_feats_t(aux, 39, 40, tid);
    *bias_slot = weff[tb.boff[NET_S * LAYERS + 0] + bias_feat];
    bn = weff[tb.boff[NET_S * LAYERS + 1] + bias_feat];
    QT_SYNC();
    {
        f32x16 acc[2][PTC];
        acc_zero(acc);
        gemm_seg_t<5, PTC, LDA, DBG>(acc, aux, packed + tb.segoff[SF0], 2 * wave, lane, dbg);
        const Bias8 bias = load_bias8(bias_row);
        for_quads_t<PTC>(acc, bias, [&](int mi, int pi, int q, float(&v)[4], const float(&b)[4]) {
            add_bias4v(v, b);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (DBG & 2) ? v[i] : softplus100(v[i]);
            *reinterpret_cast<float4*>(est[pi] + 32 * mi + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
        });
        *bias_slot = bn;
    }
    QT_SYNC();
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        f32x16 acc[2][PTC];
        acc_zero(acc);
        if (l < 7) bn = weff[tb.boff[NET_S * LAYERS + l + 1] + bias_feat];
        const int seg = l <= 4 ? SF0 + l : SF0 + l + 1;
        gemm_seg_t<32, PTC, LDM, DBG, (DBG & 64) ? 4 : 2>(acc, mainT, packed + tb.segoff[seg], 2 * wave, lane, dbg);
        if (l == 4) gemm_seg_t<5, PTC, LDA, DBG>(acc, aux, packed + tb.segoff[SF4A], 2 * wave, lane, dbg);   // NeRF skip: + enc part
        const Bias8 bias = load_bias8(bias_row);
        QT_SYNC();
        for_quads_t<PTC>(acc, bias, [&](int mi, int pi, int q, float(&v)[4], const float(&b)[4]) {
            add_bias4v(v, b);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (DBG & 2) ? v[i] : softplus100(v[i]);
            *reinterpret_cast<float4*>(est[pi] + 32 * mi + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
        });
        if (l < 7) *bias_slot = bn;
        QT_SYNC();
    }
    smalln_partial_t<1>(mainT, weff + tb.woff[NET_S * LAYERS + 8], 256, red, tid);
    QT_SYNC();
    if (tid < PTS && row0 + tid < src.M) {
        const int i = row0 + tid;
        const size_t o = ld_out > 0 ? (size_t)(i / src.n_per_ray) * ld_out + (i % src.n_per_ray) : (size_t)i;   // [ray][ld_out] or flat
        sdf_out[o] = smalln_reduce<1>(red, 0, tid) + weff[tb.boff[NET_S * LAYERS + 8]];
    }
}

int query_sdf_t(const PointSrc& src, const float* packed, const float* weff, float* sdf_out, int use_deform, hipStream_t st, int ld_out,
                const int* ray_done, bool half, int dbg) {
    static DeviceOnce attr_done;
    if (attr_done.first()) {
        if (int e = allow_big_lds(k_query_sdf_t<true, false>, QT_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_query_sdf_t<false, false>, QT_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_query_sdf_t<true, true>, QT_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_query_sdf_t<false, true>, QT_LDS_BYTES)) return e;
        attr_done.done();
    }
    if (src.M <= 0) return ST_OK;
    const Tabs tb = make_tabs();
    const int pts = half ? 32 : 64;
    const dim3 grid((src.M + pts - 1) / pts), block(NTHREADS);
    const float4* pk = reinterpret_cast<const float4*>(packed);
    ScopedTimer tm(ray_done ? KID_QUERY_EXIT : KID_QUERY, src.M, st);
#ifdef ES_DEV_SWITCHES
    if (dbg > 1 && use_deform && !half) {      // timing experiments (ES_QT=100 + mask)
        static bool attr = false;
        if (!attr) {
            allow_big_lds(k_query_sdf_t<true, false, 2>, QT_LDS_BYTES); allow_big_lds(k_query_sdf_t<true, false, 4>, QT_LDS_BYTES);
            allow_big_lds(k_query_sdf_t<true, false, 8>, QT_LDS_BYTES); allow_big_lds(k_query_sdf_t<true, false, 16>, QT_LDS_BYTES);
            allow_big_lds(k_query_sdf_t<true, false, 30>, QT_LDS_BYTES); allow_big_lds(k_query_sdf_t<true, false, 12>, QT_LDS_BYTES);
            attr = true;
        }
        if (dbg == 128) {
            static bool a128 = false;
            if (!a128) { allow_big_lds(k_query_sdf_t<true, false, 128>, QT_LDS_BYTES); a128 = true; }
            hipLaunchKernelGGL((k_query_sdf_t<true, false, 128>), grid, block, QT_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done, 0);
            return hip_last("query_sdf_t");
        }
        if (dbg == 64) {
            static bool a64 = false;
            if (!a64) { allow_big_lds(k_query_sdf_t<true, false, 64>, QT_LDS_BYTES); a64 = true; }
            hipLaunchKernelGGL((k_query_sdf_t<true, false, 64>), grid, block, QT_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done, 0);
            return hip_last("query_sdf_t");
        }
        if (dbg == 32) {
            static bool a32 = false;
            if (!a32) { allow_big_lds(k_query_sdf_t<true, false, 32>, QT_LDS_BYTES); a32 = true; }
            static const int delay = getenv("ES_QT_DELAY") ? atoi(getenv("ES_QT_DELAY")) : 0;
            void* cnt = nullptr;
            hipGetSymbolAddress(&cnt, HIP_SYMBOL(qt_cu_arrivals));
            hipMemsetAsync(cnt, 0, sizeof(unsigned) * 4096, st);
            hipLaunchKernelGGL((k_query_sdf_t<true, false, 32>), grid, block, QT_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done, delay);
            return hip_last("query_sdf_t");
        }
        switch (dbg) {
            case 2: hipLaunchKernelGGL((k_query_sdf_t<true, false, 2>), grid, block, QT_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done, 0); break;
            case 4: hipLaunchKernelGGL((k_query_sdf_t<true, false, 4>), grid, block, QT_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done, 0); break;
            case 8: hipLaunchKernelGGL((k_query_sdf_t<true, false, 8>), grid, block, QT_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done, 0); break;
            case 16: hipLaunchKernelGGL((k_query_sdf_t<true, false, 16>), grid, block, QT_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done, 0); break;
            case 12: hipLaunchKernelGGL((k_query_sdf_t<true, false, 12>), grid, block, QT_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done, 0); break;
            default: hipLaunchKernelGGL((k_query_sdf_t<true, false, 30>), grid, block, QT_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done, 0); break;
        }
        return hip_last("query_sdf_t");
    }
#endif
    if (use_deform) {
        if (half) hipLaunchKernelGGL((k_query_sdf_t<true, true>), grid, block, QT_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done, dbg);
        else hipLaunchKernelGGL((k_query_sdf_t<true, false>), grid, block, QT_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done, dbg);
    } else {
        if (half) hipLaunchKernelGGL((k_query_sdf_t<false, true>), grid, block, QT_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done, dbg);
        else hipLaunchKernelGGL((k_query_sdf_t<false, false>), grid, block, QT_LDS_BYTES, st, src, tb, pk, weff, sdf_out, ld_out, ray_done, dbg);
    }
    return hip_last("query_sdf_t");
}

}  // namespace es

#endif  // ES_DEV_SWITCHES
