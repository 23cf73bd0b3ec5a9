// Weight-gradient GEMMs of the backward pass:  dW[n][k] += sum_m dA[m][n] * X[m][k]  for every layer, from the
// (layer input X, pre-activation adjoint dA) pairs streamed to the workspace by the forward/backward chains.
// All problems of one network are batched in ONE launch (grouped GEMM): independent workgroup tasks (problem, 256 x 128
// tile of dW, row chunk) whose operand panels are staged through LDS, contraction on v_mfma_f32_32x32x2_f32, results reduced
// with fp32 atomics (few row chunks per output tile).  Bias gradients are column sums of dA taken on the fly by the
// k-block-0 tasks.  The last layers' tiny-N gradients (3 / 1 outputs: an HBM stream over the layer input) are sliced over
// the GEMM tasks of the same launch.
// Deterministic mode (a scratch buffer is passed): every task stores its partial tile / column sums / small slices to its own
// slot of the scratch instead of issuing atomics, and k_wgrad_reduce sums the slots in a FIXED order (one pass per group of
// problems that accumulate into the same output): bit-identical gradients from run to run, at ~80 MB of extra traffic per launch.
#include <hip/hip_runtime.h>

#include "arch.h"
#include "chain_common.h"
#include "launch.h"
#include "tabs.h"
#include "timing.h"
#include "workspace.h"

namespace es {

#ifdef ES_PROFILE_WGRAD       // dev builds only: cycle stamps of block 0 / thread 0 inside the fp32 task (tools/wgrad_profile.py)
__device__ long long w_prof[128];
#define W_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) w_prof[i] = __builtin_readcyclecounter(); } while (0)
extern "C" int es_debug_w_profile(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(w_prof), sizeof(long long) * (n < 128 ? n : 128)); }
#else
#define W_STAMP(i) do {} while (0)
#endif

constexpr int WG_MAX_PROBS = 20;

struct WgProb {
    const float* X; const float* dA; float* out; float* bias_out;
    int ldx, lda, ldo, M, K, N, bias_stride, task_begin;
    int x_frag, a_frag;      // operand stored as fragment-ordered [64 x 256] tiles (chain_common.h frag_off) instead of row-major
    int round;               // deterministic mode: reduction pass (problems accumulating into the same output get consecutive passes)
};
// tiny-N layers (3 / 1 outputs): out[n][k] += sum_m dA[m][n] X[m][k] — a latency-bound HBM stream over X (dA == nullptr means
// dA = 1: column sums of X).  Every GEMM task of the hosting launch streams a slice of it, half of the tasks before and
// half after their GEMM, so that the two workgroups of a CU are out of phase and the matrix pipes stay busy meanwhile.
struct WgSmall {
    const float* X; const float* dA; float* out; float* bias_out;
    int ldx, lda, ldo, M, K, N, bias_stride, x_frag, round;
};
constexpr int WG_MAX_SMALL = 4;
constexpr int WG_MAX_TASKS = 512;                // one full round of workgroup slots (launch_group)
struct WgArgs {
    WgProb p[WG_MAX_PROBS];
    WgSmall s[WG_MAX_SMALL];
    int nprob, total_tasks, MC, nsmall;
    float* det;              // deterministic mode: scratch of WG_DET_FLOATS floats (nullptr: fp32 atomics)
};
// scratch layout: [task][256][128] partial tiles | [task][256] partial bias sums | [small][task][5][256] (4 outputs + bias row)
constexpr size_t WG_DET_TILE = (size_t)256 * 128;
constexpr size_t WG_DET_BIAS_OFF = (size_t)WG_MAX_TASKS * WG_DET_TILE;
constexpr size_t WG_DET_SMALL_OFF = WG_DET_BIAS_OFF + (size_t)WG_MAX_TASKS * 256;
constexpr size_t WG_DET_FLOATS = WG_DET_SMALL_OFF + (size_t)WG_MAX_SMALL * WG_MAX_TASKS * 5 * 256;

// task index of a problem <-> (k block, row chunk).  The kblk tasks of one row chunk read the same dA rows: they get block ids 8
// apart (same XCD under the round-robin block -> XCD dispatch, started back to back) so that the second reader hits that XCD's
// L2 instead of HBM
__host__ __device__ inline void wg_decode(int local, int kblk, int nchunk, int& kb, int& mc) {
    if (kblk == 2) {
        const int grp = local / 16, j = local % 16;
        const int full = (nchunk / 8) * 8;                 // chunks covered by complete groups of 8
        if (grp * 8 < full) { mc = grp * 8 + (j & 7); kb = j >> 3; }
        else { const int rem = local - 2 * full; kb = rem & 1; mc = full + (rem >> 1); }
    } else { kb = local % kblk; mc = local / kblk; }
}
__host__ __device__ inline int wg_encode(int kb, int mc, int kblk, int nchunk) {
    if (kblk == 2) {
        const int full = (nchunk / 8) * 8;
        return mc < full ? (mc / 8) * 16 + (mc & 7) + 8 * kb : 2 * full + (((mc - full) << 1) | kb);
    }
    return mc * kblk + kb;
}

// One workgroup (8 waves) = one task: a [256 x 128] tile of dW (all 256 output features x 128 input features) over a chunk
// of rows.  Both operand panels are staged through LDS in 16-row stages (double buffered, one barrier per stage, loads two
// stages ahead), so every dA / X element is read from HBM once per task instead of once per 64x64 wave tile.  48 KB of LDS
// and <= 128 registers => two workgroups (16 waves) per CU whose barrier phases interleave.
// wave w: n-block w&3 (64 features = 2 MFMA tiles interleaved 2i+t) x k-half w>>2 (64 features = 2 tiles interleaved 2j+t').
constexpr int WG_THREADS = 512;
constexpr int WG_R = 16;                         // rows per stage
constexpr int WG_KW = 128;                       // input features per task
constexpr int WG_LDS_FLOATS = 2 * WG_R * (256 + WG_KW);

// AF / XF: dA / X arrive fragment-ordered (a float4 = 4 consecutive rows of one column: scattered into the row-major LDS
// panels with four ds_write_b32; the rows of a 16-row stage are the quads q = 2j, 2j+1 of row tile ri, both lane halves)
template <bool AF, bool XF, bool DET>
__device__ __forceinline__ void wgrad_task(const WgProb& P, int kb, int m0, int m1, float* lds, float* det_tile, float* det_bias) {
    constexpr int KW = WG_KW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = w & 3, kh = w >> 2;
    const int lo = lane & 31, hi = lane >> 5;
    auto Apan = [&](int buf) { return lds + buf * (WG_R * 256); };
    auto Bpan = [&](int buf) { return lds + 2 * WG_R * 256 + buf * (WG_R * KW); };
    const int kcol0 = kb * KW;

    typedef float v4f_t __attribute__((ext_vector_type(4)));
    f32x16 acc[2][2];
    acc_zero(acc);
    // bias gradient = column sums of dA over the rows r with r % bias_stride == 0 (strides 1 or 4; stages start at multiples
    // of 16): taken from the staging registers on their way to LDS (thread tid holds columns 4*(tid&63).. of rows tid>>6 and
    // 8 + (tid>>6) of every stage), so the MFMA loop stays one branch-free basic block per stage
    const bool do_bias = P.bias_out != nullptr && kb == 0;
    const float bmask = (do_bias && (AF || ((tid >> 6) & (P.bias_stride - 1)) == 0)) ? 1.f : 0.f;       // AF: stride 1 only
    v4f_t bsum = {0.f, 0.f, 0.f, 0.f};

    // global -> register loads of one 16-row stage (two stages in flight: sets 0/1), register -> LDS stores.
    // Plain ext-vector locals (not HIP float4 structs) so that they stay in VGPRs.
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f vzero = {0.f, 0.f, 0.f, 0.f};
    const int fa1 = tid + WG_THREADS;
    // row-major operands: thread -> (row, 4 columns); offsets relative to the stage's first row
    const size_t offA0 = (size_t)(tid >> 6) * P.lda + 4 * (tid & 63), offA1 = (size_t)(fa1 >> 6) * P.lda + 4 * (fa1 & 63);
    const int colB = kcol0 + 4 * (tid % (KW / 4));
    const size_t offB = (size_t)(tid / (KW / 4)) * P.ldx + colB;
    const bool okB = XF || colB < P.ldx;
    // fragment-ordered operands: unit u -> (wave block u>>8, ni, quad parity qq, lane) of the stage
    const int fqq = (tid >> 6) & 1, fni = (tid >> 7) & 1, fw = tid >> 8;                // fw in 0..1 (unit tid), +2 for unit tid+512
    const size_t foffA0 = (size_t)(((fw * 16 + fni * 4 + fqq) * 64 + lane) * 4), foffA1 = foffA0 + (size_t)2 * 16 * 64 * 4;
    const size_t foffB = (size_t)((((2 * kb + fw) * 16 + fni * 4 + fqq) * 64 + lane) * 4);
    const int flrow = 8 * fqq + 4 * hi;                                                // first of the 4 local rows of the unit
    const int flcolA = 64 * fw + 32 * fni + lo, flcolB = 64 * fw + 32 * fni + lo;      // A: + 128 for the second unit
    auto stage_ptr = [&](const float* base, int ld, bool frag, int m) {
        return frag ? base + (size_t)(m >> 6) * (64 * 256) + (size_t)((8 * ((m >> 5) & 1) + 2 * ((m >> 4) & 1)) * 256) : base + (size_t)m * ld;
    };
#define WG_GLOAD(S, m)                                                                                \
    {                                                                                                 \
        const float* pa = stage_ptr(P.dA, P.lda, AF, m);                                              \
        S##a0 = *reinterpret_cast<const v4f*>(pa + (AF ? foffA0 : offA0));                            \
        S##a1 = *reinterpret_cast<const v4f*>(pa + (AF ? foffA1 : offA1));                            \
        const float* px = stage_ptr(P.X, P.ldx, XF, m);                                               \
        S##b0 = okB ? *reinterpret_cast<const v4f*>(px + (XF ? foffB : offB)) : vzero;                \
    }
#define WG_SSTORE(S, buf)                                                                             \
    {                                                                                                 \
        if constexpr (AF) {                                                                           \
            float* la = Apan(buf) + flrow * 256 + flcolA;                                             \
            la[0] = S##a0[0]; la[256] = S##a0[1]; la[512] = S##a0[2]; la[768] = S##a0[3];             \
            la[128] = S##a1[0]; la[128 + 256] = S##a1[1]; la[128 + 512] = S##a1[2]; la[128 + 768] = S##a1[3]; \
            bsum[0] += bmask * (S##a0[0] + S##a0[1] + S##a0[2] + S##a0[3]);                           \
            bsum[1] += bmask * (S##a1[0] + S##a1[1] + S##a1[2] + S##a1[3]);                           \
        } else {                                                                                      \
            *reinterpret_cast<v4f*>(Apan(buf) + 4 * tid) = S##a0;                                     \
            *reinterpret_cast<v4f*>(Apan(buf) + 4 * fa1) = S##a1;                                     \
            bsum += bmask * (S##a0 + S##a1);                                                          \
        }                                                                                             \
        if constexpr (XF) {                                                                           \
            float* lb = Bpan(buf) + flrow * KW + flcolB;                                              \
            lb[0] = S##b0[0]; lb[KW] = S##b0[1]; lb[2 * KW] = S##b0[2]; lb[3 * KW] = S##b0[3];        \
        } else {                                                                                      \
            *reinterpret_cast<v4f*>(Bpan(buf) + 4 * tid) = S##b0;                                     \
        }                                                                                             \
    }
    auto compute = [&](int buf) {
        const float* A = Apan(buf) + nb * 64 + 2 * lo;
        const float* B = Bpan(buf) + kh * 64 + 2 * lo;
#pragma unroll
        for (int s = 0; s < WG_R / 2; ++s) {
            const float2 av = *reinterpret_cast<const float2*>(A + (2 * s + hi) * 256);
            const float2 bv = *reinterpret_cast<const float2*>(B + (2 * s + hi) * KW);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[1][1], 0, 0, 0);
        }
    };

    // software pipeline: stage st computes from LDS[st&1] while the loads of stages st+1 (landing) and st+2 (just issued)
    // are in flight; one barrier per stage.  nst is even (chunks are multiples of 64 rows).
    const int nst = (m1 - m0) / WG_R;
    v4f p0a0, p0a1, p0b0, p1a0, p1a1, p1b0;
    WG_GLOAD(p0, m0);
    WG_GLOAD(p1, m0 + WG_R);
    WG_SSTORE(p0, 0);
    __syncthreads();
#pragma unroll 1
    W_STAMP(0);
    for (int st = 0; st < nst; st += 2) {
        const bool stamp = st == 32;                     // one iteration in the steady state
        if (stamp) W_STAMP(1);
        // (Issuing these loads unconditionally -- clamped to the last stage -- lets the compiler count outstanding loads exactly
        // (s_waitcnt vmcnt(3) instead of vmcnt(0) before the second stage store); measured: no change, 14.92 vs 14.96 ms per step.)
        if (st + 2 < nst) WG_GLOAD(p0, m0 + WG_R * (st + 2));
        compute(0);
        if (stamp) W_STAMP(2);
        WG_SSTORE(p1, 1);                                // stage st+1 (loaded one iteration ago)
        if (stamp) W_STAMP(3);
        __syncthreads();
        if (stamp) W_STAMP(4);
        if (st + 3 < nst) WG_GLOAD(p1, m0 + WG_R * (st + 3));
        compute(1);
        if (stamp) W_STAMP(5);
        if (st + 2 < nst) WG_SSTORE(p0, 0);              // stage st+2
        if (stamp) W_STAMP(6);
        __syncthreads();
        if (stamp) W_STAMP(7);
    }
    W_STAMP(8);
#ifdef ES_PROFILE_WGRAD
    if (blockIdx.x == 0 && threadIdx.x == 0) w_prof[20] = nst;
#endif
#undef WG_GLOAD
#undef WG_SSTORE
    // acc[t][tp][r]: n = nb*64 + 2*i + t, i = (r&3) + 8*(r>>2) + 4*hi ; k = kb*128 + kh*64 + 2*lo + tp
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const int k = kcol0 + kh * 64 + 2 * lo + tp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * hi) + t;
                if constexpr (DET) det_tile[n * WG_KW + (k - kcol0)] = acc[t][tp][r];
                else if (n < P.N && k < P.K) atomicAdd(P.out + (size_t)n * P.ldo + k, acc[t][tp][r]);
            }
        }
    if (do_bias) {      // workgroup-uniform: reduce the partial column sums through LDS (all stages consumed)
        __syncthreads();
        if constexpr (AF) {         // 4 threads (qq, hi) per column; unit tid holds column flcolA, unit tid+512 column flcolA + 128
            lds[(fqq * 2 + hi) * 256 + flcolA] = bsum[0];
            lds[(fqq * 2 + hi) * 256 + flcolA + 128] = bsum[1];
        } else {
            *reinterpret_cast<v4f*>(lds + 4 * tid) = bsum;
        }
        __syncthreads();
        if (tid < 256) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < (AF ? 4 : 8); ++r) s += lds[r * 256 + tid];
            if constexpr (DET) det_bias[tid] = s;
            else if (tid < P.N) atomicAdd(P.bias_out + tid, s);
        }
    }
    W_STAMP(9);
}

// Slice `slot` of `nslots` of a small problem: thread = (input feature k, row-block parity); 16-row blocks, the loads of two
// steps (2 x 16 rows of X per thread) in flight, the <= 4 adjoint columns go through LDS (double buffered, one barrier per step).
constexpr int WS_ROWS = 16;
template <bool DET>
__device__ __forceinline__ void wgrad_small_task(const WgSmall& P, int slot, int nslots, float* lds, float* det_slot) {
    float(*sd)[WS_ROWS][4] = reinterpret_cast<float(*)[WS_ROWS][4]>(lds);       // [half * 2 + buffer]
    const int tid = threadIdx.x, k = tid & 255, half = tid >> 8;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const int nblk = (P.M + WS_ROWS - 1) / WS_ROWS;
    const int step = 2 * nslots;
    // X rows up to the next multiple of 64 exist (finite padding rows of the workspace; their adjoints are read as 0)
    typedef float v4f __attribute__((ext_vector_type(4)));
    auto loadx = [&](float(&x)[WS_ROWS], int b0) {
        const int m0 = b0 + half < nblk ? (b0 + half) * WS_ROWS : 0;
        if (P.x_frag) {     // fragment-ordered [64 x 256] tiles: the 16 rows of column k are 4 float4 (quad parity x lane half)
            const float* xp = P.X + (size_t)(m0 >> 6) * (64 * 256) + (size_t)((8 * ((m0 >> 5) & 1) + 2 * ((m0 >> 4) & 1)) * 256)
                              + (size_t)((((k >> 6) * 16 + ((k >> 5) & 1) * 4) * 64 + (k & 31)) * 4);
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(xp + (qq * 64 + hh * 32) * 4));
#pragma unroll
                    for (int i = 0; i < 4; ++i) x[8 * qq + 4 * hh + i] = t[i];
                }
        } else {
            const float* xp = P.X + (size_t)m0 * P.ldx + k;
#pragma unroll
            for (int r = 0; r < WS_ROWS; ++r) x[r] = __builtin_nontemporal_load(xp + (size_t)r * P.ldx);
        }
    };
    auto consume = [&](const float(&x)[WS_ROWS], int b0, int it) {
        const bool live = b0 + half < nblk;                                     // half-uniform
        const int m0 = live ? (b0 + half) * WS_ROWS : 0;
        float(&sdb)[WS_ROWS][4] = sd[half * 2 + (it & 1)];
        if (k < WS_ROWS * 4) {
            const int r = k >> 2, n = k & 3;
            sdb[r][n] = (live && m0 + r < P.M && n < P.N) ? (P.dA ? P.dA[(size_t)(m0 + r) * P.lda + n] : 1.f) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < WS_ROWS; ++r)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = fmaf(sdb[r][n], x[r], acc[n]);
        if (P.bias_out && k < P.N)
            for (int r = 0; r < WS_ROWS; ++r)
                if (((m0 + r) % P.bias_stride) == 0) bsum += sdb[r][k];
    };
    float xa[WS_ROWS], xb[WS_ROWS];
    int b0 = 2 * slot, it = 0;
    if (b0 < nblk) loadx(xa, b0);
#pragma unroll 1
    while (b0 < nblk) {
        if (b0 + step < nblk) loadx(xb, b0 + step);
        consume(xa, b0, it++);
        b0 += step;
        if (b0 >= nblk) break;
        if (b0 + step < nblk) loadx(xa, b0 + step);
        consume(xb, b0, it++);
        b0 += step;
    }
    // the two row-block parities are summed through LDS: one atomic per (output, k) and task
    __syncthreads();
    float* red = lds;
    if (half) {
#pragma unroll
        for (int n = 0; n < 4; ++n) red[n * 256 + k] = acc[n];
        red[1024 + k] = bsum;
    }
    __syncthreads();
    if (!half) {
        if constexpr (DET) {        // det_slot: [5][256] = 4 output rows + bias row of this (small problem, task)
            for (int n = 0; n < 4; ++n) det_slot[n * 256 + k] = acc[n] + red[n * 256 + k];
            det_slot[4 * 256 + k] = bsum + red[1024 + k];
        } else {
            for (int n = 0; n < P.N; ++n) atomicAdd(P.out + (size_t)n * P.ldo + k, acc[n] + red[n * 256 + k]);
            if (P.bias_out && k < P.N) atomicAdd(P.bias_out + k, bsum + red[1024 + k]);
        }
    }
    __syncthreads();
}

// NET only names the instantiation (0 deform, 1 sdf, 2 colour) so that profilers list the three grouped launches separately
template <int NET, bool DET>
__global__ __launch_bounds__(WG_THREADS, 4) void k_wgrad(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    const int task = blockIdx.x;
    auto small_slot = [&](int i) { return DET ? a.det + WG_DET_SMALL_OFF + ((size_t)i * WG_MAX_TASKS + task) * (5 * 256) : nullptr; };
    // blocks b and b + (tasks of the round)/2 tend to share a CU: one of them streams its small slices first, the other last
    const bool small_first = task < a.total_tasks / 2;
    if (small_first)
#pragma unroll 1
        for (int i = 0; i < a.nsmall; ++i) wgrad_small_task<DET>(a.s[i], task, a.total_tasks, wlds, small_slot(i));
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < a.nprob; ++i)
        if (task >= a.p[i].task_begin) pi = i;
    const WgProb& P = a.p[pi];
    const int local = task - P.task_begin;
    const int kblk = (P.K + WG_KW - 1) / WG_KW;
    int kb, mc;
    wg_decode(local, kblk, (P.M + a.MC - 1) / a.MC, kb, mc);
    const int m0 = mc * a.MC, m1 = min(m0 + a.MC, P.M);
    float* dt = DET ? a.det + (size_t)task * WG_DET_TILE : nullptr;
    float* db = DET ? a.det + WG_DET_BIAS_OFF + (size_t)task * 256 : nullptr;
    if constexpr (NET == 1) {        // only the SDF network's stacks are fragment-ordered
        if (P.a_frag) {
            if (P.x_frag) wgrad_task<true, true, DET>(P, kb, m0, m1, wlds, dt, db);
            else wgrad_task<true, false, DET>(P, kb, m0, m1, wlds, dt, db);
        } else {
            if (P.x_frag) wgrad_task<false, true, DET>(P, kb, m0, m1, wlds, dt, db);
            else wgrad_task<false, false, DET>(P, kb, m0, m1, wlds, dt, db);
        }
    } else {
        wgrad_task<false, false, DET>(P, kb, m0, m1, wlds, dt, db);
    }
    if (!small_first) {
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < a.nsmall; ++i) wgrad_small_task<DET>(a.s[i], task, a.total_tasks, wlds, small_slot(i));
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// OPT-IN split-precision variant (flag PF_X3; csrc/query_x3.hip explains the arithmetic): the same task decomposition, but both
// operand panels are split EXACTLY into three bf16 planes on their way into LDS and the contraction runs on
// v_mfma_f32_32x32x16_bf16 (six partial products per tile, fp32 accumulation): ~2.7x the fp32 matrix rate, which turns these GEMMs
// from MFMA-bound into HBM-bound (each operand element is still read once per task).  A stage = 16 rows = one MFMA k-step.
// LDS per plane: units of 16 B = 8 consecutive ROWS of one column, [row group (2)][column] -> the A / B fragments (lane = column,
// 8 rows) are single conflict-free ds_read_b128.  Every thread stages one (row group, column) unit of dA (512 units per stage)
// and threads 0..255 one unit of X: 8 dword loads (row-major operands, coalesced over columns) or 2 float4 loads (fragment-ordered).
typedef unsigned wx_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wx_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int WX_R = 16;
constexpr int WX_A_PLANE = 2 * 256 * 16;                     // 8 KiB
constexpr int WX_B_PLANE = 2 * WG_KW * 16;                   // 4 KiB
constexpr int WX_BUF = 3 * (WX_A_PLANE + WX_B_PLANE);        // 36 KiB per stage buffer
constexpr int WX_LDS_BYTES = 4 * WX_BUF;                     // ring of four: 144 KiB, one workgroup (8 waves, <= 256 registers) per CU
static_assert(WX_LDS_BYTES >= WG_LDS_FLOATS * 4, "the small-layer slices and the bias reduction reuse the buffer as float scratch");

__device__ __forceinline__ unsigned wx_cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// 8 fp32 values -> three bf16x8 planes with v = h + m + l exactly
__device__ __forceinline__ void wx_split8(const float (&v)[8], wx_u32x4& h, wx_u32x4& m, wx_u32x4& l) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x0 = v[2 * j], x1 = v[2 * j + 1];
        const unsigned hh = wx_cvt_pk(x0, x1);
        float r0 = x0 - __uint_as_float(hh << 16), r1 = x1 - __uint_as_float(hh & 0xffff0000u);
        const unsigned mm = wx_cvt_pk(r0, r1);
        r0 -= __uint_as_float(mm << 16); r1 -= __uint_as_float(mm & 0xffff0000u);
        h[j] = hh; m[j] = mm; l[j] = wx_cvt_pk(r0, r1);
    }
}

struct WxRegs { float a[8]; float b[8]; };

template <bool AF, bool XF, bool DET>
__device__ __forceinline__ void wgrad_task_x3(const WgProb& P, int kb, int m0, int m1, unsigned char* lds, float* det_tile, float* det_bias) {
    constexpr int KW = WG_KW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = w & 3, kh = w >> 2;
    const int lo = lane & 31, hi = lane >> 5;
    const int kcol0 = kb * KW;
    typedef float v4f __attribute__((ext_vector_type(4)));

    f32x16 acc[2][2];
    acc_zero(acc);
    // staging roles: A unit (ga, na) for every thread, B unit (gb, kk) for threads 0..255
    const int ga = tid >> 8, na = tid & 255;
    const bool has_b = tid < 256;
    const int gb = (tid >> 7) & 1, kk = tid & 127;
    const int colB = kcol0 + kk;
    const bool okB = has_b && (XF || colB < P.ldx);
    // bias gradient = column sums of dA over the rows r with r % bias_stride == 0 (strides 1, 2 or 4; row groups start at
    // multiples of 8), taken from the staging registers
    const bool do_bias = P.bias_out != nullptr && kb == 0;
    float bsum = 0.f;
    // fragment-ordered operands: the 8 rows of row group g of a stage are the quads (q = 2j + g, hi = 0 / 1) of the tile
    auto frag_base = [&](const float* base, int m) {
        return base + (size_t)(m >> 6) * (64 * 256) + (size_t)((8 * ((m >> 5) & 1) + 2 * ((m >> 4) & 1)) * 256);
    };
    const size_t foffA = (size_t)((((na >> 6) * 16 + ((na >> 5) & 1) * 4 + ga) * 64 + (na & 31)) * 4);
    const int cB = (XF ? colB : 0);
    const size_t foffB = (size_t)((((cB >> 6) * 16 + ((cB >> 5) & 1) * 4 + gb) * 64 + (cB & 31)) * 4);

    auto gload = [&](WxRegs& R, int m) {
        if constexpr (AF) {
            const float* pa = frag_base(P.dA, m) + foffA;
            const v4f t0 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(pa));
            const v4f t1 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(pa + 32 * 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) { R.a[i] = t0[i]; R.a[4 + i] = t1[i]; }
        } else {
            const float* pa = P.dA + (size_t)(m + 8 * ga) * P.lda + na;
#pragma unroll
            for (int r = 0; r < 8; ++r) R.a[r] = __builtin_nontemporal_load(pa + (size_t)r * P.lda);
        }
        if (okB) {
            if constexpr (XF) {
                const float* px = frag_base(P.X, m) + foffB;
                const v4f t0 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(px));
                const v4f t1 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(px + 32 * 4));
#pragma unroll
                for (int i = 0; i < 4; ++i) { R.b[i] = t0[i]; R.b[4 + i] = t1[i]; }
            } else {
                const float* px = P.X + (size_t)(m + 8 * gb) * P.ldx + colB;
#pragma unroll
                for (int r = 0; r < 8; ++r) R.b[r] = __builtin_nontemporal_load(px + (size_t)r * P.ldx);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) R.b[r] = 0.f;
        }
    };
    auto sstore = [&](const WxRegs& R, int buf, bool live = true) {      // live = false: a clamped repeat of the last stage (no bias sums)
        unsigned char* Ab = lds + buf * WX_BUF;
        unsigned char* Bb = Ab + 3 * WX_A_PLANE;
        wx_u32x4 h, m, l;
        wx_split8(R.a, h, m, l);
        const int oa = (ga * 256 + na) * 16;
        *reinterpret_cast<wx_u32x4*>(Ab + oa) = h;
        *reinterpret_cast<wx_u32x4*>(Ab + WX_A_PLANE + oa) = m;
        *reinterpret_cast<wx_u32x4*>(Ab + 2 * WX_A_PLANE + oa) = l;
        if (do_bias && live) {
            if (P.bias_stride == 1) bsum += ((R.a[0] + R.a[1]) + (R.a[2] + R.a[3])) + ((R.a[4] + R.a[5]) + (R.a[6] + R.a[7]));
            else if (P.bias_stride == 2) bsum += (R.a[0] + R.a[2]) + (R.a[4] + R.a[6]);
            else bsum += R.a[0] + R.a[4];
        }
        if (has_b) {
            wx_split8(R.b, h, m, l);
            const int ob = (gb * KW + kk) * 16;
            *reinterpret_cast<wx_u32x4*>(Bb + ob) = h;
            *reinterpret_cast<wx_u32x4*>(Bb + WX_B_PLANE + ob) = m;
            *reinterpret_cast<wx_u32x4*>(Bb + 2 * WX_B_PLANE + ob) = l;
        }
    };
    auto compute = [&](int buf) {
        const unsigned char* Ab = lds + buf * WX_BUF + (hi * 256 + nb * 64 + lo) * 16;
        const unsigned char* Bb = lds + buf * WX_BUF + 3 * WX_A_PLANE + (hi * KW + kh * 64 + lo) * 16;
        wx_u32x4 a[2][3], b[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[t][p] = *reinterpret_cast<const wx_u32x4*>(Ab + p * WX_A_PLANE + t * 32 * 16);
                b[t][p] = *reinterpret_cast<const wx_u32x4*>(Bb + p * WX_B_PLANE + t * 32 * 16);
            }
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};       // smallest partial products first
        // partial product q of all four accumulators before q + 1 (dependent MFMAs on one accumulator 3 issues apart): measured 5 %
        // faster here than six dependent MFMAs per accumulator in a row (the opposite holds in query_x3.hip's loop)
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp)
                    acc[t][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wx_bf16x8, a[t][TA[q]]),
                                                                        __builtin_bit_cast(wx_bf16x8, b[tp][TB[q]]), acc[t][tp], 0, 0, 0);
    };

    // Software pipeline: the kernel is HBM-latency bound (one workgroup per CU), so the loads of stage st + 3 are issued at the
    // start of stage st (four register sets, ~2.5 stages = 60 KB per CU in flight; with two sets the launch sustained 3.4 TB/s =
    // 24 KB per stage time, Little's law) and the stages rotate through four LDS buffers with ONE barrier per stage.
    const int nst = (m1 - m0) / WX_R;           // multiple of 4 (chunks are multiples of 64 rows): the body handles 4 stages
    WxRegs p0, p1, p2, p3;
    gload(p0, m0);
    gload(p1, m0 + WX_R);
    gload(p2, m0 + 2 * WX_R);
    sstore(p0, 0);
    __syncthreads();
#pragma unroll 1
    // Loads are issued UNCONDITIONALLY (clamped to the last stage): behind `if (st + 4 < nst)` the compiler cannot count the outstanding
    // loads and waits for all of them (s_waitcnt vmcnt(0)) before each stage store; [sdf] 1.13 -> 0.95 ms, [deform] 1.06 -> 1.00 ms.
    for (int st = 0; st < nst; st += 4) {
        gload(p3, m0 + WX_R * (st + 3));
        compute(0);
        sstore(p1, 1);
        __syncthreads();
        gload(p0, m0 + WX_R * min(st + 4, nst - 1));
        compute(1);
        sstore(p2, 2);
        __syncthreads();
        gload(p1, m0 + WX_R * min(st + 5, nst - 1));
        compute(2);
        sstore(p3, 3);
        __syncthreads();
        gload(p2, m0 + WX_R * min(st + 6, nst - 1));
        compute(3);
        sstore(p0, 0, st + 4 < nst);
        __syncthreads();
    }
    // acc[t][tp][r]: n = nb*64 + 32 t + (r & 3) + 8 (r >> 2) + 4 hi ;  k = kb*128 + kh*64 + 32 tp + lo
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const int k = kcol0 + kh * 64 + 32 * tp + lo;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb * 64 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if constexpr (DET) det_tile[n * WG_KW + (k - kcol0)] = acc[t][tp][r];
                else if (n < P.N && k < P.K) atomicAdd(P.out + (size_t)n * P.ldo + k, acc[t][tp][r]);
            }
        }
    if (do_bias) {      // the two row groups of a column are summed through LDS (all stages consumed)
        float* red = reinterpret_cast<float*>(lds);
        red[ga * 256 + na] = bsum;
        __syncthreads();
        if (tid < 256) {
            const float sum = red[tid] + red[256 + tid];
            if constexpr (DET) det_bias[tid] = sum;
            else if (tid < P.N) atomicAdd(P.bias_out + tid, sum);
        }
    }
}

template <int NET, bool DET>
__global__ __launch_bounds__(WG_THREADS, 2) void k_wgrad_x3(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wxlds[];
    float* wlds = reinterpret_cast<float*>(wxlds);
    const int task = blockIdx.x;
    auto small_slot = [&](int i) { return DET ? a.det + WG_DET_SMALL_OFF + ((size_t)i * WG_MAX_TASKS + task) * (5 * 256) : nullptr; };
    const bool small_first = task < a.total_tasks / 2;
    if (small_first) {
#pragma unroll 1
        for (int i = 0; i < a.nsmall; ++i) wgrad_small_task<DET>(a.s[i], task, a.total_tasks, wlds, small_slot(i));
        __syncthreads();
    }
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < a.nprob; ++i)
        if (task >= a.p[i].task_begin) pi = i;
    const WgProb& P = a.p[pi];
    const int local = task - P.task_begin;
    const int kblk = (P.K + WG_KW - 1) / WG_KW;
    int kb, mc;
    wg_decode(local, kblk, (P.M + a.MC - 1) / a.MC, kb, mc);
    const int m0 = mc * a.MC, m1 = min(m0 + a.MC, P.M);
    float* dt = DET ? a.det + (size_t)task * WG_DET_TILE : nullptr;
    float* db = DET ? a.det + WG_DET_BIAS_OFF + (size_t)task * 256 : nullptr;
    if constexpr (NET == 1) {
        if (P.a_frag) {
            if (P.x_frag) wgrad_task_x3<true, true, DET>(P, kb, m0, m1, wxlds, dt, db);
            else wgrad_task_x3<true, false, DET>(P, kb, m0, m1, wxlds, dt, db);
        } else {
            if (P.x_frag) wgrad_task_x3<false, true, DET>(P, kb, m0, m1, wxlds, dt, db);
            else wgrad_task_x3<false, false, DET>(P, kb, m0, m1, wxlds, dt, db);
        }
    } else {
        wgrad_task_x3<false, false, DET>(P, kb, m0, m1, wxlds, dt, db);
    }
    if (!small_first) {
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < a.nsmall; ++i) wgrad_small_task<DET>(a.s[i], task, a.total_tasks, wlds, small_slot(i));
    }
}

// Deterministic mode, pass `round`: out += sum of the task slots in ascending row-chunk / task order.  blockIdx.y = problem
// (GEMM problems first, then the small ones); problems of one pass write disjoint outputs.
__global__ __launch_bounds__(256) void k_wgrad_reduce(WgArgs a, int round) {
    const int pi = blockIdx.y;
    const int tid0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    if (pi < a.nprob) {
        const WgProb& P = a.p[pi];
        if (P.round != round) return;
        const int kblk = (P.K + WG_KW - 1) / WG_KW, nchunk = (P.M + a.MC - 1) / a.MC;
        for (int e = tid0; e < P.N * P.K; e += stride) {
            const int n = e / P.K, k = e % P.K, kb = k / WG_KW, kk = k % WG_KW;
            float s = 0.f;
            for (int mc = 0; mc < nchunk; ++mc)
                s += a.det[(size_t)(P.task_begin + wg_encode(kb, mc, kblk, nchunk)) * WG_DET_TILE + n * WG_KW + kk];
            P.out[(size_t)n * P.ldo + k] += s;
        }
        if (P.bias_out)
            for (int n = tid0; n < P.N; n += stride) {
                float s = 0.f;
                for (int mc = 0; mc < nchunk; ++mc)
                    s += a.det[WG_DET_BIAS_OFF + (size_t)(P.task_begin + wg_encode(0, mc, kblk, nchunk)) * 256 + n];
                P.bias_out[n] += s;
            }
    } else {
        const int i = pi - a.nprob;
        const WgSmall& P = a.s[i];
        if (P.round != round) return;
        const float* base = a.det + WG_DET_SMALL_OFF + (size_t)i * WG_MAX_TASKS * (5 * 256);
        for (int e = tid0; e < P.N * 256; e += stride) {
            const int n = e >> 8, k = e & 255;
            float s = 0.f;
            for (int t = 0; t < a.total_tasks; ++t) s += base[(size_t)t * (5 * 256) + n * 256 + k];
            P.out[(size_t)n * P.ldo + k] += s;
        }
        if (P.bias_out)
            for (int n = tid0; n < P.N; n += stride) {
                float s = 0.f;
                for (int t = 0; t < a.total_tasks; ++t) s += base[(size_t)t * (5 * 256) + 4 * 256 + n];
                P.bias_out[n] += s;
            }
    }
}

static int wg_kblk(const WgProb& p) { return (p.K + WG_KW - 1) / WG_KW; }

static int launch_group(WgProb* probs, int nprob, WgSmall* small, int nsmall, int kid_timer, long long rows, float* det, bool x3, hipStream_t st) {
    const int kid = kid_timer == KID_WGRAD_D_X3 ? KID_WGRAD_D : (kid_timer == KID_WGRAD_S_X3 ? KID_WGRAD_S : (kid_timer == KID_WGRAD_C_X3 ? KID_WGRAD_C : kid_timer));
    static DeviceOnce attr_done;
    if (attr_done.first()) {
        if (int e = allow_big_lds(k_wgrad<0, false>, WG_LDS_FLOATS * 4)) return e;
        if (int e = allow_big_lds(k_wgrad<1, false>, WG_LDS_FLOATS * 4)) return e;
        if (int e = allow_big_lds(k_wgrad<2, false>, WG_LDS_FLOATS * 4)) return e;
        if (int e = allow_big_lds(k_wgrad<0, true>, WG_LDS_FLOATS * 4)) return e;
        if (int e = allow_big_lds(k_wgrad<1, true>, WG_LDS_FLOATS * 4)) return e;
        if (int e = allow_big_lds(k_wgrad<2, true>, WG_LDS_FLOATS * 4)) return e;
        if (int e = allow_big_lds(k_wgrad_x3<0, false>, WX_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_wgrad_x3<1, false>, WX_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_wgrad_x3<2, false>, WX_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_wgrad_x3<0, true>, WX_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_wgrad_x3<1, true>, WX_LDS_BYTES)) return e;
        if (int e = allow_big_lds(k_wgrad_x3<2, true>, WX_LDS_BYTES)) return e;
        attr_done.done();
    }
    if (nprob == 0) return ST_OK;
    ES_REQUIRE(nprob <= WG_MAX_PROBS, "too many weight-gradient problems in one group");
    // rows per task: the smallest chunk (multiple of 64 rows) for which the whole group fits in ONE full round of the 512
    // workgroup slots (2 per CU x 256 CUs): long tasks amortise the prologue and the fp32-atomic epilogue (one round
    // measured 1-4.5 % faster than three), and a second, nearly empty round would double the launch
    auto count = [&](int mc) {
        long long t = 0;
        for (int i = 0; i < nprob; ++i) t += (long long)wg_kblk(probs[i]) * ((probs[i].M + mc - 1) / mc);
        return t;
    };
    ES_REQUIRE(nsmall <= WG_MAX_SMALL, "too many small weight-gradient problems in one group");
    int MC = 128;
    const int slots = x3 ? WG_MAX_TASKS / 2 : WG_MAX_TASKS;       // the split-precision kernel runs one workgroup per CU
    while (MC < 65536 && count(MC) > slots) MC += 64;
    ES_REQUIRE(count(MC) <= slots, "weight-gradient group does not fit one round of workgroup slots");
    WgArgs a;
    a.det = det;
    int total = 0, max_round = 0;
    for (int i = 0; i < nprob; ++i) {
        ES_REQUIRE(probs[i].lda == 256 && probs[i].M % 64 == 0, "weight-gradient operands must be [64k][256] adjoints");
        ES_REQUIRE((probs[i].bias_stride & (probs[i].bias_stride - 1)) == 0, "bias stride must be a power of two");
        ES_REQUIRE(!probs[i].a_frag || probs[i].bias_stride == 1 || !probs[i].bias_out, "fragment-ordered adjoints: bias over every row");
        ES_REQUIRE(!probs[i].x_frag || (probs[i].ldx == 256 && probs[i].K == 256), "fragment-ordered inputs are [64 x 256] tiles");
        ES_REQUIRE(kid == KID_WGRAD_S || !(probs[i].x_frag || probs[i].a_frag), "fragment-ordered operands: SDF launch only");
        probs[i].task_begin = total;
        total += wg_kblk(probs[i]) * ((probs[i].M + MC - 1) / MC);
        probs[i].round = 0;
        for (int j = 0; j < i; ++j)
            if (probs[j].out == probs[i].out && probs[j].round >= probs[i].round) probs[i].round = probs[j].round + 1;
        max_round = probs[i].round > max_round ? probs[i].round : max_round;
        a.p[i] = probs[i];
    }
    a.nprob = nprob; a.total_tasks = total; a.MC = MC; a.nsmall = nsmall;
    for (int i = 0; i < nsmall; ++i) {
        ES_REQUIRE(small[i].K == 256 && small[i].N <= 4, "small weight-gradient problems are [<=4 x 256]");
        small[i].round = 0;
        for (int j = 0; j < i; ++j)
            if (small[j].out == small[i].out && small[j].round >= small[i].round) small[i].round = small[j].round + 1;
        max_round = small[i].round > max_round ? small[i].round : max_round;
        a.s[i] = small[i];
    }
    const dim3 grid(total);
    ScopedTimer tm(kid_timer, rows, st);
    if (x3) {
        auto reduce = [&]() {
            for (int r = 0; det && r <= max_round; ++r)
                hipLaunchKernelGGL(k_wgrad_reduce, dim3(32, nprob + nsmall), dim3(256), 0, st, a, r);
        };
        if (det) {
            if (kid == KID_WGRAD_D) hipLaunchKernelGGL((k_wgrad_x3<0, true>), grid, dim3(WG_THREADS), WX_LDS_BYTES, st, a);
            else if (kid == KID_WGRAD_S) hipLaunchKernelGGL((k_wgrad_x3<1, true>), grid, dim3(WG_THREADS), WX_LDS_BYTES, st, a);
            else hipLaunchKernelGGL((k_wgrad_x3<2, true>), grid, dim3(WG_THREADS), WX_LDS_BYTES, st, a);
        } else {
            if (kid == KID_WGRAD_D) hipLaunchKernelGGL((k_wgrad_x3<0, false>), grid, dim3(WG_THREADS), WX_LDS_BYTES, st, a);
            else if (kid == KID_WGRAD_S) hipLaunchKernelGGL((k_wgrad_x3<1, false>), grid, dim3(WG_THREADS), WX_LDS_BYTES, st, a);
            else hipLaunchKernelGGL((k_wgrad_x3<2, false>), grid, dim3(WG_THREADS), WX_LDS_BYTES, st, a);
        }
        reduce();
        return ST_OK;
    }
    if (det) {
        if (kid == KID_WGRAD_D) hipLaunchKernelGGL((k_wgrad<0, true>), grid, dim3(WG_THREADS), WG_LDS_FLOATS * 4, st, a);
        else if (kid == KID_WGRAD_S) hipLaunchKernelGGL((k_wgrad<1, true>), grid, dim3(WG_THREADS), WG_LDS_FLOATS * 4, st, a);
        else hipLaunchKernelGGL((k_wgrad<2, true>), grid, dim3(WG_THREADS), WG_LDS_FLOATS * 4, st, a);
        for (int r = 0; r <= max_round; ++r)
            hipLaunchKernelGGL(k_wgrad_reduce, dim3(32, nprob + nsmall), dim3(256), 0, st, a, r);
        return ST_OK;
    }
    if (kid == KID_WGRAD_D) hipLaunchKernelGGL((k_wgrad<0, false>), grid, dim3(WG_THREADS), WG_LDS_FLOATS * 4, st, a);
    else if (kid == KID_WGRAD_S) hipLaunchKernelGGL((k_wgrad<1, false>), grid, dim3(WG_THREADS), WG_LDS_FLOATS * 4, st, a);
    else hipLaunchKernelGGL((k_wgrad<2, false>), grid, dim3(WG_THREADS), WG_LDS_FLOATS * 4, st, a);
    return ST_OK;
}
size_t wgrad_det_floats() { return WG_DET_FLOATS; }
// One bare GEMM of the weight-gradient kernels: out[n][k] += sum_m dA[m][n] X[m][k] for row-major X [M][256], dA [M][256], out [256][256]
// (M a multiple of 64), on the fp32 matrix pipes (x3 = 0) or in split precision (x3 = 1: both operands split exactly into three bf16
// planes, six partial products, fp32 accumulation).  The accuracy tests drive both with adversarial operands (tests/test_gpu_split_accuracy.py).
int gemm_atb(const float* X, const float* dA, int M, float* out, int x3, float* det, hipStream_t st) {
    WgProb g[1] = {WgProb{X, dA, out, nullptr, 256, 256, 256, M, 256, 256, 1, 0, 0, 0, 0}};
    WgSmall sm[1];
    return launch_group(g, 1, sm, 0, x3 ? KID_WGRAD_D_X3 : KID_WGRAD_D, M, det, x3 != 0, st);
}
// All weight gradients of one point evaluation, accumulated (+=) into dweff (es_weff layout).
// ``det`` (nullable): scratch of wgrad_det_floats() floats => deterministic reduction instead of fp32 atomics.
int point_wgrad(int M, float* ws, int flags, int m_color, const float* d_sdf, float* dweff, float* det, hipStream_t st) {
    if (M <= 0) return ST_OK;
    const WsLayout L = ws_layout(M, flags);
    const Tabs tb = make_tabs();
    const bool x3 = flags & PF_X3;
    const int Mp = L.Mp;
    const int Mc = (flags & PF_COLOR) ? round_up64(m_color > 0 ? m_color : M) : 0;   // rows that went through the colour network
    const size_t t256 = (size_t)Mp * 256;
    auto B = [&](int buf) { return ws + L.off[buf]; };
    auto dW = [&](int net, int l) { return dweff + tb.woff[net * LAYERS + l]; };
    auto dB = [&](int net, int l) { return dweff + tb.boff[net * LAYERS + l]; };
    WgProb g[WG_MAX_PROBS];
    WgSmall sm[WG_MAX_SMALL];
    int n = 0, ns = 0;
    auto small = [&](const float* X, int ldx, const float* dA, int lda, int rows, int K, int N, float* out, int ldo, float* bias, int bstride,
                     int x_frag = 0) {
        sm[ns++] = WgSmall{X, dA, out, bias, ldx, lda, ldo, rows, K, N, bstride, x_frag, 0};
    };
    auto add = [&](const float* X, int ldx, const float* dA, int lda, int rows, int K, int N, float* out, int ldo, float* bias, int bstride,
                   int x_frag = 0, int a_frag = 0) {
        g[n++] = WgProb{X, dA, out, bias, ldx, lda, ldo, rows, K, N, bstride, 0, x_frag, a_frag, 0};
    };
    if (flags & PF_DEFORM) {
        // value + J d rows: (u_l, abar_l) over 2 rows per point (bias gradient from the value rows only); VJP / tangent pair:
        // (tau_l, r_l) over 1 row per point (g_o = J^T g_c is linear in every W_l: dW_l += r_l tau_l^T)
        const int R = 2 * Mp;
        const size_t r256 = (size_t)R * 256;
        n = 0;
        add(B(WS_D_U0), 64, B(WS_D_A), 256, R, 52, 256, dW(NET_D, 0), 52, dB(NET_D, 0), 2);
        add(B(WS_D_T0), 64, B(WS_D_R), 256, Mp, 52, 256, dW(NET_D, 0), 52, nullptr, 1);
        for (int l = 1; l <= 7; ++l) {
            add(B(WS_D_U) + (size_t)(l - 1) * r256, 256, B(WS_D_A) + (size_t)l * r256, 256, R, 256, LAYER_N[NET_D][l], dW(NET_D, l), 256,
                dB(NET_D, l), 2);
            add(B(WS_D_T) + (size_t)(l - 1) * t256, 256, B(WS_D_R) + (size_t)l * t256, 256, Mp, 256, LAYER_N[NET_D][l], dW(NET_D, l), 256,
                nullptr, 1);
        }
        // the deformation launch (the longest) stays a pure GEMM: its last layer's slices ride with the two shorter launches
        if (int e = launch_group(g, n, sm, 0, x3 ? KID_WGRAD_D_X3 : KID_WGRAD_D, M, det, x3, st)) return e;
    }
    {   // SDF: value-pass pairs (s_l, zbar_l) and reverse-pass pairs (tau_l, rho_l); the four [8][Mp][256] stacks of the SDF
        // kernels (s, rho, tau, zbar) are fragment-ordered when the fp32 kernels wrote them (their epilogues load AND store them: one
        // dwordx4 per quad) and row-major when the split-precision family's SDF kernels did (PF_X3_CHAIN | PF_X3_SDF, infer_x3r.hip / train_x3r.hip)
        const int fr = ((flags & PF_X3_CHAIN) && (flags & PF_X3_SDF)) ? 0 : 1;
        n = 0;
        add(B(WS_S_S0), 64, B(WS_S_ZB), 256, Mp, 39, 256, dW(NET_S, 0), 39, dB(NET_S, 0), 1, 0, fr);
        add(B(WS_S_TAU0), 64, B(WS_S_RHO), 256, Mp, 39, 256, dW(NET_S, 0), 39, nullptr, 1, 0, fr);
        for (int l = 1; l <= 7; ++l) {
            const int K = LAYER_K[NET_S][l];
            add(B(WS_S_ACT) + (size_t)(l - 1) * t256, 256, B(WS_S_ZB) + (size_t)l * t256, 256, Mp, 256, 256, dW(NET_S, l), K, dB(NET_S, l), 1, fr, fr);
            add(B(WS_S_TAU) + (size_t)(l - 1) * t256, 256, B(WS_S_RHO) + (size_t)l * t256, 256, Mp, 256, 256, dW(NET_S, l), K, nullptr, 1, fr, fr);
            if (l == 4) {   // skip layer: encoding columns 256..294
                add(B(WS_S_S0), 64, B(WS_S_ZB) + (size_t)4 * t256, 256, Mp, 39, 256, dW(NET_S, 4) + 256, K, nullptr, 1, 0, fr);
                add(B(WS_S_TAU0), 64, B(WS_S_RHO) + (size_t)4 * t256, 256, Mp, 39, 256, dW(NET_S, 4) + 256, K, nullptr, 1, 0, fr);
            }
        }
        if (flags & PF_COLOR)   // feature rows 1..256 of the last layer
            add(B(WS_S_ACT) + (size_t)7 * t256, 256, B(WS_FEATBAR), 256, Mc, 256, 256, dW(NET_S, 8) + 256, 256, dB(NET_S, 8) + 1, 1, fr, 0);
        // row 0 of the last layer: sdfbar^T s_8  +  column sums of tau_8 (adjoint of the reverse sweep's seed row)
        ns = 0;
        if (flags & PF_DEFORM) {     // last deformation layer (3 outputs): value + J d rows here, (tau_8, g_c) pair with the colour launch
            const size_t r256 = (size_t)2 * Mp * 256;
            small(B(WS_D_U) + (size_t)7 * r256, 256, B(WS_D_A8), 4, 2 * Mp, 256, 3, dW(NET_D, 8), 256, dB(NET_D, 8), 2);
            if (!(flags & PF_COLOR)) small(B(WS_D_T) + (size_t)7 * t256, 256, B(WS_GC), 3, Mp, 256, 3, dW(NET_D, 8), 256, nullptr, 1);
        }
        small(B(WS_S_ACT) + (size_t)7 * t256, 256, d_sdf, 1, M, 256, 1, dW(NET_S, 8), 256, dB(NET_S, 8), 1, fr);   // real rows only: d_sdf is [M]
        small(B(WS_S_TAU) + (size_t)7 * t256, 256, nullptr, 1, Mp, 256, 1, dW(NET_S, 8), 256, nullptr, 1, fr);
        if (int e = launch_group(g, n, sm, ns, x3 ? KID_WGRAD_S_X3 : KID_WGRAD_S, M, det, x3, st)) return e;
    }
    if (flags & PF_COLOR) {
        n = 0;
        add(B(WS_C_IN), 128, B(WS_C_Y), 256, Mc, 93, 256, dW(NET_C, 0), 349, dB(NET_C, 0), 1);
        add(B(WS_FEAT), 256, B(WS_C_Y), 256, Mc, 256, 256, dW(NET_C, 0) + 93, 349, nullptr, 1);
        for (int l = 1; l <= 7; ++l) {
            const int K = LAYER_K[NET_C][l];
            add(B(WS_C_H) + (size_t)(l - 1) * t256, 256, B(WS_C_Y) + (size_t)l * t256, 256, Mc, 256, 256, dW(NET_C, l), K, dB(NET_C, l), 1);
            if (l == 4) {
                add(B(WS_C_IN), 128, B(WS_C_Y) + (size_t)4 * t256, 256, Mc, 93, 256, dW(NET_C, 4) + 256, K, nullptr, 1);
                add(B(WS_FEAT), 256, B(WS_C_Y) + (size_t)4 * t256, 256, Mc, 256, 256, dW(NET_C, 4) + 349, K, nullptr, 1);
            }
        }
        ns = 0;
        if (flags & PF_DEFORM) small(B(WS_D_T) + (size_t)7 * t256, 256, B(WS_GC), 3, Mp, 256, 3, dW(NET_D, 8), 256, nullptr, 1);
        small(B(WS_C_H) + (size_t)7 * t256, 256, B(WS_C_Y8), 4, Mc, 256, 3, dW(NET_C, 8), 256, dB(NET_C, 8), 1);
        if (int e = launch_group(g, n, sm, ns, x3 ? KID_WGRAD_C_X3 : KID_WGRAD_C, Mc, det, x3, st)) return e;
    }
    return hip_last("point_wgrad");
}

}  // namespace es
