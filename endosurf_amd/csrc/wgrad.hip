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

#include <type_traits>
#include <utility>

#include "arch.h"
#include "chain_common.h"
#include "launch.h"
#include "tabs.h"
#include "timing.h"
#include "workspace.h"

namespace es {

#ifdef ES_PROFILE_WGRAD       // dev builds only: cycle stamps of block 0 / thread 0 inside the fp32 task (tools/dev/wgrad_profile.py)
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
    int KW;                  // input features per task: WG_KW (fp32 kernel: [256 x 128] tiles) or 256 (split-precision kernel: the whole dW)
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
#ifdef ES_WG_NOPAIR      // dev builds only (A/B of DEAD_ENDS C5): the k-block tasks of a row chunk on NEIGHBOURING block ids (different XCDs)
constexpr bool WG_PAIR = false;
#else
constexpr bool WG_PAIR = true;
#endif
__host__ __device__ inline void wg_decode(int local, int kblk, int nchunk, int& kb, int& mc) {
    if (WG_PAIR && kblk == 2) {
        const int grp = local / 16, j = local % 16;
        const int full = (nchunk / 8) * 8;                 // chunks covered by complete groups of 8
        if (grp * 8 < full) { mc = grp * 8 + (j & 7); kb = j >> 3; }
        else { const int rem = local - 2 * full; kb = rem & 1; mc = full + (rem >> 1); }
    } else { kb = local % kblk; mc = local / kblk; }
}
__host__ __device__ inline int wg_encode(int kb, int mc, int kblk, int nchunk) {
    if (WG_PAIR && kblk == 2) {
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
    // (every thread sums the rows it stages; the row mask of a strided problem is applied once, in the epilogue)
    const bool do_bias = P.bias_out != nullptr && kb == 0;
    v4f_t bsum = {0.f, 0.f, 0.f, 0.f};

    // global -> register loads of one 16-row stage (two stages in flight: sets 0/1), register -> LDS stores.
    // Plain ext-vector locals (not HIP float4 structs) so that they stay in VGPRs.
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f vzero = {0.f, 0.f, 0.f, 0.f};
    const int fa1 = tid + WG_THREADS;
    // row-major operands: thread -> (row, 4 columns); offsets relative to the stage's first row
    // (32-bit lane offsets, applied as BYTE offsets to the stage's wave-uniform base pointer -- a stage spans 16 rows -- so that every
    // load is "scalar base + 32-bit lane offset": as size_t element offsets they were 64-bit pairs, one of them reloaded from scratch in
    // every stage of the loop at the 128-register cap)
    const unsigned offA0 = (unsigned)(tid >> 6) * (unsigned)P.lda + 4u * (tid & 63);
    const int colB = kcol0 + 4 * (tid % (KW / 4));
    const unsigned offB = (unsigned)(tid / (KW / 4)) * (unsigned)P.ldx + (unsigned)colB;
    const bool okB = XF || colB < P.ldx;
    // fragment-ordered operands: unit u -> (wave block u>>8, ni, quad parity qq, lane) of the stage
    const int fqq = (tid >> 6) & 1, fni = (tid >> 7) & 1, fw = tid >> 8;                // fw in 0..1 (unit tid), +2 for unit tid+512
    const unsigned foffA0 = (unsigned)(((fw * 16 + fni * 4 + fqq) * 64 + lane) * 4);
    const unsigned foffB = (unsigned)((((2 * kb + fw) * 16 + fni * 4 + fqq) * 64 + lane) * 4);
    const int flrow = 8 * fqq + 4 * hi;                                                // first of the 4 local rows of the unit
    const int flcolA = 64 * fw + 32 * fni + lo, flcolB = 64 * fw + 32 * fni + lo;      // A: + 128 for the second unit
    // the three byte offsets a thread applies to the stage's base pointers: laundered through an empty asm at every use, otherwise their
    // zero-extensions are hoisted out of the stage loop as 64-bit pairs and the loads lose the scalar-base addressing form
    const unsigned boffA0 = 4u * (AF ? foffA0 : offA0), boffB = 4u * (XF ? foffB : offB);
    const size_t dA1 = AF ? (size_t)2 * 16 * 64 * 4 * 4 : (size_t)8 * P.lda * 4;      // bytes: unit tid + 512 = 8 rows (row-major) / 2 x 16 units (fragment order) further
    auto stage_ptr = [&](const float* base, int ld, bool frag, int m) {
        return frag ? base + (size_t)(m >> 6) * (64 * 256) + (size_t)((8 * ((m >> 5) & 1) + 2 * ((m >> 4) & 1)) * 256) : base + (size_t)m * ld;
    };
#define WG_GLOAD(S, m)                                                                                \
    {                                                                                                 \
        const char* pa = reinterpret_cast<const char*>(stage_ptr(P.dA, P.lda, AF, m));                \
        unsigned oa0 = boffA0, ob = boffB;                                                            \
        asm volatile("" : "+v"(oa0), "+v"(ob));      /* keeps the zero-extension next to the load: see boffA0 */ \
        S##a0 = *reinterpret_cast<const v4f*>(pa + oa0);                                              \
        S##a1 = *reinterpret_cast<const v4f*>(pa + dA1 + oa0);      /* second unit: a wave-uniform distance away */ \
        const char* px = reinterpret_cast<const char*>(stage_ptr(P.X, P.ldx, XF, m));                 \
        S##b0 = okB ? *reinterpret_cast<const v4f*>(px + ob) : vzero;                                 \
    }
#define WG_SSTORE(S, buf)                                                                             \
    {                                                                                                 \
        if constexpr (AF) {                                                                           \
            float* la = Apan(buf) + flrow * 256 + flcolA;                                             \
            la[0] = S##a0[0]; la[256] = S##a0[1]; la[512] = S##a0[2]; la[768] = S##a0[3];             \
            la[128] = S##a1[0]; la[128 + 256] = S##a1[1]; la[128 + 512] = S##a1[2]; la[128 + 768] = S##a1[3]; \
            bsum[0] += S##a0[0] + S##a0[1] + S##a0[2] + S##a0[3];                                     \
            bsum[1] += S##a1[0] + S##a1[1] + S##a1[2] + S##a1[3];                                     \
        } else {                                                                                      \
            *reinterpret_cast<v4f*>(Apan(buf) + 4 * tid) = S##a0;                                     \
            *reinterpret_cast<v4f*>(Apan(buf) + 4 * fa1) = S##a1;                                     \
            bsum += S##a0 + S##a1;                                                                    \
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
    // The epilogue's lane indices are RE-DERIVED from the thread id (laundered through an empty asm so that they are new values): kept
    // live across the stage loop they were what the allocator spilled at the 128-register cap of four waves per SIMD (round 4).
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, lo_e = lane_e & 31, hi_e = lane_e >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const int k = kcol0 + kh * 64 + 2 * lo_e + tp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * hi_e) + t;
                if constexpr (DET) det_tile[n * WG_KW + (k - kcol0)] = acc[t][tp][r];
                else if (n < P.N && k < P.K) atomicAdd(P.out + (size_t)n * P.ldo + k, acc[t][tp][r]);
            }
        }
    if (do_bias) {      // workgroup-uniform: reduce the partial column sums through LDS (all stages consumed)
        if (!AF && ((tid_e >> 6) & (P.bias_stride - 1)) != 0) bsum = v4f{0.f, 0.f, 0.f, 0.f};       // AF: stride 1 only
        __syncthreads();
        if constexpr (AF) {         // 4 threads (qq, hi) per column; unit tid holds column flcolA, unit tid+512 column flcolA + 128
            const int fqq_e = (tid_e >> 6) & 1, col_e = 64 * (tid_e >> 8) + 32 * ((tid_e >> 7) & 1) + lo_e;
            lds[(fqq_e * 2 + hi_e) * 256 + col_e] = bsum[0];
            lds[(fqq_e * 2 + hi_e) * 256 + col_e + 128] = bsum[1];
        } else {
            *reinterpret_cast<v4f*>(lds + 4 * tid_e) = bsum;
        }
        __syncthreads();
        if (tid_e < 256) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < (AF ? 4 : 8); ++r) s += lds[r * 256 + tid_e];
            if constexpr (DET) det_bias[tid_e] = s;
            else if (tid_e < P.N) atomicAdd(P.bias_out + tid_e, s);
        }
    }
    W_STAMP(9);
}

// Slice `slot` of `nslots` of a small problem: thread = (input feature k, row-block parity); 16-row blocks, the loads of two
// steps (2 x 16 rows of X per thread) in flight, the <= 4 adjoint columns go through LDS (double buffered, one barrier per step).
constexpr int WS_ROWS = 16;
// NH: 256-thread halves of the workgroup (2: the fp32 kernel's 512 threads, 1: the split-precision kernel's 256); every half streams its own row blocks
template <bool DET, int NH = 2>
__device__ __forceinline__ void wgrad_small_task(const WgSmall& P, int slot, int nslots, float* lds, float* det_slot) {
    float(*sd)[WS_ROWS][4] = reinterpret_cast<float(*)[WS_ROWS][4]>(lds);       // [half * 2 + buffer]
    const int tid = threadIdx.x, k = tid & 255, half = NH == 2 ? tid >> 8 : 0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const int nblk = (P.M + WS_ROWS - 1) / WS_ROWS;
    const int step = NH * nslots;
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
    int b0 = NH * slot, it = 0;
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
    if (NH == 2 && half) {
#pragma unroll
        for (int n = 0; n < 4; ++n) red[n * 256 + k] = acc[n];
        red[1024 + k] = bsum;
    }
    if (NH == 2) __syncthreads();
    if (!half) {
        if constexpr (NH == 2) {
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] += red[n * 256 + k];
            bsum += red[1024 + k];
        }
        if constexpr (DET) {        // det_slot: [5][256] = 4 output rows + bias row of this (small problem, task)
            for (int n = 0; n < 4; ++n) det_slot[n * 256 + k] = acc[n];
            det_slot[4 * 256 + k] = bsum;
        } else {
            for (int n = 0; n < P.N; ++n) atomicAdd(P.out + (size_t)n * P.ldo + k, acc[n]);
            if (P.bias_out && k < P.N) atomicAdd(P.bias_out + k, bsum);
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
// OPT-IN split-precision variant (flag PF_X3; csrc/query_x3.hip explains the arithmetic).  Both operand panels are split EXACTLY into
// three bf16 planes on their way into LDS and the contraction runs on v_mfma_f32_32x32x16_bf16 (six partial products per tile, fp32
// accumulation): ~2.7x the fp32 matrix rate, which turns these GEMMs from MFMA-bound into HBM-bound -- so a task here is the WHOLE
// [256 x 256] dW of a layer over a chunk of rows: every dA / X element is read from HBM exactly once per launch (with [256 x 128]
// tiles the dA rows were fetched by two tasks: 4.67 GB fetched for 3.4 GB of operands, at 4.3 TB/s).
// One workgroup = FOUR waves, one per SIMD, each with the SIMD's whole register file (256 accumulators = a [128 x 128] tile of dW, 4 x 4
// MFMA tiles): with one wave per SIMD nothing overlaps by itself, so the stage (16 rows = one MFMA k-step) is software-pipelined by hand:
//   * panels are written TWO stages ahead (stage s splits and writes the rows of stage s + 2 into buffer (s + 2) % 3, last read in
//     stage s - 1's prefetches), so the fragments of stage s + 1 are already visible during stage s and are fetched between its MFMAs;
//   * the X fragments are processed two 32-column tiles at a time (B0 / B1: the other pair is fetched meanwhile); the dA fragments (4
//     tiles x 3 planes) are replaced plane by plane as they die, which fixes the order of the six partial products per half (always
//     smallest class first: 2^-16, 2^-16, 2^-16, 2^-8, 2^-8, 1);
//   * two register sets of 32 staged floats per thread (the loads of stage s + 4 are issued when stage s has split the rows of s + 2);
//   * one MFMA per "slot" (96 per stage), the side work (ds_read / split / ds_write / global loads) distributed over the slots and
//     pinned with sched_barrier; the stage body is branch-free (rows past the chunk's end are staged as zeros through selects).
// Problems with few input features (first layers, skip columns: K = 39 / 52 / 93) take the same path (their missing columns are
// whatever the clamped loads return; those output columns are never stored).
// Measured (tools/dev/wgrad_x3_probe.py: one [1.65 M x 256] x [1.65 M x 256] problem = the deformation launch's rows; tools/dev/pmc_probe.sh;
// -DES_WX_NO_MFMA / -DES_WX_NO_SPLIT builds): gaussian operands 1.31 ms at 1.56 GHz (MFMA busy 0.61); MFMAs + fragment reads alone
// 0.78 ms at 1.80 GHz (busy 0.895); loads + split + panel writes alone 0.60 ms at 1.99 GHz (5.7 TB/s); all-zero operands 0.92 / 0.62 /
// 0.53 ms -- the clock follows the power drawn, and the parts ADD: a stage costs 3 465 cycles of MFMA stream + ~1 585 for its 257 VALU and
// 44 memory instructions, interleaved or not.  Two waves per SIMD (8 waves of [64 x 128], stage phases of SIMD partners in lock step or
// offset by half a stage) measured the same 0.95-1.0 ms per deformation launch of the training step; what moved it was the operand
// traffic ([256 x 128] tiles: 1.03 ms).
// LDS per plane: units of 16 B = 8 consecutive ROWS of one column, [row group (2)][column] -> the A / B fragments (lane = column,
// 8 rows) are single conflict-free ds_read_b128.  A thread stages column tid of both row groups of both operands: 16 dword loads per
// operand (row-major, coalesced over columns) or 4 float4 loads (fragment-ordered).
typedef unsigned wx_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wx_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int WX_THREADS = 256;
constexpr int WX_R = 16;
constexpr int WX_KW = 256;                                   // input features per task: all of them
constexpr int WX_PLANE = 2 * 256 * 16;                       // 8 KiB
constexpr int WX_BUF = 6 * WX_PLANE;                         // 48 KiB per stage buffer: 3 planes of dA, 3 of X
constexpr int WX_NBUF = 3;
constexpr int WX_LDS_BYTES = WX_NBUF * WX_BUF;               // 144 KiB, one workgroup per CU
static_assert(WX_LDS_BYTES >= WG_LDS_FLOATS * 4, "the small-layer slices reuse the buffer as float scratch");

__device__ __forceinline__ unsigned wx_cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
template <int... Is, class F>
__device__ __forceinline__ void wx_static_for(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>()), ...);
}

template <bool AF, bool XF, bool DET>
__device__ __forceinline__ void wgrad_task_x3(const WgProb& P, int m0, int m1, unsigned char* lds, float* det_tile, float* det_bias) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 1, wk = w >> 1;
    const int lo = lane & 31, hi = lane >> 5;
    typedef float v4f __attribute__((ext_vector_type(4)));

    f32x16 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int tp = 0; tp < 4; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][tp][r] = 0.f;
    const int c = tid;                                       // staging column of this thread (both row groups, both operands)
    const int bstride = P.bias_stride;
    float bsum = 0.f;                                        // bias gradient = column sum of dA over the rows r with r % bias_stride == 0
    const unsigned wr_off = (unsigned)c * 16;
    const unsigned ra_off = (unsigned)(hi * 256 + wn * 128 + lo) * 16;
    const unsigned rb_off = 3 * WX_PLANE + (unsigned)(hi * 256 + wk * 128 + lo) * 16;
    const int nst = (m1 - m0) / WX_R;
    // operand addressing: 32-bit element offsets from the (uniform) base pointers; dA is a [rows][256] stack (launch_group checks)
    const float* __restrict__ gA = P.dA;
    const float* __restrict__ gX = P.X;
    constexpr unsigned lda = 256;
    const unsigned ldx = (unsigned)P.ldx;
    const unsigned cx = min((unsigned)c, ldx - 1);           // X stacks narrower than 256 columns: clamped (those outputs are not stored)
    const unsigned foff = (unsigned)((((c >> 6) * 16 + ((c >> 5) & 1) * 4) * 64 + (c & 31)) * 4);
    // fragment-ordered stacks (chain_common.h frag_off): rows 8 g .. 8 g + 7 of the stage = quad 2 j + g of the tile, lane halves 0 / 1
    auto frag_off = [&](unsigned m) { return (m >> 6) * (64u * 256u) + (8u * ((m >> 5) & 1u) + 2u * ((m >> 4) & 1u)) * 256u + foff; };
    auto stage_row = [&](int st) { return (unsigned)(m0 + WX_R * min(st, nst - 1)); };

    // one of the 16 loads of a stage (row-major: row r; fragment-ordered: the float4 of rows r .. r + 3 when r % 4 == 0)
    auto loadA = [&](float (&v)[16], int st, int r) {
        const unsigned m = stage_row(st);
        if constexpr (AF) {
            if (r % 4 == 0) {
                const float* pa = gA + frag_off(m) + (r >> 3) * 256 + ((r >> 2) & 1) * 128;
                const v4f t0 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(pa));
#pragma unroll
                for (int i = 0; i < 4; ++i) v[r + i] = t0[i];
            }
        } else {
            v[r] = __builtin_nontemporal_load(gA + (m * lda + c) + r * lda);
        }
    };
    auto loadB = [&](float (&v)[16], int st, int r) {
        const unsigned m = stage_row(st);
        if constexpr (XF) {
            if (r % 4 == 0) {
                const float* px = gX + frag_off(m) + (r >> 3) * 256 + ((r >> 2) & 1) * 128;
                const v4f t0 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(px));
#pragma unroll
                for (int i = 0; i < 4; ++i) v[r + i] = t0[i];
            }
        } else {
            v[r] = __builtin_nontemporal_load(gX + (m * ldx + cx) + r * ldx);
        }
    };
    // two floats -> word j of the three planes (v = h + m + l exactly)
    auto split2 = [&](float x0, float x1, bool live, wx_u32x4& h, wx_u32x4& m, wx_u32x4& l, int j) {
        x0 = live ? x0 : 0.f; x1 = live ? x1 : 0.f;
        const unsigned hh = wx_cvt_pk(x0, x1);
        float r0 = x0 - __uint_as_float(hh << 16), r1 = x1 - __uint_as_float(hh & 0xffff0000u);
        const unsigned mm = wx_cvt_pk(r0, r1);
        r0 -= __uint_as_float(mm << 16); r1 -= __uint_as_float(mm & 0xffff0000u);
        h[j] = hh; m[j] = mm; l[j] = wx_cvt_pk(r0, r1);
    };
    auto bias_add = [&](const float (&v)[16], bool live) {
        const float s4 = (v[0] + v[4]) + (v[8] + v[12]);
        const float s2 = s4 + ((v[2] + v[6]) + (v[10] + v[14]));
        const float s1 = s2 + (((v[1] + v[3]) + (v[5] + v[7])) + ((v[9] + v[11]) + (v[13] + v[15])));
        const float sel = bstride == 1 ? s1 : (bstride == 2 ? s2 : s4);
        bsum += live ? sel : 0.f;
    };
    auto wr3 = [&](int buf, int operand, int g, const wx_u32x4& h, const wx_u32x4& m, const wx_u32x4& l) {
        unsigned char* p = lds + buf * WX_BUF + operand * 3 * WX_PLANE + g * 4096 + wr_off;
        *reinterpret_cast<wx_u32x4*>(p) = h;
        *reinterpret_cast<wx_u32x4*>(p + WX_PLANE) = m;
        *reinterpret_cast<wx_u32x4*>(p + 2 * WX_PLANE) = l;
    };
    wx_u32x4 A[4][3], B0[2][3], B1[2][3];
    auto rdA = [&](int buf, int t, int p) { A[t][p] = *reinterpret_cast<const wx_u32x4*>(lds + buf * WX_BUF + ra_off + p * WX_PLANE + t * 512); };
    auto rdB0 = [&](int buf, int j, int p) { B0[j][p] = *reinterpret_cast<const wx_u32x4*>(lds + buf * WX_BUF + rb_off + p * WX_PLANE + j * 512); };
    auto rdB1 = [&](int buf, int j, int p) { B1[j][p] = *reinterpret_cast<const wx_u32x4*>(lds + buf * WX_BUF + rb_off + p * WX_PLANE + (2 + j) * 512); };
    auto bar = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    float pa0[16], pb0[16], pa1[16], pb1[16];        // register sets 0 / 1
    // One stage: MFMAs on the fragments of buffer CUR; fetches the rest of CUR's and the first fragments of NXT; splits the set
    // (rows of stage s + 2; `live` = they exist) into buffer WRT and reloads it with the rows of stage `reload`.
    auto stage = [&](auto cur_c, auto nxt_c, auto wrt_c, float (&sa)[16], float (&sb)[16], bool live, int reload) {
        constexpr int CUR = decltype(cur_c)::value, NXT = decltype(nxt_c)::value, WRT = decltype(wrt_c)::value;
        constexpr int TA0[6] = {1, 2, 0, 1, 0, 0}, TB0[6] = {1, 0, 2, 0, 1, 0};      // first half:  (a1 b1) (a2 b0) (a0 b2) | (a1 b0) (a0 b1) | (a0 b0)
        constexpr int TA1[6] = {0, 1, 2, 0, 1, 0}, TB1[6] = {2, 1, 0, 1, 0, 0};      // second half: (a0 b2) (a1 b1) (a2 b0) | (a0 b1) (a1 b0) | (a0 b0)
        wx_u32x4 h, m, l;
        // dA plane 0 died with the last MFMA of the previous stage (needed from slot 16 on)
        rdA(CUR, 0, 0); rdA(CUR, 1, 0); rdA(CUR, 2, 0); rdA(CUR, 3, 0);
        wx_static_for(std::make_integer_sequence<int, 96>(), [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int half = i / 48, q = (i % 48) / 8, t = (i % 8) / 2, j = i % 2;
#ifndef ES_WX_NO_MFMA
            if constexpr (half == 0)
                acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wx_bf16x8, A[t][TA0[q]]), __builtin_bit_cast(wx_bf16x8, B0[j][TB0[q]]),
                                                                    acc[t][j], 0, 0, 0);
            else
                acc[t][2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wx_bf16x8, A[t][TA1[q]]), __builtin_bit_cast(wx_bf16x8, B1[j][TB1[q]]),
                                                                        acc[t][2 + j], 0, 0, 0);
#endif
            // ---- side work of the slot ----
            if constexpr (i == 7) bias_add(sa, live);
            if constexpr (i >= 2 && i < 8) rdB1(CUR, (i - 2) & 1, 2 - (i - 2) / 2);              // X tiles 2, 3 of this stage: planes 2, 1, 0
#ifndef ES_WX_NO_SPLIT
            if constexpr (i >= 8 && i < 56 && (i - 8) % 3 == 0) {                                 // 16 pair splits: operand, row group, word
                constexpr int k = (i - 8) / 3, g = (k / 4) % 2, jj = k % 4;
                if constexpr (k < 8) split2(sa[8 * g + 2 * jj], sa[8 * g + 2 * jj + 1], live, h, m, l, jj);
                else split2(sb[8 * g + 2 * jj], sb[8 * g + 2 * jj + 1], true, h, m, l, jj);     // dead rows: dA is zero, X is finite (clamped loads)
            }
            if constexpr (i == 18) wr3(WRT, 0, 0, h, m, l);
            if constexpr (i == 30) wr3(WRT, 0, 1, h, m, l);
#ifndef ES_WX_NO_LOAD
            if constexpr (i >= 32 && i < 48) loadA(sa, reload, i - 32);
#endif
            if constexpr (i == 42) wr3(WRT, 1, 0, h, m, l);
            if constexpr (i == 54) wr3(WRT, 1, 1, h, m, l);
#endif
#ifndef ES_WX_NO_LOAD
            if constexpr (i >= 56 && i < 72) loadB(sb, reload, i - 56);
#endif
            if constexpr (i >= 60 && i < 66) rdB0(NXT, (i - 60) & 1, (i - 60) / 2);               // next stage, X tiles 0, 1
            if constexpr (i >= 72 && i < 76) rdA(NXT, i - 72, 2);                                 // dA plane 2 died with slot 71
            if constexpr (i >= 88 && i < 92) rdA(NXT, i - 88, 1);                                 // dA plane 1 died with slot 87
            __builtin_amdgcn_sched_barrier(0);
        });
        bar();
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;

    // prologue: rows of stages 0 / 1 into buffers 0 / 1, sets reloaded with stages 2 / 3, first fragments of buffer 0
#pragma unroll
    for (int r = 0; r < 16; ++r) { loadA(pa0, 0, r); loadB(pb0, 0, r); }
#pragma unroll
    for (int r = 0; r < 16; ++r) { loadA(pa1, 1, r); loadB(pb1, 1, r); }
    {
        wx_u32x4 h, m, l;
        auto stage_set = [&](int buf, int operand, const float (&v)[16], bool live) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) split2(v[8 * g + 2 * jj], v[8 * g + 2 * jj + 1], live, h, m, l, jj);
                wr3(buf, operand, g, h, m, l);
            }
        };
        stage_set(0, 0, pa0, true); bias_add(pa0, true);
        stage_set(0, 1, pb0, true);
#pragma unroll
        for (int r = 0; r < 16; ++r) { loadA(pa0, 2, r); loadB(pb0, 2, r); }
        stage_set(1, 0, pa1, 1 < nst); bias_add(pa1, 1 < nst);
        stage_set(1, 1, pb1, true);
#pragma unroll
        for (int r = 0; r < 16; ++r) { loadA(pa1, 3, r); loadB(pb1, 3, r); }
    }
    bar();
#pragma unroll
    for (int t = 0; t < 4; ++t) { rdA(0, t, 1); rdA(0, t, 2); }
#pragma unroll
    for (int p = 0; p < 3; ++p) { rdB0(0, 0, p); rdB0(0, 1, p); }
#pragma unroll 1
    for (int st = 0; st < nst; st += 6) {       // stages past the end of the chunk multiply zero panels
        stage(I0(), I1(), I2(), pa0, pb0, st + 2 < nst, st + 4);
        stage(I1(), I2(), I0(), pa1, pb1, st + 3 < nst, st + 5);
        stage(I2(), I0(), I1(), pa0, pb0, st + 4 < nst, st + 6);
        stage(I0(), I1(), I2(), pa1, pb1, st + 5 < nst, st + 7);
        stage(I1(), I2(), I0(), pa0, pb0, st + 6 < nst, st + 8);
        stage(I2(), I0(), I1(), pa1, pb1, st + 7 < nst, st + 9);
    }
    // acc[t][tp][r]: n = wn*128 + 32 t + (r & 3) + 8 (r >> 2) + 4 hi ;  k = wk*128 + 32 tp + lo
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const int k = wk * 128 + 32 * tp + lo;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = wn * 128 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if constexpr (DET) det_tile[n * WX_KW + k] = acc[t][tp][r];
                else if (n < P.N && k < P.K) atomicAdd(P.out + (size_t)n * P.ldo + k, acc[t][tp][r]);
            }
        }
    if (P.bias_out != nullptr) {
        if constexpr (DET) det_bias[tid] = bsum;
        else if (tid < P.N) atomicAdd(P.bias_out + tid, bsum);
    }
    __syncthreads();        // the next user of the LDS buffer (small-layer slices) must not overtake this task's last fragment reads
}

template <int NET, bool DET>
__global__ __launch_bounds__(WX_THREADS, 1) void k_wgrad_x3(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wxlds[];
    float* wlds = reinterpret_cast<float*>(wxlds);
    const int task = blockIdx.x;
    auto small_slot = [&](int i) { return DET ? a.det + WG_DET_SMALL_OFF + ((size_t)i * WG_MAX_TASKS + task) * (5 * 256) : nullptr; };
    const bool small_first = task < a.total_tasks / 2;
    if (small_first) {
#pragma unroll 1
        for (int i = 0; i < a.nsmall; ++i) wgrad_small_task<DET, 1>(a.s[i], task, a.total_tasks, wlds, small_slot(i));
        __syncthreads();
    }
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < a.nprob; ++i)
        if (task >= a.p[i].task_begin) pi = i;
    const WgProb& P = a.p[pi];
    const int mc = task - P.task_begin;           // one task per row chunk: the whole [256 x 256] dW
    const int m0 = mc * a.MC, m1 = min(m0 + a.MC, P.M);
    float* dt = DET ? a.det + (size_t)task * (256 * WX_KW) : nullptr;
    float* db = DET ? a.det + WG_DET_BIAS_OFF + (size_t)task * 256 : nullptr;
    if constexpr (NET == 1) {      // only the SDF network's stacks can be fragment-ordered
        if (P.a_frag) {
            if (P.x_frag) wgrad_task_x3<true, true, DET>(P, m0, m1, wxlds, dt, db);
            else wgrad_task_x3<true, false, DET>(P, m0, m1, wxlds, dt, db);
        } else {
            if (P.x_frag) wgrad_task_x3<false, true, DET>(P, m0, m1, wxlds, dt, db);
            else wgrad_task_x3<false, false, DET>(P, m0, m1, wxlds, dt, db);
        }
    } else {
        wgrad_task_x3<false, false, DET>(P, m0, m1, wxlds, dt, db);
    }
    if (!small_first) {
#pragma unroll 1
        for (int i = 0; i < a.nsmall; ++i) wgrad_small_task<DET, 1>(a.s[i], task, a.total_tasks, wlds, small_slot(i));
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
        const int kblk = (P.K + a.KW - 1) / a.KW, nchunk = (P.M + a.MC - 1) / a.MC;
        const size_t tile = (size_t)256 * a.KW;
        for (int e = tid0; e < P.N * P.K; e += stride) {
            const int n = e / P.K, k = e % P.K, kb = k / a.KW, kk = k % a.KW;
            float s = 0.f;
            for (int mc = 0; mc < nchunk; ++mc)
                s += a.det[(size_t)(P.task_begin + wg_encode(kb, mc, kblk, nchunk)) * tile + n * a.KW + kk];
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

static int wg_kblk(const WgProb& p, int kw) { return (p.K + kw - 1) / kw; }

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
    const int KW = x3 ? WX_KW : WG_KW;
    auto count = [&](int mc) {
        long long t = 0;
        for (int i = 0; i < nprob; ++i) t += (long long)wg_kblk(probs[i], KW) * ((probs[i].M + mc - 1) / mc);
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
        total += wg_kblk(probs[i], KW) * ((probs[i].M + MC - 1) / MC);
        probs[i].round = 0;
        for (int j = 0; j < i; ++j)
            if (probs[j].out == probs[i].out && probs[j].round >= probs[i].round) probs[i].round = probs[j].round + 1;
        max_round = probs[i].round > max_round ? probs[i].round : max_round;
        a.p[i] = probs[i];
    }
    a.nprob = nprob; a.total_tasks = total; a.MC = MC; a.nsmall = nsmall; a.KW = KW;
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
            if (kid == KID_WGRAD_D) hipLaunchKernelGGL((k_wgrad_x3<0, true>), grid, dim3(WX_THREADS), WX_LDS_BYTES, st, a);
            else if (kid == KID_WGRAD_S) hipLaunchKernelGGL((k_wgrad_x3<1, true>), grid, dim3(WX_THREADS), WX_LDS_BYTES, st, a);
            else hipLaunchKernelGGL((k_wgrad_x3<2, true>), grid, dim3(WX_THREADS), WX_LDS_BYTES, st, a);
        } else {
            if (kid == KID_WGRAD_D) hipLaunchKernelGGL((k_wgrad_x3<0, false>), grid, dim3(WX_THREADS), WX_LDS_BYTES, st, a);
            else if (kid == KID_WGRAD_S) hipLaunchKernelGGL((k_wgrad_x3<1, false>), grid, dim3(WX_THREADS), WX_LDS_BYTES, st, a);
            else hipLaunchKernelGGL((k_wgrad_x3<2, false>), grid, dim3(WX_THREADS), WX_LDS_BYTES, st, a);
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
// net_mask: bit n = launch network n's group (deform, sdf, colour); 7 = all (es_point_backward).  The groups are independent launches that
// accumulate into disjoint parts of dweff -- except the deformation network's LAST layer, whose slices ride in the sdf and colour launches.
int point_wgrad(int M, float* ws, int flags, int m_color, const float* d_sdf, float* dweff, float* det, hipStream_t st, int net_mask) {
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
    if ((flags & PF_DEFORM) && (net_mask & 1)) {
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
    if (net_mask & 2) {   // SDF: value-pass pairs (s_l, zbar_l) and reverse-pass pairs (tau_l, rho_l); the four [8][Mp][256] stacks of the SDF
        // kernels (s, rho, tau, zbar) are fragment-ordered: the fp32 SDF kernels, which write them in both kernel families, load AND
        // store them in their epilogues (one dwordx4 per quad); the operand-layout flag stays a per-problem property
        const int fr = 1;
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
    if ((flags & PF_COLOR) && (net_mask & 4)) {
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
