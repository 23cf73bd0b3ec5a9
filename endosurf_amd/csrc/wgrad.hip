// Weight-gradient GEMMs of the backward pass:  dW[n][k] += sum_m dA[m][n] * X[m][k]  for every layer, from the
// (layer input X, pre-activation adjoint dA) pairs streamed to the workspace by the forward/backward chains.
// All problems of one network are batched in ONE launch (grouped GEMM): independent wavefront tasks
// (problem, 64x64 output tile, row chunk), operands loaded global->VGPR as float2 (two MFMA fragments per load,
// feature-interleaved), contraction on v_mfma_f32_32x32x2_f32, results reduced with fp32 atomics (few row chunks per
// output tile because the layer dimension supplies the parallelism).  Bias gradients are column sums of dA taken on
// the fly by the k-block-0 tasks.
#include <hip/hip_runtime.h>

#include "arch.h"
#include "chain_common.h"
#include "launch.h"
#include "tabs.h"
#include "timing.h"
#include "workspace.h"

namespace es {

constexpr int WG_MAX_PROBS = 20;

struct WgProb {
    const float* X; const float* dA; float* out; float* bias_out;
    int ldx, lda, ldo, M, K, N, bias_stride, task_begin;
};
struct WgArgs {
    WgProb p[WG_MAX_PROBS];
    int nprob, total_tasks, MC;
};

__global__ __launch_bounds__(256) void k_wgrad(WgArgs a) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (task >= a.total_tasks) return;
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < a.nprob; ++i)
        if (task >= a.p[i].task_begin) pi = i;
    const WgProb& P = a.p[pi];
    const int kblk = (P.K + 63) / 64, nblk = (P.N + 63) / 64;
    const int local = task - P.task_begin;
    const int kb = local % kblk, nb = (local / kblk) % nblk, mc = local / (kblk * nblk);
    const int m0 = mc * a.MC, m1 = min(m0 + a.MC, P.M);
    const int lo = lane & 31, hi = lane >> 5;
    const float* Ap = P.dA + (size_t)(m0 + hi) * P.lda + nb * 64 + 2 * lo;
    const float* Bp = P.X + (size_t)(m0 + hi) * P.ldx + kb * 64 + 2 * lo;
    const bool do_bias = P.bias_out != nullptr && kb == 0;
    const int bstride = P.bias_stride;

    f32x16 acc[2][2];
    acc_zero(acc);
    float bs0 = 0.f, bs1 = 0.f;
    constexpr int U = 8;   // k-steps (pairs of rows) per register set
    float2 a0[U], b0[U], a1[U], b1[U];
    auto load = [&](float2(&av)[U], float2(&bv)[U], int m) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (m + 2 * u < m1) {
                av[u] = *reinterpret_cast<const float2*>(Ap + (size_t)(m - m0 + 2 * u) * P.lda);
                bv[u] = *reinterpret_cast<const float2*>(Bp + (size_t)(m - m0 + 2 * u) * P.ldx);
            } else {
                av[u] = make_float2(0.f, 0.f); bv[u] = make_float2(0.f, 0.f);
            }
        }
    };
    auto comp = [&](const float2(&av)[U], const float2(&bv)[U], int m) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc[1][1], 0, 0, 0);
            if (do_bias && ((m + 2 * u + hi) % bstride) == 0) { bs0 += av[u].x; bs1 += av[u].y; }
        }
    };
    load(a0, b0, m0);
#pragma unroll 1
    for (int m = m0; m < m1; m += 4 * U) {
        load(a1, b1, m + 2 * U);
        comp(a0, b0, m);
        load(a0, b0, m + 4 * U);
        comp(a1, b1, m + 2 * U);
    }
    // acc[t][t'][r]: n = nb*64 + 2*i + t with i = (r&3) + 8*(r>>2) + 4*hi ; k = kb*64 + 2*lo + t'
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const int k = kb * 64 + 2 * lo + tp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * hi) + t;
                if (n < P.N && k < P.K) atomicAdd(P.out + (size_t)n * P.ldo + k, acc[t][tp][r]);
            }
        }
    if (do_bias) {
        bs0 += __shfl_xor(bs0, 32, 64);
        bs1 += __shfl_xor(bs1, 32, 64);
        if (hi == 0) {
            const int n = nb * 64 + 2 * lo;
            if (n < P.N) atomicAdd(P.bias_out + n, bs0);
            if (n + 1 < P.N) atomicAdd(P.bias_out + n + 1, bs1);
        }
    }
}

// tiny-N layers (3 / 1 outputs): out[n][k] += sum_m dA[m][n] X[m][k], one thread per k (K <= 256).
// dA == nullptr means dA = 1 (column sums of X).
__global__ __launch_bounds__(256) void k_wgrad_small(const float* __restrict__ X, int ldx, const float* __restrict__ dA, int lda, int M, int K,
                                                     int N, float* __restrict__ out, int ldo, float* __restrict__ bias_out, int bias_stride,
                                                     int MC) {
    const int k = threadIdx.x;
    const int m0 = blockIdx.x * MC, m1 = min(m0 + MC, M);
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int m = m0; m < m1; ++m) {
        const float x = k < K ? X[(size_t)m * ldx + k] : 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (n < N) {
                const float d = dA ? dA[(size_t)m * lda + n] : 1.f;
                acc[n] = fmaf(d, x, acc[n]);
                if ((m % bias_stride) == 0) bs[n] += d;
            }
        }
    }
    if (k < K)
        for (int n = 0; n < N; ++n) atomicAdd(out + (size_t)n * ldo + k, acc[n]);
    if (bias_out && k == 0)
        for (int n = 0; n < N; ++n) atomicAdd(bias_out + n, bs[n]);
}

static int launch_group(WgProb* probs, int nprob, int kid, long long rows, hipStream_t st) {
    if (nprob == 0) return ST_OK;
    ES_REQUIRE(nprob <= WG_MAX_PROBS, "too many weight-gradient problems in one group");
    // rows per task: aim at a few thousand wavefront tasks per launch
    double work = 0;
    for (int i = 0; i < nprob; ++i) work += (double)probs[i].M * ((probs[i].K + 63) / 64) * ((probs[i].N + 63) / 64);
    int MC = (int)(work / 3072.0);
    MC = (MC + 63) / 64 * 64;
    if (MC < 256) MC = 256;
    if (MC > 8192) MC = 8192;
    WgArgs a;
    int total = 0;
    for (int i = 0; i < nprob; ++i) {
        probs[i].task_begin = total;
        total += ((probs[i].K + 63) / 64) * ((probs[i].N + 63) / 64) * ((probs[i].M + MC - 1) / MC);
        a.p[i] = probs[i];
    }
    a.nprob = nprob; a.total_tasks = total; a.MC = MC;
    ScopedTimer tm(kid, rows, st);
    hipLaunchKernelGGL(k_wgrad, dim3((total + 3) / 4), dim3(256), 0, st, a);
    return ST_OK;
}
static void launch_small(const float* X, int ldx, const float* dA, int lda, int M, int K, int N, float* out, int ldo, float* bias_out,
                         int bias_stride, hipStream_t st) {
    int MC = (M + 4095) / 4096;        // up to 4096 blocks: the kernel is a pure stream over X
    MC = (MC + 31) / 32 * 32;
    if (MC < 32) MC = 32;
    ScopedTimer tm(KID_WGRAD_SMALL, M, st);
    hipLaunchKernelGGL(k_wgrad_small, dim3((M + MC - 1) / MC), dim3(256), 0, st, X, ldx, dA, lda, M, K, N, out, ldo, bias_out, bias_stride, MC);
}

// All weight gradients of one point evaluation, accumulated (+=) into dweff (es_weff layout).
int point_wgrad(int M, float* ws, int flags, const float* d_sdf, float* dweff, hipStream_t st) {
    if (M <= 0) return ST_OK;
    const WsLayout L = ws_layout(M, flags);
    const Tabs tb = make_tabs();
    const int Mp = L.Mp;
    const size_t t256 = (size_t)Mp * 256;
    auto B = [&](int buf) { return ws + L.off[buf]; };
    auto dW = [&](int net, int l) { return dweff + tb.woff[net * LAYERS + l]; };
    auto dB = [&](int net, int l) { return dweff + tb.boff[net * LAYERS + l]; };
    WgProb g[WG_MAX_PROBS];
    int n = 0;
    auto add = [&](const float* X, int ldx, const float* dA, int lda, int rows, int K, int N, float* out, int ldo, float* bias, int bstride) {
        g[n++] = WgProb{X, dA, out, bias, ldx, lda, ldo, rows, K, N, bstride, 0};
    };
    if (flags & PF_DEFORM) {
        const int R = 4 * Mp;
        const size_t r256 = (size_t)R * 256;
        n = 0;
        add(B(WS_D_U0), 64, B(WS_D_A), 256, R, 52, 256, dW(NET_D, 0), 52, dB(NET_D, 0), 4);
        for (int l = 1; l <= 7; ++l)
            add(B(WS_D_U) + (size_t)(l - 1) * r256, 256, B(WS_D_A) + (size_t)l * r256, 256, R, 256, LAYER_N[NET_D][l], dW(NET_D, l), 256,
                dB(NET_D, l), 4);
        if (int e = launch_group(g, n, KID_WGRAD_D, M, st)) return e;
        launch_small(B(WS_D_U) + (size_t)7 * r256, 256, B(WS_D_A8), 4, R, 256, 3, dW(NET_D, 8), 256, dB(NET_D, 8), 4, st);
    }
    {   // SDF: value-pass pairs (s_l, zbar_l) and reverse-pass pairs (tau_l, rho_l)
        n = 0;
        add(B(WS_S_S0), 64, B(WS_S_ZB), 256, Mp, 39, 256, dW(NET_S, 0), 39, dB(NET_S, 0), 1);
        add(B(WS_S_TAU0), 64, B(WS_S_RHO), 256, Mp, 39, 256, dW(NET_S, 0), 39, nullptr, 1);
        for (int l = 1; l <= 7; ++l) {
            const int K = LAYER_K[NET_S][l];
            add(B(WS_S_ACT) + (size_t)(l - 1) * t256, 256, B(WS_S_ZB) + (size_t)l * t256, 256, Mp, 256, 256, dW(NET_S, l), K, dB(NET_S, l), 1);
            add(B(WS_S_TAU) + (size_t)(l - 1) * t256, 256, B(WS_S_RHO) + (size_t)l * t256, 256, Mp, 256, 256, dW(NET_S, l), K, nullptr, 1);
            if (l == 4) {   // skip layer: encoding columns 256..294
                add(B(WS_S_S0), 64, B(WS_S_ZB) + (size_t)4 * t256, 256, Mp, 39, 256, dW(NET_S, 4) + 256, K, nullptr, 1);
                add(B(WS_S_TAU0), 64, B(WS_S_RHO) + (size_t)4 * t256, 256, Mp, 39, 256, dW(NET_S, 4) + 256, K, nullptr, 1);
            }
        }
        if (flags & PF_COLOR)   // feature rows 1..256 of the last layer
            add(B(WS_S_ACT) + (size_t)7 * t256, 256, B(WS_FEATBAR), 256, Mp, 256, 256, dW(NET_S, 8) + 256, 256, dB(NET_S, 8) + 1, 1);
        if (int e = launch_group(g, n, KID_WGRAD_S, M, st)) return e;
        // row 0 of the last layer: sdfbar^T s_8  +  column sums of tau_8 (adjoint of the reverse sweep's seed row)
        launch_small(B(WS_S_ACT) + (size_t)7 * t256, 256, d_sdf, 1, M, 256, 1, dW(NET_S, 8), 256, dB(NET_S, 8), 1, st);   // real rows only: d_sdf is [M]
        launch_small(B(WS_S_TAU) + (size_t)7 * t256, 256, nullptr, 1, Mp, 256, 1, dW(NET_S, 8), 256, nullptr, 1, st);
    }
    if (flags & PF_COLOR) {
        n = 0;
        add(B(WS_C_IN), 128, B(WS_C_Y), 256, Mp, 93, 256, dW(NET_C, 0), 349, dB(NET_C, 0), 1);
        add(B(WS_FEAT), 256, B(WS_C_Y), 256, Mp, 256, 256, dW(NET_C, 0) + 93, 349, nullptr, 1);
        for (int l = 1; l <= 7; ++l) {
            const int K = LAYER_K[NET_C][l];
            add(B(WS_C_H) + (size_t)(l - 1) * t256, 256, B(WS_C_Y) + (size_t)l * t256, 256, Mp, 256, 256, dW(NET_C, l), K, dB(NET_C, l), 1);
            if (l == 4) {
                add(B(WS_C_IN), 128, B(WS_C_Y) + (size_t)4 * t256, 256, Mp, 93, 256, dW(NET_C, 4) + 256, K, nullptr, 1);
                add(B(WS_FEAT), 256, B(WS_C_Y) + (size_t)4 * t256, 256, Mp, 256, 256, dW(NET_C, 4) + 349, K, nullptr, 1);
            }
        }
        if (int e = launch_group(g, n, KID_WGRAD_C, M, st)) return e;
        launch_small(B(WS_C_H) + (size_t)7 * t256, 256, B(WS_C_Y8), 4, Mp, 256, 3, dW(NET_C, 8), 256, dB(NET_C, 8), 1, st);
    }
    return hip_last("point_wgrad");
}

}  // namespace es
