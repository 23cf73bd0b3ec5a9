// Dev micro-benchmark: intra-workgroup overlap of a layer's epilogue (operand load, arithmetic, LDS + HBM stores) with the
// GEMM of the other half of the tile ("ping-pong" over two 32-row halves) against the plain GEMM -> barrier -> epilogue ->
// barrier structure of the chain kernels.  Same work: 64-row tiles, 256x256 layers, one load + one store per element.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "chain_common.h"
using namespace es;

// gemm_seg<32, 1, 2> with a callback after each outer iteration (8 of them for KG = 32, PF = 2)
template <class CB>
__device__ __forceinline__ void gemm_half_cb(f32x16 (&acc)[1][2], const float* At, const float4* __restrict__ W, int rt0, int nt0, int lane,
                                             CB&& cb) {
    constexpr int KG = 32, PF = 2, NTC = 2;
    const int lo = lane & 31, hi = lane >> 5;
    float4 b0[PF][NTC], b1[PF][NTC];
    const float4* wl = W + lane;
    int aoff[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) aoff[c] = (((rt0 * 32 + lo) ^ (hi << 2)) ^ (8 * c)) + 64 * hi;
    auto loadB = [&](float4(&b)[PF][NTC], int g0) {
#pragma unroll
        for (int gi = 0; gi < PF; ++gi)
#pragma unroll
            for (int ni = 0; ni < NTC; ++ni) b[gi][ni] = wl[(size_t)((nt0 + ni) * KG + g0 + gi) * 64];
    };
    auto comp = [&](const float4(&b)[PF][NTC], int g0) {
        const float* Ag = At + 512 * g0;
#pragma unroll
        for (int gi = 0; gi < PF; ++gi)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = Ag[aoff[4 * (gi & 1) + j] + 512 * gi + 128 * j];
#pragma unroll
                for (int ni = 0; ni < NTC; ++ni)
                    acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, f4c(b[gi][ni], j), acc[0][ni], 0, 0, 0);
            }
    };
    loadB(b0, 0);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int g0 = 4 * it;
        loadB(b1, g0 + PF);
        comp(b0, g0);
        if (g0 + 2 * PF < KG) loadB(b0, g0 + 2 * PF);
        comp(b1, g0 + PF);
        cb(it);
    }
}

// MODE 0: plain structure (full-tile GEMM, barrier, epilogue with load + store, barrier);  MODE 1: ping-pong halves
template <int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void k(const float4* __restrict__ W, const float* __restrict__ bias, float* __restrict__ out,
                                                  const float* __restrict__ in, int layers) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < MAIN_FLOATS; i += NTHREADS) mainT[i] = 1e-3f * (i & 31);
    __syncthreads();
    const size_t grow0 = (size_t)blockIdx.x * TM;
    const size_t lstride = (size_t)gridDim.x * TM * 256;
    if (MODE == 0) {
#pragma unroll 1
        for (int l = 0; l < layers; ++l) {
            f32x16 acc[2][2];
            acc_zero(acc);
            gemm_seg<32, 2, 2>(acc, mainT, W + (size_t)(l & 7) * 8 * 32 * 64, 0, 2 * wave, lane);
            __syncthreads();
            float* ol = out + (size_t)(l & 7) * lstride;
            const float* il = in + (size_t)((l + 3) & 7) * lstride;
            for_quads(acc, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4]) {
                float s[4];
                g_load_quad(il, grow0, 256, row, col, s);
                const float b = bias[col];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i] * 1e-3f + b + s[i], 0.f);
                lds_store_quad(mainT, col, row, v);
                g_store_quad(ol, grow0, 256, row, col, v);
            });
            __syncthreads();
        }
    } else {
        f32x16 accA[1][2], accB[1][2];
        float opA[8][4], opB[8][4];
        auto quad_rc = [&](int half, int q8, int& row, int& col) {      // quad q8 = ni*4 + q of a half
            row = half * 32 + 8 * (q8 & 3) + 4 * hi;
            col = (2 * wave + (q8 >> 2)) * 32 + lo;
        };
        auto prefetch = [&](float(&op)[8][4], int half, int l) {
            const float* il = in + (size_t)((l + 3) & 7) * lstride;
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) { int row, col; quad_rc(half, q8, row, col); g_load_quad(il, grow0, 256, row, col, op[q8]); }
        };
        auto epi_quad = [&](f32x16(&acc)[1][2], const float(&op)[8][4], int half, int l, int q8) {
            int row, col; quad_rc(half, q8, row, col);
            const int ni = q8 >> 2, q = q8 & 3;
            float v[4] = {acc[0][ni][4 * q], acc[0][ni][4 * q + 1], acc[0][ni][4 * q + 2], acc[0][ni][4 * q + 3]};
            const float b = bias[col];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i] * 1e-3f + b + op[q8][i], 0.f);
            lds_store_quad(mainT, col, row, v);
            g_store_quad(out + (size_t)(l & 7) * lstride, grow0, 256, row, col, v);
        };
        // prologue: GEMM A(0)
        acc_zero(accA);
        prefetch(opA, 0, 0);
        gemm_half_cb(accA, mainT, W, 0, 2 * wave, lane, [](int) {});
        __syncthreads();
#pragma unroll 1
        for (int l = 0; l < layers; ++l) {
            const float4* Wl = W + (size_t)(l & 7) * 8 * 32 * 64;
            const float4* Wn = W + (size_t)((l + 1) & 7) * 8 * 32 * 64;
            // phase X: GEMM B(l) with the epilogue of A(l) inside
            acc_zero(accB);
            prefetch(opB, 1, l);
            gemm_half_cb(accB, mainT, Wl, 1, 2 * wave, lane, [&](int it) { epi_quad(accA, opA, 0, l, it); });
            __syncthreads();
            // phase Y: GEMM A(l+1) with the epilogue of B(l) inside
            acc_zero(accA);
            prefetch(opA, 0, l + 1);
            gemm_half_cb(accA, mainT, Wn, 0, 2 * wave, lane, [&](int it) { epi_quad(accB, opB, 1, l, it); });
            __syncthreads();
        }
        float sink = 0.f;
        for (int r = 0; r < 16; ++r) sink += accA[0][0][r] + accA[0][1][r];
        if (sink == 123.456f) out[tid] = sink;
    }
}

template <int MODE>
static void run(const char* name, const float4* W, const float* bias, float* out, const float* in, int blocks, int layers) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LEAN_LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(NTHREADS), LEAN_LDS_BYTES, 0, W, bias, out, in, layers);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(NTHREADS), LEAN_LDS_BYTES, 0, W, bias, out, in, layers);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double fl = 2.0 * 64 * 256 * 256 * (double)layers * blocks;
    printf("%-50s %4d blocks %7.3f ms  %6.1f TFLOP/s (%.1f %%)\n", name, blocks, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}

int main() {
    float4* W; float *bias, *out, *in;
    const int blocks = 1024, layers = 64;
    hipMalloc(&W, (size_t)8 * 8 * 32 * 64 * 16); hipMalloc(&bias, 1024);
    hipMalloc(&out, (size_t)8 * blocks * TM * 256 * 4); hipMalloc(&in, (size_t)8 * blocks * TM * 256 * 4);
    hipMemset(W, 0, (size_t)8 * 8 * 32 * 64 * 16); hipMemset(bias, 0, 1024); hipMemset(in, 0, (size_t)8 * blocks * TM * 256 * 4);
    run<0>("plain: GEMM | barrier | load+store epilogue", W, bias, out, in, blocks, layers);
    run<1>("ping-pong halves: epilogue inside the other GEMM", W, bias, out, in, blocks, layers);
    run<0>("plain, 512 blocks", W, bias, out, in, 512, layers);
    run<1>("ping-pong, 512 blocks", W, bias, out, in, 512, layers);
    return 0;
}
