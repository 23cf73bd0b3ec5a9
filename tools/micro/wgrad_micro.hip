// Dev micro-benchmark: the k_wgrad task loop on one synthetic [M x 256]^T [M x 256] problem, with features toggled at
// compile time, to find what keeps the loop below the LDS-operand MFMA ceiling (mfma_peak.hip: ~93 %).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int WG_R = 16, KW = 128, WG_THREADS = 512;
constexpr int LDSF = 2 * WG_R * (256 + KW);

// F bits: 1 bias branch (as in the product kernel), 2 global loads in loop, 4 LDS stores + barriers, 8 atomics epilogue
template <int F>
__global__ __launch_bounds__(WG_THREADS, 4) void k(const float* __restrict__ dA, const float* __restrict__ X, float* out, float* bias_out,
                                                    int M, int MC, int bias_stride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int task = blockIdx.x;
    const int grp = task / 16, j = task % 16;
    const int mc = grp * 8 + (j & 7), kb = j >> 3;
    const int m0 = mc * MC, m1 = min(m0 + MC, M);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = w & 3, kh = w >> 2, lo = lane & 31, hi = lane >> 5;
    auto Apan = [&](int buf) { return lds + buf * (WG_R * 256); };
    auto Bpan = [&](int buf) { return lds + 2 * WG_R * 256 + buf * (WG_R * KW); };
    const int kcol0 = kb * KW;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bs0 = 0.f, bs1 = 0.f;
    const bool do_bias = (F & 1) && bias_out != nullptr && kb == 0 && kh == 0;
    const int fa1 = tid + WG_THREADS;
    const size_t offA0 = (size_t)(tid >> 6) * 256 + 4 * (tid & 63), offA1 = (size_t)(fa1 >> 6) * 256 + 4 * (fa1 & 63);
    const int colB = kcol0 + 4 * (tid % (KW / 4));
    const size_t offB = (size_t)(tid / (KW / 4)) * 256 + colB;
#define GLOAD(S, m) { const float* pa = dA + (size_t)(m) * 256; S##a0 = *reinterpret_cast<const v4f*>(pa + offA0); \
        S##a1 = *reinterpret_cast<const v4f*>(pa + offA1); S##b0 = *reinterpret_cast<const v4f*>(X + (size_t)(m) * 256 + offB); }
#define SSTORE(S, buf) { *reinterpret_cast<v4f*>(Apan(buf) + 4 * tid) = S##a0; *reinterpret_cast<v4f*>(Apan(buf) + 4 * fa1) = S##a1; \
        *reinterpret_cast<v4f*>(Bpan(buf) + 4 * tid) = S##b0; }
    auto compute = [&](int buf, int m) {
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
            if (do_bias && ((m + 2 * s + hi) & (bias_stride - 1)) == 0) { bs0 += av.x; bs1 += av.y; }
        }
    };
    const int nst = (m1 - m0) / WG_R;
    v4f p0a0, p0a1, p0b0, p1a0, p1a1, p1b0;
    GLOAD(p0, m0);
    GLOAD(p1, m0 + WG_R);
    SSTORE(p0, 0);
    __syncthreads();
#pragma unroll 1
    for (int st = 0; st < nst; st += 2) {
        if (F & 2) if (st + 2 < nst) GLOAD(p0, m0 + WG_R * (st + 2));
        compute(0, m0 + WG_R * st);
        if (F & 4) { SSTORE(p1, 1); __syncthreads(); }
        if (F & 2) if (st + 3 < nst) GLOAD(p1, m0 + WG_R * (st + 3));
        compute(1, m0 + WG_R * (st + 1));
        if (F & 4) { if (st + 2 < nst) SSTORE(p0, 0); __syncthreads(); }
    }
    float sink = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const int kk = kcol0 + kh * 64 + 2 * lo + tp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * hi) + t;
                if (F & 8) atomicAdd(out + (size_t)n * 256 + kk, acc[t][tp][r]); else sink += acc[t][tp][r];
            }
        }
    if (!(F & 8) && sink == 123.456f) out[tid] = sink + p0a0.x + p1a0.x + p0b0.x + p1b0.x + p0a1.x + p1a1.x;
    if (do_bias) {
        bs0 += __shfl_xor(bs0, 32, 64); bs1 += __shfl_xor(bs1, 32, 64);
        if (hi == 0) { atomicAdd(bias_out + nb * 64 + 2 * lo, bs0); atomicAdd(bias_out + nb * 64 + 2 * lo + 1, bs1); }
    }
}

template <int F>
static void run(const char* name, const float* dA, const float* X, float* out, float* bias, int M, int MC) {
    hipFuncSetAttribute((const void*)k<F>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSF * 4);
    const int chunks = M / MC, tasks = 2 * chunks;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<F>, dim3(tasks), dim3(WG_THREADS), LDSF * 4, 0, dA, X, out, bias, M, MC, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<F>, dim3(tasks), dim3(WG_THREADS), LDSF * 4, 0, dA, X, out, bias, M, MC, 4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double fl = 2.0 * M * 256 * 256;
    printf("%-52s tasks %4d  %7.3f ms  %6.1f TFLOP/s (%.1f %%)\n", name, tasks, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}

int main() {
    const int MC = 2688, chunks = 768;            // 1536 tasks = 3 rounds of 512
    const int M = MC * chunks;                    // 2.06M rows: 2 x 2.1 GB operands
    float *dA, *X, *out, *bias;
    hipMalloc(&dA, (size_t)M * 256 * 4); hipMalloc(&X, (size_t)M * 256 * 4); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&bias, 1024);
    hipMemset(dA, 0, (size_t)M * 256 * 4); hipMemset(X, 0, (size_t)M * 256 * 4); hipMemset(out, 0, 256 * 256 * 4); hipMemset(bias, 0, 1024);
    run<15>("product loop (bias branch, loads, stores, atomics)", dA, X, out, bias, M, MC);
    run<14>("no bias branch", dA, X, out, bias, M, MC);
    run<6>("no bias branch, no atomics", dA, X, out, bias, M, MC);
    run<4>("stores+barriers only (no loads, bias, atomics)", dA, X, out, bias, M, MC);
    run<2>("loads only", dA, X, out, bias, M, MC);
    run<0>("LDS reads + MFMA only", dA, X, out, bias, M, MC);
    run<1>("bias branch only", dA, X, out, bias, M, MC);
    return 0;
}
