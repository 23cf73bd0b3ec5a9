// Dev micro-benchmark (round 5): an fp32 chain with REGISTER-RESIDENT activations -- the formulation of the split-precision
// kernels (csrc/x3r_core.h) on v_mfma_f32_32x32x2_f32.  Question: with the transposed product Y^T = W X^T a wave that owns 32 points
// and ALL 256 features holds, after a layer, exactly the B operands of the next one (step (b, r): feature 32 b + 8 (r >> 2) + (r & 3)
// from the hi = 0 lanes, + 4 from the hi = 1 lanes = accumulator register r of block b of EVERY lane), so activations never touch LDS and
// the weights -- the only stream -- can be shared by the 4 waves of a workgroup through an LDS ring filled by direct loads
// (global_load_lds): 256 KB of L2 traffic per layer and 128 points instead of per 64, no weight VGPR staging, no LDS activation traffic.
// Does one wave per SIMD then run closer to the MFMA rate than the shipped k-major kernels (0.82-0.88 of it in the GEMM)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/regres_micro tools/micro/regres_micro.hip && /tmp/regres_micro
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int RING = 8;                 // k-groups (of 4 k-steps = 8 k) resident in LDS: 8 KB each
constexpr int GROUP_BYTES = 8 * 64 * 16;
constexpr int LDS_BYTES = RING * GROUP_BYTES + 1024;

__device__ __forceinline__ float f4c(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }
__device__ __forceinline__ float softplus100(float z) {
    const float e = __builtin_amdgcn_exp2f(-144.26950408889634f * fabsf(z));
    return fmaf(0.006931471805599453f, __builtin_amdgcn_logf(1.f + e), fmaxf(z, 0.f));
}

// MODE bits: 1 = weight ring refilled by direct loads (else the ring is static), 2 = barriers, 4 = softplus (else ReLU), 8 = no activation at all,
// 16 = only ONE of a wave's two pieces per k-group is loaded (half the instructions, half the bytes), 32 = both pieces, but always the
// SAME 2 KB of the packed buffer (same instruction count, the bytes come from a hot cache line set)
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_regres(const float4* __restrict__ Wp, const float* __restrict__ bias, float* __restrict__ out, int layers) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* btab = reinterpret_cast<float*>(lds + RING * GROUP_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    btab[tid] = bias[tid];
    for (int i = tid; i < RING * GROUP_BYTES / 16; i += 256) reinterpret_cast<float4*>(lds)[i] = make_float4(1e-3f, -1e-3f, 2e-3f, -2e-3f);
    f32x16 X[8], Y[8];
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { X[b][r] = 0.01f * (lane & 15) - 0.05f + 0.001f * r; Y[b][r] = 0.f; }
    const float4* gsrc = Wp + lane;
    const long n_groups = (long)layers * 32;
    auto dma = [&](long G) {      // this wave's 2 of the 8 1-KB pieces of k-group G
        if (!(MODE & 1) || G >= n_groups) return;
#pragma unroll
        for (int i = 0; i < ((MODE & 16) ? 1 : 2); ++i) {
            const int piece = 2 * wave + i;
            const float4* src = gsrc + (((MODE & 32) ? 0 : G) * 8 + piece) * 64;
            unsigned char* dst = lds + ((G & (RING - 1)) * 8 + piece) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    auto frags = [&](float4(&A)[8], long G) {
        const float4* p = reinterpret_cast<const float4*>(lds + (G & (RING - 1)) * GROUP_BYTES) + lane;
#pragma unroll
        for (int m = 0; m < 8; ++m) A[m] = p[m * 64];
    };
    __syncthreads();
    for (long G = 0; G < 4; ++G) dma(G);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float4 A0[8], A1[8];
    frags(A0, 0);

    auto layer = [&](f32x16(&S)[8], f32x16(&D)[8], int l) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) D[m][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            const int b = g >> 2, q = g & 3;
            const long G = (long)l * 32 + g;
            if ((g & 1) == 0) { dma(G + 4); dma(G + 5); }
            if (!(MODE & 8)) {      // lazy activation of this group's four operand elements
                const float4 bb = *reinterpret_cast<const float4*>(btab + 32 * b + 8 * q + 4 * hi);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float z = S[b][4 * q + i] + f4c(bb, i);
                    S[b][4 * q + i] = (MODE & 4) ? softplus100(z) : fmaxf(z, 0.f);
                }
            }
            float4(&A)[8] = (g & 1) ? A1 : A0;
            float4(&An)[8] = (g & 1) ? A0 : A1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if ((g & 1) == 0 && i == 0) frags(An, G + 1);                  // the pair's second group: landed with the pair
                if ((g & 1) == 1 && i == 2) {                                    // middle of the pair's second group: the next pair must have landed
                    if (MODE & 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    if (MODE & 2) __builtin_amdgcn_s_barrier();
                    frags(An, G + 1);
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) D[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(A[m], i), S[b][4 * q + i], D[m], 0, 0, 0);
            }
        }
    };
#pragma unroll 1
    for (int l = 0; l < layers; l += 2) {
        layer(X, Y, l);
        layer(Y, X, l + 1);
    }
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += X[b][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static void run(const char* what, const float4* W, const float* bias, float* out, int blocks, int layers) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_regres<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_regres<MODE>, dim3(blocks), dim3(256), LDS_BYTES, 0, W, bias, out, layers);
    hipEventRecord(a);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_regres<MODE>, dim3(blocks), dim3(256), LDS_BYTES, 0, W, bias, out, layers);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    const double flop = 2.0 * 128 * 256 * 256 * (double)layers * blocks;
    printf("%-78s %8.3f ms  %6.1f TF  %5.1f %% of 157.3   (%s)\n", what, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3 * 100, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int blocks = 1024, layers = 14;
    float4* W; float *bias, *out;
    const size_t wbytes = (size_t)64 * 32 * GROUP_BYTES;
    hipMalloc(&W, wbytes); hipMalloc(&bias, 1024); hipMalloc(&out, (size_t)blocks * 256 * 4);
    std::vector<float> h(wbytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((int)(i * 2654435761u >> 20) % 200 - 100) * 1e-4f;
    hipMemcpy(W, h.data(), wbytes, hipMemcpyHostToDevice);
    hipMemset(bias, 0, 1024);
    printf("register-resident fp32 chain, 128 points per workgroup (1 wave per SIMD), %d blocks x %d layers of 256 x 256\n", blocks, layers);
    run<8 + 2>("MFMAs + fragment reads from a static ring + one barrier per pair of k-groups", W, bias, out, blocks, layers);
    run<8 + 2 + 1>("+ weight ring refilled by direct loads (256 KB per layer and workgroup)", W, bias, out, blocks, layers);
    run<8 + 2 + 1 + 16>("  ... one piece per wave and k-group instead of two (half instructions, half bytes)", W, bias, out, blocks, layers);
    run<8 + 2 + 1 + 32>("  ... two pieces, always the same 2 KB (same instructions, hot lines)", W, bias, out, blocks, layers);
    run<2 + 1>("+ bias + ReLU, lazily on the operand registers", W, bias, out, blocks, layers);
    run<4 + 2 + 1>("+ bias + softplus(100) instead", W, bias, out, blocks, layers);
    run<4 + 2 + 1>("same, 64 layers", W, bias, out, blocks, 64);
    run<4 + 2 + 1>("same, 256 blocks (one round)", W, bias, out, 256, layers);
    return 0;
}
