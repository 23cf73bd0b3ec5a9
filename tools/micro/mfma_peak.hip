// Dev micro-benchmark: what fraction of the fp32 MFMA peak do simple loops reach on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/micro/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ __launch_bounds__(512, 4) void k512(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 16 * 384];
    const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5, w = tid >> 6;
    for (int i = tid; i < 2 * 16 * 384; i += 512) lds[i] = 1e-3f * (i & 63);
    __syncthreads();
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = lane * 1e-3f, b = 1.f;
    const float* A = lds + (w & 3) * 64 + 2 * lo;
    const float* B = lds + 2 * 16 * 256 + (w >> 2) * 64 + 2 * lo;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j % NACC], 0, 0, 0);
        } else {
            const int buf = it & 1;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float2 av = *reinterpret_cast<const float2*>(A + buf * 16 * 256 + (2 * s + hi) * 256);
                const float2 bv = *reinterpret_cast<const float2*>(B + buf * 16 * 128 + (2 * s + hi) * 128);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[0], 0, 0, 0);
                acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.y, acc[1 % NACC], 0, 0, 0);
                acc[2 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, acc[2 % NACC], 0, 0, 0);
                acc[3 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[3 % NACC], 0, 0, 0);
            }
            if (MODE == 2) __syncthreads();
        }
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 512 + tid] = s;
}

// 256-thread variant: 4 waves per workgroup, WPS workgroups... occupancy chosen by launch bounds
template <int MINW>
__global__ __launch_bounds__(256, MINW) void k256(float* out, int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = lane * 1e-3f, b = 1.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <class F>
static void run(const char* name, F launch, double flops) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 157.3)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main() {
    float* out; hipMalloc(&out, 4 * 1024 * 1024 * 4);
    const int iters = 4000;
    const double per_wave_iter = 32.0 * 32 * 32 * 2 * 2;      // 32 MFMAs
    {
        const int blocks = 512;
        const double fl = per_wave_iter * iters * 8.0 * blocks;
        run("512thr x512 blocks, regs only, 4 acc", [&] { hipLaunchKernelGGL((k512<0, 4>), dim3(blocks), dim3(512), 0, 0, out, iters); }, fl);
        run("512thr x512 blocks, regs only, 2 acc", [&] { hipLaunchKernelGGL((k512<0, 2>), dim3(blocks), dim3(512), 0, 0, out, iters); }, fl);
        run("512thr x512 blocks, regs only, 1 acc", [&] { hipLaunchKernelGGL((k512<0, 1>), dim3(blocks), dim3(512), 0, 0, out, iters); }, fl);
        run("512thr x512 blocks, LDS operands", [&] { hipLaunchKernelGGL((k512<1, 4>), dim3(blocks), dim3(512), 0, 0, out, iters); }, fl);
        run("512thr x512 blocks, LDS operands + barrier", [&] { hipLaunchKernelGGL((k512<2, 4>), dim3(blocks), dim3(512), 0, 0, out, iters); }, fl);
        run("512thr x256 blocks (1/CU), LDS + barrier", [&] { hipLaunchKernelGGL((k512<2, 4>), dim3(256), dim3(512), 0, 0, out, iters); }, fl / 2);
    }
    {
        const double fl1 = per_wave_iter * iters * 4.0;
        run("256thr x256 blocks (1 wave/SIMD)", [&] { hipLaunchKernelGGL((k256<1>), dim3(256), dim3(256), 0, 0, out, iters); }, fl1 * 256);
        run("256thr x512 blocks (2 waves/SIMD)", [&] { hipLaunchKernelGGL((k256<2>), dim3(512), dim3(256), 0, 0, out, iters); }, fl1 * 512);
        run("256thr x1024 blocks (4 waves/SIMD)", [&] { hipLaunchKernelGGL((k256<4>), dim3(1024), dim3(256), 0, 0, out, iters); }, fl1 * 1024);
    }
    return 0;
}
