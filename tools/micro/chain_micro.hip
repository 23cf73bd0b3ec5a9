// Dev micro-benchmark: the chain kernels' layer loop (gemm_seg<32,2,2>: A from the swizzled LDS tile, B streamed from L2 in
// packed-fragment order) at 2 workgroups per CU, with the per-layer barriers / epilogue features toggled.
//   hipcc --offload-arch=gfx950 -O3 -I endosurf_amd/csrc -I include -o tools/micro/chain_micro.bin tools/micro/chain_micro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "chain_common.h"
using namespace es;

// candidate inner loop: the swizzled A addresses are 16 per-lane registers (8 XOR patterns x 2 row tiles) + immediates
template <int KG, int RTC, int NTC>
__device__ __forceinline__ void gemm_seg2(f32x16 (&acc)[RTC][NTC], const float* At, const float4* __restrict__ W, int rt0, int nt0, int lane) {
    constexpr int PF = 4;
    static_assert(KG % (2 * PF) == 0, "micro: full groups only");
    const int lo = lane & 31, hi = lane >> 5;
    float4 b0[PF][NTC], b1[PF][NTC];
    const float4* wl = W + lane;
    int aoff[RTC][8];
#pragma unroll
    for (int ri = 0; ri < RTC; ++ri)
#pragma unroll
        for (int c = 0; c < 8; ++c) aoff[ri][c] = ((((rt0 + ri) * 32 + lo) ^ (hi << 2)) ^ (8 * c)) + 64 * hi;
    auto loadB = [&](float4(&b)[PF][NTC], int g0) {
#pragma unroll
        for (int gi = 0; gi < PF; ++gi)
#pragma unroll
            for (int ni = 0; ni < NTC; ++ni) b[gi][ni] = wl[(size_t)((nt0 + ni) * KG + g0 + gi) * 64];
    };
    auto comp = [&](const float4(&b)[PF][NTC], const float* Ag, int gpar) {      // Ag = At + 512 * g0
#pragma unroll
        for (int gi = 0; gi < PF; ++gi) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a[RTC];
#pragma unroll
                for (int ri = 0; ri < RTC; ++ri) a[ri] = Ag[aoff[ri][4 * ((gi + gpar) & 1) + j] + 512 * gi + 128 * j];
#pragma unroll
                for (int ri = 0; ri < RTC; ++ri)
#pragma unroll
                    for (int ni = 0; ni < NTC; ++ni)
                        acc[ri][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ri], f4c(b[gi][ni], j), acc[ri][ni], 0, 0, 0);
            }
        }
    };
    loadB(b0, 0);
#pragma unroll 1
    for (int g0 = 0; g0 < KG; g0 += 2 * PF) {
        loadB(b1, g0 + PF);
        comp(b0, At + 512 * g0, 0);
        if (g0 + 2 * PF < KG) loadB(b0, g0 + 2 * PF);
        comp(b1, At + 512 * (g0 + PF), 0);
    }
}

// F bits: 1 = two barriers per layer, 2 = epilogue (bias + relu + LDS store), 4 = epilogue streams the tile to HBM
template <int F>
__global__ __launch_bounds__(NTHREADS, 2) void k(const float4* __restrict__ W, const float* __restrict__ bias, float* __restrict__ out,
                                                  int layers, int nlayer_w, int stag) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < MAIN_FLOATS; i += NTHREADS) mainT[i] = 1e-3f * (i & 31);
    __syncthreads();
    const size_t grow0 = (size_t)blockIdx.x * TM;
    float sink = 0.f;
    if (F & 32) {      // stagger the two co-resident workgroups by about half a layer so that epilogues and GEMMs interleave
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if (hwid & 1) { for (int i = 0; i < stag; ++i) __builtin_amdgcn_s_sleep(127); }
    }
#pragma unroll 1
    for (int l = 0; l < layers; ++l) {
        f32x16 acc[2][2];
        acc_zero(acc);
        if (F & 16) gemm_seg<32, 2, 2, 2>(acc, mainT, W + (size_t)(l % nlayer_w) * 8 * 32 * 64, 0, 2 * wave, lane);
        else if (F & 8) gemm_seg2<32, 2, 2>(acc, mainT, W + (size_t)(l % nlayer_w) * 8 * 32 * 64, 0, 2 * wave, lane);
        else gemm_seg<32, 2, 2>(acc, mainT, W + (size_t)(l % nlayer_w) * 8 * 32 * 64, 0, 2 * wave, lane);
        if (F & 1) __syncthreads();
        if (F & 2) {
            float* ol = out + (size_t)(l & 7) * gridDim.x * TM * 256;
            const float* il = out + (size_t)((l + 3) & 7) * gridDim.x * TM * 256;
            for_quads_qi(acc, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int qi) {
                const float b = bias[col];
                float s[4] = {0.f, 0.f, 0.f, 0.f};
                if (F & 128) {          // also LOAD an operand per element (backward-style epilogue)
                    if (F & 64) { const float4 t = *reinterpret_cast<const float4*>(il + grow0 * 256 + ((size_t)(wave * 16 + qi) * 64 + lane) * 4); s[0] = t.x; s[1] = t.y; s[2] = t.z; s[3] = t.w; }
                    else g_load_quad(il, grow0, 256, row, col, s);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i] * 1e-3f + b + s[i], 0.f);
                lds_store_quad(mainT, col, row, v);
                if (F & 4) {
                    if (F & 64) *reinterpret_cast<float4*>(ol + grow0 * 256 + ((size_t)(wave * 16 + qi) * 64 + lane) * 4) = make_float4(v[0], v[1], v[2], v[3]);
                    else g_store_quad(ol, grow0, 256, row, col, v);
                }
            });
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sink += acc[a][b][r];
        }
        if (F & 1) __syncthreads();
    }
    if (sink == 123.456f) out[tid] = sink;
}


// 8 waves per 64-row tile: wave = one 32-column n-tile (RTC = 2, NTC = 1), 4 waves per SIMD at 2 workgroups per CU (<= 128 VGPRs)
template <int F>
__global__ __launch_bounds__(512, 4) void k8(const float4* __restrict__ W, const float* __restrict__ bias, float* __restrict__ out, int layers,
                                              int nlayer_w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < MAIN_FLOATS; i += 512) mainT[i] = 1e-3f * (i & 31);
    __syncthreads();
    const size_t grow0 = (size_t)blockIdx.x * TM;
    float sink = 0.f;
#pragma unroll 1
    for (int l = 0; l < layers; ++l) {
        f32x16 acc[2][1];
        acc_zero(acc);
        gemm_seg<32, 2, 1, 2>(acc, mainT, W + (size_t)(l % nlayer_w) * 8 * 32 * 64, 0, wave, lane);
        if (F & 1) __syncthreads();
        if (F & 2) {
            float* ol = out + (size_t)(l & 7) * gridDim.x * TM * 256;
            const float* il = out + (size_t)((l + 3) & 7) * gridDim.x * TM * 256;
            for_quads_qi(acc, 0, wave, lane, [&](int row, int col, float(&v)[4], int qi) {
                const float b = bias[col];
                float s[4] = {0.f, 0.f, 0.f, 0.f};
                if (F & 128) g_load_quad_f(il, grow0, row, col, s);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i] * 1e-3f + b + s[i], 0.f);
                lds_store_quad(mainT, col, row, v);
                if (F & 4) g_store_quad_f(ol, grow0, row, col, v);
            });
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) sink += acc[a][0][r];
        }
        if (F & 1) __syncthreads();
    }
    if (sink == 123.456f) out[tid] = sink;
}
template <int F>
static void run8(const char* name, const float4* W, const float* bias, float* out, int blocks, int layers) {
    hipFuncSetAttribute((const void*)k8<F>, hipFuncAttributeMaxDynamicSharedMemorySize, LEAN_LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k8<F>, dim3(blocks), dim3(512), LEAN_LDS_BYTES, 0, W, bias, out, layers, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k8<F>, dim3(blocks), dim3(512), LEAN_LDS_BYTES, 0, W, bias, out, layers, 8);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double fl = 2.0 * 64 * 256 * 256 * (double)layers * blocks;
    printf("%-56s %4d blocks %7.3f ms  %6.1f TFLOP/s (%.1f %%)\n", name, blocks, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}


// candidate: the first weight batch of the NEXT layer is requested before this layer's epilogue stores, so that waiting for it
// does not wait for the stores (gfx9 counts loads and stores in one in-order vmcnt)
struct WB2 { float4 b[2][2]; };
__device__ __forceinline__ void wb_load(WB2& w, const float4* __restrict__ W, int nt0, int g0, int lane) {
    const float4* wl = W + lane;
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) w.b[gi][ni] = wl[(size_t)((nt0 + ni) * 32 + g0 + gi) * 64];
}
__device__ __forceinline__ void gemm_seg3(f32x16 (&acc)[2][2], const float* At, const float4* __restrict__ W, int nt0, int lane, WB2& b0) {
    const int lo = lane & 31, hi = lane >> 5;
    int aoff[2][8];
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int c = 0; c < 8; ++c) aoff[ri][c] = (((ri * 32 + lo) ^ (hi << 2)) ^ (8 * c)) + 64 * hi;
    auto comp = [&](const WB2& b, const float* Ag) {
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a[2];
#pragma unroll
                for (int ri = 0; ri < 2; ++ri) a[ri] = Ag[aoff[ri][4 * (gi & 1) + j] + 512 * gi + 128 * j];
#pragma unroll
                for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[ri][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ri], f4c(b.b[gi][ni], j), acc[ri][ni], 0, 0, 0);
            }
    };
    WB2 b1;
#pragma unroll 1
    for (int g0 = 0; g0 < 32; g0 += 4) {
        wb_load(b1, W, nt0, g0 + 2, lane);
        comp(b0, At + 512 * g0);
        if (g0 + 4 < 32) wb_load(b0, W, nt0, g0 + 4, lane);
        comp(b1, At + 512 * (g0 + 2));
    }
}
template <int F>
__global__ __launch_bounds__(NTHREADS, 2) void k3(const float4* __restrict__ W, const float* __restrict__ bias, float* __restrict__ out,
                                                   int layers, int nlayer_w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < MAIN_FLOATS; i += NTHREADS) mainT[i] = 1e-3f * (i & 31);
    __syncthreads();
    const size_t grow0 = (size_t)blockIdx.x * TM;
    WB2 b0;
    wb_load(b0, W, 2 * wave, 0, lane);
#pragma unroll 1
    for (int l = 0; l < layers; ++l) {
        f32x16 acc[2][2];
        acc_zero(acc);
        gemm_seg3(acc, mainT, W + (size_t)(l % nlayer_w) * 8 * 32 * 64, 2 * wave, lane, b0);
        __syncthreads();
        wb_load(b0, W + (size_t)((l + 1) % nlayer_w) * 8 * 32 * 64, 2 * wave, 0, lane);      // before the epilogue's stores
        float* ol = out + (size_t)(l & 7) * gridDim.x * TM * 256;
        const float* il = out + (size_t)((l + 3) & 7) * gridDim.x * TM * 256;
        for_quads_qi(acc, 0, 2 * wave, lane, [&](int row, int col, float(&v)[4], int qi) {
            const float b = bias[col];
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            if (F & 128) g_load_quad_f(il, grow0, row, col, s);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i] * 1e-3f + b + s[i], 0.f);
            lds_store_quad(mainT, col, row, v);
            g_store_quad_f(ol, grow0, row, col, v);
        });
        __syncthreads();
    }
}
template <int F>
static void run3(const char* name, const float4* W, const float* bias, float* out, int blocks, int layers) {
    hipFuncSetAttribute((const void*)k3<F>, hipFuncAttributeMaxDynamicSharedMemorySize, LEAN_LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k3<F>, dim3(blocks), dim3(NTHREADS), LEAN_LDS_BYTES, 0, W, bias, out, layers, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k3<F>, dim3(blocks), dim3(NTHREADS), LEAN_LDS_BYTES, 0, W, bias, out, layers, 8);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double fl = 2.0 * 64 * 256 * 256 * (double)layers * blocks;
    printf("%-56s %4d blocks %7.3f ms  %6.1f TFLOP/s (%.1f %%)\n", name, blocks, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}


// 32-row tiles: 32 KB of LDS and <= 128 VGPRs per workgroup => FOUR workgroups (4 waves per SIMD) per CU; twice the weight stream
// per point.  Own k-major layout [k][32 rows] with an XOR swizzle; fragment-ordered stream in / out as above (half tiles).
__device__ __forceinline__ int swz32(int k, int r) { return k * 32 + (r ^ ((k & 7) << 2)); }
template <int F>
__global__ __launch_bounds__(NTHREADS, 4) void k32(const float4* __restrict__ W, const float* __restrict__ bias, float* __restrict__ out,
                                                    int layers, int nlayer_w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 32 * 256; i += NTHREADS) mainT[i] = 1e-3f * (i & 31);
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * 32 * 256;
    const int nt0 = 2 * wave;
#pragma unroll 1
    for (int l = 0; l < layers; ++l) {
        f32x16 acc[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
        const float4* wl = W + (size_t)(l % nlayer_w) * 8 * 32 * 64 + lane;
        float4 b0[2][2], b1[2][2];
        auto loadB = [&](float4(&b)[2][2], int g0) {
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) b[gi][ni] = wl[(size_t)((nt0 + ni) * 32 + g0 + gi) * 64];
        };
        auto comp = [&](const float4(&b)[2][2], int g0) {
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = mainT[swz32(8 * (g0 + gi) + 2 * j + hi, lo)];
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, f4c(b[gi][ni], j), acc[ni], 0, 0, 0);
                }
        };
        loadB(b0, 0);
#pragma unroll 1
        for (int g0 = 0; g0 < 32; g0 += 4) {
            loadB(b1, g0 + 2);
            comp(b0, g0);
            if (g0 + 4 < 32) loadB(b0, g0 + 4);
            comp(b1, g0 + 2);
        }
        __syncthreads();
        float* ol = out + (size_t)(l & 7) * gridDim.x * 32 * 256;
        const float* il = out + (size_t)((l + 3) & 7) * gridDim.x * 32 * 256;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = (nt0 + ni) * 32 + lo, row = 8 * q + 4 * hi;
                const size_t off = base + (size_t)(((wave * 8 + ni * 4 + q) * 64 + lane) * 4);
                float v[4] = {acc[ni][4 * q], acc[ni][4 * q + 1], acc[ni][4 * q + 2], acc[ni][4 * q + 3]};
                float s[4] = {0.f, 0.f, 0.f, 0.f};
                if (F & 128) { const v4f_frag t = __builtin_nontemporal_load(reinterpret_cast<const v4f_frag*>(il + off)); s[0] = t[0]; s[1] = t[1]; s[2] = t[2]; s[3] = t[3]; }
                const float b = bias[col];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i] * 1e-3f + b + s[i], 0.f);
                *reinterpret_cast<float4*>(&mainT[swz32(col, row)]) = make_float4(v[0], v[1], v[2], v[3]);
                if (F & 4) { const v4f_frag t = {v[0], v[1], v[2], v[3]}; __builtin_nontemporal_store(t, reinterpret_cast<v4f_frag*>(ol + off)); }
            }
        __syncthreads();
    }
}
template <int F>
static void run32(const char* name, const float4* W, const float* bias, float* out, int blocks, int layers) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k32<F>, dim3(blocks), dim3(NTHREADS), 32 * 256 * 4, 0, W, bias, out, layers, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k32<F>, dim3(blocks), dim3(NTHREADS), 32 * 256 * 4, 0, W, bias, out, layers, 8);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double fl = 2.0 * 32 * 256 * 256 * (double)layers * blocks;
    printf("%-56s %4d blocks %7.3f ms  %6.1f TFLOP/s (%.1f %%)\n", name, blocks, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}

template <int F>
static void run(const char* name, const float4* W, const float* bias, float* out, int blocks, int layers, int stag = 2) {
    hipFuncSetAttribute((const void*)k<F>, hipFuncAttributeMaxDynamicSharedMemorySize, LEAN_LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(NTHREADS), LEAN_LDS_BYTES, 0, W, bias, out, layers, 8, stag);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(NTHREADS), LEAN_LDS_BYTES, 0, W, bias, out, layers, 8, stag);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double fl = 2.0 * 64 * 256 * 256 * (double)layers * blocks;
    printf("%-56s %4d blocks %7.3f ms  %6.1f TFLOP/s (%.1f %%)\n", name, blocks, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}

int main() {
    float4* W; float *bias, *out;
    const int blocks = 1024, layers = 64;
    hipMalloc(&W, (size_t)8 * 8 * 32 * 64 * 16); hipMalloc(&bias, 1024); hipMalloc(&out, (size_t)8 * blocks * TM * 256 * 4);
    hipMemset(W, 0, (size_t)8 * 8 * 32 * 64 * 16); hipMemset(bias, 0, 1024);
    run<0>("gemm_seg only (no barriers, no epilogue)", W, bias, out, blocks, layers);
    run<1>("+ 2 barriers per layer", W, bias, out, blocks, layers);
    run<3>("+ epilogue (bias, relu, LDS store)", W, bias, out, blocks, layers);
    run<7>("+ epilogue streams the layer output to HBM", W, bias, out, blocks, layers);
    run<8>("gemm_seg2 only (precomputed A addresses)", W, bias, out, blocks, layers);
    run<15>("gemm_seg2 + barriers + epilogue + HBM stream", W, bias, out, blocks, layers);
    run<0>("library gemm_seg PF=4", W, bias, out, blocks, layers);
    run<16>("library gemm_seg PF=2", W, bias, out, blocks, layers);
    run<16 + 7>("library gemm_seg PF=2 + barriers + epilogue + HBM", W, bias, out, blocks, layers);
    run<7>("library gemm_seg PF=4 + barriers + epilogue + HBM", W, bias, out, blocks, layers);
    run<16 + 7>("row-major stream-out (dword stores)", W, bias, out, blocks, layers);
    run<16 + 7 + 64>("fragment-order stream-out (dwordx4 stores)", W, bias, out, blocks, layers);
    run<16 + 7 + 128>("row-major load + store per element", W, bias, out, blocks, layers);
    run<16 + 7 + 128 + 64>("fragment-order load + store per element", W, bias, out, blocks, layers);
    run<16 + 7 + 32>("stream-out epilogue, odd wave slots start 1 x 3.4 us late", W, bias, out, blocks, layers, 1);
    run<16 + 7 + 32>("stream-out epilogue, odd wave slots start 2 x 3.4 us late", W, bias, out, blocks, layers, 2);
    run<16 + 7 + 32>("stream-out epilogue, odd wave slots start 4 x 3.4 us late", W, bias, out, blocks, layers, 4);
    run<16 + 7 + 128 + 64 + 32>("load + store epilogue, odd wave slots 1 x 3.4 us late", W, bias, out, blocks, layers, 1);
    run<16 + 7 + 128 + 64 + 32>("load + store epilogue, odd wave slots 2 x 3.4 us late", W, bias, out, blocks, layers, 2);
    run<16 + 7 + 128 + 64 + 32>("load + store epilogue, odd wave slots 4 x 3.4 us late", W, bias, out, blocks, layers, 4);
    run3<0>("first weight batch requested before the epilogue: stream-out", W, bias, out, blocks, layers);
    run3<128>("first weight batch requested before the epilogue: load+store", W, bias, out, blocks, layers);
    run32<0>("32-row tiles, 4 workgroups per CU: barriers + epilogue", W, bias, out, 2 * blocks, layers);
    run32<4>("32-row tiles, 4 workgroups per CU: + stream-out", W, bias, out, 2 * blocks, layers);
    run32<4 + 128>("32-row tiles, 4 workgroups per CU: load + store", W, bias, out, 2 * blocks, layers);
    run8<0>("8 waves per tile: gemm_seg only", W, bias, out, blocks, layers);
    run8<3>("8 waves per tile: + barriers + epilogue", W, bias, out, blocks, layers);
    run8<7>("8 waves per tile: + fragment-order stream-out", W, bias, out, blocks, layers);
    run8<7 + 128>("8 waves per tile: fragment-order load + store", W, bias, out, blocks, layers);
    run<3>("same, 512 blocks (one round)", W, bias, out, 512, layers);
    run<3>("same, 256 blocks (1 workgroup per CU)", W, bias, out, 256, layers);
    return 0;
}
