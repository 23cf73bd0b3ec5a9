// Small-batch variant of K2 (sdf(x + deform(x,t)), no grad): 16-point tiles on v_mfma_f32_16x16x4_f32.
// The secant iterations of ray marching (1024 points, 8 dependent launches) and the 8-sample up-sampling queries are
// latency-bound: a workgroup's time per layer is fixed by its tile height (a 32-row tile keeps one CU's matrix pipes busy
// for 16 384 cycles per 256x256 layer however many waves share it), so small batches want SHORT tiles and many
// workgroups.  16 rows x 256 columns per workgroup = 8 192 cycles per layer, 4x more workgroups than the 64-point kernel.
// Weights come from the 16x16x4 packing appended to the packed buffer (arch.h P16_SEGS).
#include "chain_common.h"
#include "launch.h"
#include "tabs.h"
#include "timing.h"

namespace es {

typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int T16 = 16;
constexpr int Q16_MAIN = HID * T16;          // 16 KiB activation tile [k][16 rows]
constexpr int Q16_AUX = 64 * T16;            // encodings (<= 64 rows incl. padding to the 16-k groups)
constexpr int Q16_LDS_BYTES = (Q16_MAIN + Q16_AUX + 1024) * 4;

// element (k, row) of a 16-row k-major tile; the XOR keeps ds_write_b128 epilogue stores and the A-fragment reads conflict-free
__device__ __forceinline__ int swz16(int k, int r) { return k * T16 + (r ^ (((k >> 1) & 3) << 2)); }

// acc[ni] += A[16 rows][0 .. 16*KG) * B[..][16 cols of n-tile nt0+ni], ni < 4 (this wave's 64 columns).
// A 1 024-point launch is one workgroup per CU = ONE wave per SIMD: nothing hides a weight load but distance.  The weights of a k-group
// (4 float4 per lane) are requested THREE groups (48 MFMAs, ~1 500 cycles) before their use through a ring of four register sets, and
// the first three groups of the NEXT layer before this layer's epilogue and barriers (``Wnext``; the caller passes ``primed`` to the
// next call).  With one group of distance every group waited ~100-400 cycles for L2 (57 stall cycles per 32-cycle MFMA).
struct Ring16 { float4 b[4][4]; };
// Where the weights of a segment come from: one 64-bit lane address per n-tile, centred on k-group 4 of an 8-group window, so that the
// eight groups of a window are IMMEDIATE offsets (-4 .. +3 KiB) and a layer costs 4 address adds per window (8 per 256-wide layer).
// Left to the compiler, every one of a layer's 64 loads got its own address from a scalar constant: 58 v_lshl_add_u64 + 50 v_readlane
// (the constants did not fit the SGPR file) + 29 hazard nops per layer, on the critical path of a one-wave-per-SIMD kernel.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(1))) float4* gptr4;      // global address space survives the opaque register pin (else: flat loads)
#else
typedef const float4* gptr4;                                         // (host pass of the same source: the kernels are never instantiated there)
#endif
struct WStream16 { gptr4 p[4]; };
template <int KG>
__device__ __forceinline__ WStream16 wstream16(const float4* __restrict__ W, int nt0, int lane) {
    WStream16 ws;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const float4* q = W + (size_t)((nt0 + ni) * KG + 4) * 64 + lane;
        asm volatile("" : "+v"(q));          // opaque: the loads below must use THIS register pair + an immediate
        ws.p[ni] = (gptr4)q;
    }
    return ws;
}
__device__ __forceinline__ void wstream16_next_window(WStream16& ws) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        gptr4 q = ws.p[ni] + 8 * 64;
        asm volatile("" : "+v"(q));
        ws.p[ni] = q;
    }
}
// k-group g of the stream's segment into ring slot g & 3 (g is a compile-time constant at every call site)
__device__ __forceinline__ void ring16_load(Ring16& R, const WStream16& ws, int g) {
    const int slot = g & 3;
#ifdef ES_Q16_NO_W          // dev probe: every k-group of every layer reads the same (L1-resident) weights
    const int gl = -4;
#else
    const int gl = (g & 7) - 4;
#endif
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) R.b[slot][ni] = ws.p[ni][gl * 64];
}
template <int KG>
__device__ __forceinline__ WStream16 ring16_prime(Ring16& R, const float4* __restrict__ W, int nt0, int lane) {
    WStream16 ws = wstream16<KG>(W, nt0, lane);
#pragma unroll
    for (int g = 0; g < 3 && g < KG; ++g) ring16_load(R, ws, g);
    return ws;
}
// ``ws`` = the stream of W when the previous call primed the ring (``primed``); on return it is the stream of Wnext if one was given
template <int KG, int KGN = 16>
__device__ __forceinline__ void gemm16(f32x4v (&acc)[4], const float* At, const float4* __restrict__ W, int nt0, int lane, Ring16& R, WStream16& ws,
                                       bool primed = false, const float4* __restrict__ Wnext = nullptr) {
    static_assert(KG <= 8 || KG == 16, "window bookkeeping below: one or two 8-group windows");
    const int lo = lane & 15, hi = lane >> 4;
    if (!primed) ws = ring16_prime<KG>(R, W, nt0, lane);
    WStream16 wn;
    if (Wnext != nullptr) wn = wstream16<KGN>(Wnext, nt0, lane);
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        if (g + 3 < KG) {
            if (g + 3 == 8) wstream16_next_window(ws);
            ring16_load(R, ws, g + 3);
        } else if (Wnext != nullptr) {
            ring16_load(R, wn, g + 3 - KG);      // the ring slot (g + 3) & 3 is free: KG % 4 == 0 here
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = At[swz16(16 * g + 4 * j + hi, lo)];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, f4c(R.b[g & 3][ni], j), acc[ni], 0, 0, 0);
        }
    }
    if (Wnext != nullptr) ws = wn;
}

// C/D layout of the 16x16 MFMA: col = lane&15, rows 4*(lane>>4) + reg: one quad of 4 consecutive rows per (lane, n-tile)
template <class F>
__device__ __forceinline__ void epi16(f32x4v (&acc)[4], float* mainT, int nt0, int lane, F&& f) {
    const int lo = lane & 15, rb = 4 * (lane >> 4);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int col = (nt0 + ni) * 16 + lo;
        float v[4] = {acc[ni][0], acc[ni][1], acc[ni][2], acc[ni][3]};
        f(col, rb, v, ni);
        *reinterpret_cast<float4*>(&mainT[swz16(col, rb)]) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

template <int NOUT>
__device__ __forceinline__ void smalln16(const float* At, const float* __restrict__ Wrows, int ldw, float* red, int tid) {
    // 256 threads: row = tid & 15, k-part = tid >> 4 (16 parts of 16 k)
    const int row = tid & 15, part = tid >> 4;
    float s[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) s[i] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int k = part * 16 + kk;
        const float a = At[swz16(k, row)];
#pragma unroll
        for (int i = 0; i < NOUT; ++i) s[i] = fmaf(Wrows[i * ldw + k], a, s[i]);
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) red[(part * NOUT + i) * 16 + row] = s[i];
}
template <int NOUT>
__device__ __forceinline__ float smalln16_reduce(const float* red, int i, int row) {
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) s += red[(p * NOUT + i) * 16 + row];
    return s;
}

template <int L>
__device__ __forceinline__ void encode3_16(float* At, int kbase, const float* px, int tid) {   // px[c*16 + row]
    const int row = tid & 15;
    for (int item = tid >> 4; item < 3 * L; item += 16) {
        const int c = item % 3, i = item / 3;
        float s, co;
        sincosf(px[c * 16 + row] * (float)(1 << i), &s, &co);
        At[swz16(kbase + enc_index(3, i, 0, c), row)] = s;
        At[swz16(kbase + enc_index(3, i, 1, c), row)] = co;
    }
    if (tid < 48) At[swz16(kbase + (tid >> 4), row)] = px[(tid >> 4) * 16 + row];
}

template <bool DEFORM>
__global__ __launch_bounds__(NTHREADS, 2) void k_query_sdf16(PointSrc src, Tabs tb, const float4* __restrict__ packed,
                                                             const float* __restrict__ weff, float* __restrict__ sdf_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* mainT = lds;
    float* aux = lds + Q16_MAIN;
    float* scr = aux + Q16_AUX;
    float* px = scr;          // [3][16]
    float* pt = scr + 48;     // [16]
    float* red = scr + 64;    // [16][<=3][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * T16;
    const int nt0 = 4 * wave;
    Ring16 ring;
    WStream16 wst;

    if (tid < 16) {
        float x[3], t, d[3];
        load_point(src, row0 + tid, x, t, d);
        px[tid] = x[0]; px[16 + tid] = x[1]; px[32 + tid] = x[2]; pt[tid] = t;
    }
    for (int i = tid; i < Q16_AUX; i += NTHREADS) aux[i] = 0.f;          // zero padding rows (k up to 64)
    __syncthreads();
    // biases are fetched BEFORE the layer's GEMM: an L2 round trip in the epilogue would sit on the critical path of every layer
    auto load_bias = [&](float(&bq)[4], const float* bias) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bq[ni] = bias[(nt0 + ni) * 16 + (lane & 15)];
    };
    auto relu_epi = [&](f32x4v(&acc)[4], const float(&bq)[4]) {
        epi16(acc, mainT, nt0, lane, [&](int col, int rb, float(&v)[4], int ni) {
            add_bias4(v, bq[ni]);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = relu1(v[i]);
        });
    };
    auto zero4 = [&](f32x4v(&acc)[4]) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = f32x4v{0.f, 0.f, 0.f, 0.f};
    };
    if (DEFORM) {
        encode3_16<6>(aux, 0, px, tid);
        {   // time encoding rows 39..51
            const int row = tid & 15;
            for (int i = tid >> 4; i < 6; i += 16) {
                float s, co;
                sincosf(pt[row] * (float)(1 << i), &s, &co);
                aux[swz16(39 + enc_index(1, i, 0, 0), row)] = s;
                aux[swz16(39 + enc_index(1, i, 1, 0), row)] = co;
            }
            if (tid < 16) aux[swz16(39, row)] = pt[row];
        }
        __syncthreads();
        {
            f32x4v acc[4];
            float bq[4];
            load_bias(bq, weff + tb.boff[NET_D * LAYERS + 0]);
            zero4(acc);
            gemm16<4>(acc, aux, packed + tb.p16off[0], nt0, lane, ring, wst, false, packed + tb.p16off[1]);
            relu_epi(acc, bq);
        }
        __syncthreads();
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            f32x4v acc[4];
            float bq[4];
            load_bias(bq, weff + tb.boff[NET_D * LAYERS + l]);
            zero4(acc);
            gemm16<16>(acc, mainT, packed + tb.p16off[l], nt0, lane, ring, wst, true, l < 7 ? packed + tb.p16off[l + 1] : nullptr);
            __syncthreads();
            epi16(acc, mainT, nt0, lane, [&](int col, int rb, float(&v)[4], int ni) {
                if (l == 3 && col >= 204) {                                  // IDR skip: [h(204) | enc(52)]
                    const float4 e = *reinterpret_cast<const float4*>(&aux[swz16(col - 204, rb)]);
                    v[0] = e.x; v[1] = e.y; v[2] = e.z; v[3] = e.w;
                } else {
                    add_bias4(v, bq[ni]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = relu1(v[i]);
                }
            });
            __syncthreads();
        }
        smalln16<3>(mainT, weff + tb.woff[NET_D * LAYERS + 8], 256, red, tid);
        __syncthreads();
        if (tid < 48) {
            const int i = tid >> 4, row = tid & 15;
            px[i * 16 + row] += smalln16_reduce<3>(red, i, row) + weff[tb.boff[NET_D * LAYERS + 8] + i];
        }
        __syncthreads();
        for (int i = tid; i < Q16_AUX; i += NTHREADS) aux[i] = 0.f;
        __syncthreads();
    }
    encode3_16<6>(aux, 0, px, tid);
    __syncthreads();
    auto sp_epi = [&](f32x4v(&acc)[4], const float(&bq)[4]) {
        epi16(acc, mainT, nt0, lane, [&](int col, int rb, float(&v)[4], int ni) {
            add_bias4(v, bq[ni]);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = softplus100(v[i]);
        });
    };
    {
        f32x4v acc[4];
        float bq[4];
        load_bias(bq, weff + tb.boff[NET_S * LAYERS + 0]);
        zero4(acc);
        gemm16<3>(acc, aux, packed + tb.p16off[8], nt0, lane, ring, wst);
        wst = ring16_prime<16>(ring, packed + tb.p16off[9], nt0, lane);
        sp_epi(acc, bq);
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        f32x4v acc[4];
        const int pi = l <= 4 ? 8 + l : 8 + l + 1;          // P16_SEGS order: SF0..SF3, SF4M, SF4A, SF5..SF7
        float bq[4];
        load_bias(bq, weff + tb.boff[NET_S * LAYERS + l]);
        zero4(acc);
        // the skip layer's extra columns use the ring in between: no priming across it
        const bool chain_next = l != 4 && l < 7;
        const int pn = l + 1 <= 4 ? 8 + l + 1 : 8 + l + 2;
        gemm16<16>(acc, mainT, packed + tb.p16off[pi], nt0, lane, ring, wst, l != 5, chain_next ? packed + tb.p16off[pn] : nullptr);
        if (l == 4) gemm16<3>(acc, aux, packed + tb.p16off[13], nt0, lane, ring, wst);
        __syncthreads();
        sp_epi(acc, bq);
        __syncthreads();
    }
    smalln16<1>(mainT, weff + tb.woff[NET_S * LAYERS + 8], 256, red, tid);
    __syncthreads();
    if (tid < 16 && row0 + tid < src.M) sdf_out[row0 + tid] = smalln16_reduce<1>(red, 0, tid) + weff[tb.boff[NET_S * LAYERS + 8]];
}

int query_sdf16(const PointSrc& src, const float* packed, const float* weff, float* sdf_out, int use_deform, hipStream_t st) {
    if (src.M <= 0) return ST_OK;
    const Tabs tb = make_tabs();
    const dim3 grid((src.M + T16 - 1) / T16), block(NTHREADS);
    const float4* pk = reinterpret_cast<const float4*>(packed);
    ScopedTimer tm(KID_QUERY16, src.M, st);
    // the launch asks for 56 KiB of LDS (the carve uses 24): at most TWO workgroups per CU.  At 168 registers three would fit, and a CU
    // that takes three 16-point tiles of an 8 192-point launch finishes half a tile time after the ones that took two
    constexpr int LDS_REQ = 56 * 1024;
    static_assert(LDS_REQ >= Q16_LDS_BYTES, "carve");
    if (use_deform) hipLaunchKernelGGL(k_query_sdf16<true>, grid, block, LDS_REQ, st, src, tb, pk, weff, sdf_out);
    else hipLaunchKernelGGL(k_query_sdf16<false>, grid, block, LDS_REQ, st, src, tb, pk, weff, sdf_out);
    return hip_last("query_sdf16");
}

}  // namespace es
