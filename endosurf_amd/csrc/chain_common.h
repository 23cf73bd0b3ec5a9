// Building blocks of the fused MLP "chain" kernels (gfx950 / CDNA4, wave64).
//
// Execution model of every chain kernel:
//   * one workgroup = 256 threads = 4 wavefronts (one per SIMD) owns a tile of TM = 64 rows;
//   * the activation tile lives in LDS, k-major ([k][row], XOR-swizzled) so that the MFMA A operand
//     (lane l -> row l&31, k-step l>>5) is a conflict-free ds_read_b32 and the epilogue writes 4
//     consecutive rows of one column as one conflict-free ds_write_b128;
//   * weights are pre-packed (pack.hip) into MFMA B-fragment order and streamed global->VGPR by the
//     wave that owns the output columns (each weight element is read once per workgroup, from L2),
//     software-prefetched two groups of 4 k-steps ahead; they never round-trip through LDS;
//   * the contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32, the fp32 matrix peak of gfx950);
//     each wave accumulates 64 rows x 64 columns in 4 accumulators of 16 VGPRs;
//   * the layer output is produced in-register, fused with bias/activation/tangent masking, written
//     back in place to the LDS tile (2 barriers per layer) and optionally streamed to HBM for backward.
#pragma once
#include <hip/hip_runtime.h>

#include "arch.h"

namespace es {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// max(z, 0) as ONE instruction: fmaxf() is two (the compiler canonicalises the MFMA result first: v_max z, z)
__device__ __forceinline__ float relu1(float z) {
    float r;
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(z));
    return r;
}

constexpr int TM = 64;                     // rows per workgroup tile
constexpr int NTHREADS = 256;              // 4 wavefronts
constexpr int MAIN_FLOATS = HID * TM;      // 64 KiB main activation tile
// LDS carve of the chain kernels: two workgroups per CU (<= 80 KiB each, <= 256 registers per lane):
// activation tile | 56-row auxiliary tile (encodings; reused as reduction scratch once consumed) | 1 KiB of per-row data
constexpr int AUX56_FLOATS = 56 * TM;
constexpr int LEAN_LDS_BYTES = (MAIN_FLOATS + AUX56_FLOATS + 256) * 4;   // 80 896 B

// element (k, row) of a k-major tile
__device__ __forceinline__ int swz(int k, int r) { return k * TM + (r ^ ((k & 15) << 2)); }

template <int RTC, int NTC>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[RTC][NTC]) {
#pragma unroll
    for (int i = 0; i < RTC; ++i)
#pragma unroll
        for (int j = 0; j < NTC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

__device__ __forceinline__ float f4c(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

// acc[ri][ni] += A[rows of row-tile rt0+ri][0 .. 8*KG) * B[0 .. 8*KG)[cols of n-tile nt0+ni]
// A = LDS tile ``At`` (k-major, swizzled); B = packed segment ``W`` ([nt][g][lane] float4, KG groups).
template <int KG, int RTC, int NTC, int PF = 2>   // PF = 2: 32 prefetch registers; PF = 4 gains ~2 % in isolation but spills in the bwd kernels
__device__ __forceinline__ void gemm_seg(f32x16 (&acc)[RTC][NTC], const float* At, const float4* __restrict__ W,
                                         int rt0, int nt0, int lane) {
    static_assert(PF == 2 || PF == 4, "prefetch groups: even-group parity of the swizzle table");
    constexpr bool GUARD = (KG % (2 * PF)) != 0;
    const int lo = lane & 31, hi = lane >> 5;
    float4 b0[PF][NTC], b1[PF][NTC];
    const float4* wl = W + lane;
    // A-operand addresses: element (k, r) with k = 8G + 2j + hi sits at 512G + 128j + 64hi + (r ^ (4hi) ^ (32(G&1) + 8j)).
    // The XOR takes 8 values (c = 4(G&1) + j): 8 per-lane registers per row tile, everything else is an immediate offset,
    // so the MFMA stream carries no address arithmetic (the run-time swizzle cost ~8 % of the loop; tools/micro/chain_micro).
    int aoff[RTC][8];
#pragma unroll
    for (int ri = 0; ri < RTC; ++ri)
#pragma unroll
        for (int c = 0; c < 8; ++c) aoff[ri][c] = ((((rt0 + ri) * 32 + lo) ^ (hi << 2)) ^ (8 * c)) + 64 * hi;

    auto loadB = [&](float4(&b)[PF][NTC], int g0) {
#pragma unroll
        for (int gi = 0; gi < PF; ++gi)
#pragma unroll
            for (int ni = 0; ni < NTC; ++ni)
                if (!GUARD || g0 + gi < KG) b[gi][ni] = wl[(size_t)((nt0 + ni) * KG + g0 + gi) * 64];
    };
    auto comp = [&](const float4(&b)[PF][NTC], int g0) {      // g0 is even at every call site
        const float* Ag = At + 512 * g0;
#pragma unroll
        for (int gi = 0; gi < PF; ++gi) {
            if (!GUARD || g0 + gi < KG) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a[RTC];
#pragma unroll
                    for (int ri = 0; ri < RTC; ++ri) a[ri] = Ag[aoff[ri][4 * (gi & 1) + j] + 512 * gi + 128 * j];
#pragma unroll
                    for (int ri = 0; ri < RTC; ++ri)
#pragma unroll
                        for (int ni = 0; ni < NTC; ++ni)
                            acc[ri][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ri], f4c(b[gi][ni], j), acc[ri][ni], 0, 0, 0);
                }
            }
        }
    };

    loadB(b0, 0);
    if constexpr (KG > 2 * PF) {
#pragma unroll 1
        for (int g0 = 0; g0 < KG; g0 += 2 * PF) {
            if (!GUARD || g0 + PF < KG) loadB(b1, g0 + PF);
            comp(b0, g0);
            if (g0 + 2 * PF < KG) loadB(b0, g0 + 2 * PF);
            if (!GUARD || g0 + PF < KG) comp(b1, g0 + PF);
        }
    } else {
        if (PF < KG) loadB(b1, PF);
        comp(b0, 0);
        if (PF < KG) comp(b1, PF);
    }
}

// Visit the accumulator as quads: f(row, col, v[4]) where v holds rows row..row+3 of column col
// (C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).
template <int RTC, int NTC, class F>
__device__ __forceinline__ void for_quads(f32x16 (&acc)[RTC][NTC], int rt0, int nt0, int lane, F&& f) {
    const int lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ri = 0; ri < RTC; ++ri)
#pragma unroll
        for (int ni = 0; ni < NTC; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4] = {acc[ri][ni][4 * q + 0], acc[ri][ni][4 * q + 1], acc[ri][ni][4 * q + 2], acc[ri][ni][4 * q + 3]};
                f((rt0 + ri) * 32 + 8 * q + 4 * hi, (nt0 + ni) * 32 + lo, v);
            }
}

// Same visiting order without an accumulator (element-wise stages that run "in epilogue layout").
template <int RTC, int NTC, class F>
__device__ __forceinline__ void for_quads_noacc(int rt0, int nt0, int lane, F&& f) {
    const int lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ri = 0; ri < RTC; ++ri)
#pragma unroll
        for (int ni = 0; ni < NTC; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) f((rt0 + ri) * 32 + 8 * q + 4 * hi, (nt0 + ni) * 32 + lo);
}

// Same as for_quads, additionally passing the linear quad index qi = (ri*NTC + ni)*4 + q (a compile-time constant after
// unrolling) so that epilogues can consume operands prefetched into registers BEFORE the GEMM (hides HBM latency).
template <int RTC, int NTC, class F>
__device__ __forceinline__ void for_quads_qi(f32x16 (&acc)[RTC][NTC], int rt0, int nt0, int lane, F&& f) {
    const int lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ri = 0; ri < RTC; ++ri)
#pragma unroll
        for (int ni = 0; ni < NTC; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4] = {acc[ri][ni][4 * q + 0], acc[ri][ni][4 * q + 1], acc[ri][ni][4 * q + 2], acc[ri][ni][4 * q + 3]};
                f((rt0 + ri) * 32 + 8 * q + 4 * hi, (nt0 + ni) * 32 + lo, v, (ri * NTC + ni) * 4 + q);
            }
}
// The 8 quads (ri, q) of ONE n-tile column block ni of a 64 x 64 wave tile: batch index b8 = ri*4 + q.  Epilogues that read
// saved activations issue all loads of a half (prefetch_half_f: 8 quads per operand = 32 registers) before they touch the first
// value, so an epilogue pays two memory round trips in total instead of one per compiler-chosen group of quads.
template <int NI, class F>
__device__ __forceinline__ void for_quads_half(f32x16 (&acc)[2][2], int nt0, int lane, F&& f) {      // f(row, col, v, b8)
    const int lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4] = {acc[ri][NI][4 * q + 0], acc[ri][NI][4 * q + 1], acc[ri][NI][4 * q + 2], acc[ri][NI][4 * q + 3]};
            f(ri * 32 + 8 * q + 4 * hi, (nt0 + NI) * 32 + lo, v, ri * 4 + q);
        }
}

// v[0..3] += b as two packed adds (v_pk_add_f32): the same roundings as four scalar adds
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void add_bias4(float (&v)[4], float b) {
    const f32x2v bb = {b, b};
    f32x2v lo = {v[0], v[1]}, hi = {v[2], v[3]};
    lo += bb; hi += bb;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
}
// v if bit ``k`` of ``w`` is set, else +0: signed 1-bit field extract (0 or all ones) + and = 2 VALU instructions
// (the compare + select form is 3; the result differs from it only in the sign of a zero)
__device__ __forceinline__ float keep_if_bit(float v, unsigned w, int k) {
    const int m = __builtin_amdgcn_sbfe((int)w, k, 1);
    return __uint_as_float(__float_as_uint(v) & (unsigned)m);
}
// ---- epilogue without address arithmetic or a bias add -------------------------------------------------------------------------
// Every non-MFMA instruction of a wave adds to its MFMA time on this part (measured: softplus on the raw exp / log units, -13 VALU
// instructions per element, took 3 % off k_query_sdf), so the epilogue of the hot kernels carries none it can avoid:
//  * the LDS address of a lane's quad (row tile ri, quad q) of n-tile ni is  o[ri*4 + q] + 2048 ni  floats: the XOR swizzle takes
//    RTC*4 values per lane (they do not depend on the layer), the n-tile is an immediate offset -> RTC*4 pinned registers instead of
//    ~7 VALU instructions per ds_write_b128;
//  * the bias is requested a layer ahead and added as packed pairs (add_bias4).  (As the accumulators' INITIAL value it would cost
//    nothing at all, but the no-grad queries feed DISCRETE decisions, the first
//    sign change of ray marching: a different rounding order moves a proposal that sits within 1e-7 of the surface to its other side,
//    and the golden surface-neighbour case (tests/test_gpu_render.py::test_aux_forward) holds such a ray.)
template <int RTC> struct QuadOff { int o[RTC * 4]; };
template <int RTC>
__device__ __forceinline__ QuadOff<RTC> quad_offsets(int rt0, int nt0, int lane) {
    QuadOff<RTC> qo;
    const int lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ri = 0; ri < RTC; ++ri)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int v = swz(nt0 * 32 + lo, (rt0 + ri) * 32 + 8 * q + 4 * hi);
            asm volatile("" : "+v"(v));          // opaque: keep the value in its register instead of re-deriving it at every store
            qo.o[ri * 4 + q] = v;
        }
    return qo;
}
// f(row, col, v[4], off, ni): off = LDS float offset of the quad (see QuadOff), ni = n-tile index within the wave tile
template <int RTC, int NTC, class F>
__device__ __forceinline__ void for_quads_off(f32x16 (&acc)[RTC][NTC], const QuadOff<RTC>& qo, int rt0, int nt0, int lane, F&& f) {
    const int lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ri = 0; ri < RTC; ++ri)
#pragma unroll
        for (int ni = 0; ni < NTC; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4] = {acc[ri][ni][4 * q + 0], acc[ri][ni][4 * q + 1], acc[ri][ni][4 * q + 2], acc[ri][ni][4 * q + 3]};
                f((rt0 + ri) * 32 + 8 * q + 4 * hi, (nt0 + ni) * 32 + lo, v, qo.o[ri * 4 + q] + 2048 * ni, ni);
            }
}
__device__ __forceinline__ void lds_store_quad_at(float* At, int off, const float (&v)[4]) {
    *reinterpret_cast<float4*>(At + off) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void lds_store_quad(float* At, int col, int row, const float (&v)[4]) {
    *reinterpret_cast<float4*>(&At[swz(col, row)]) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void lds_load_quad(const float* At, int col, int row, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(&At[swz(col, row)]);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void lds_add_quad(float* At, int col, int row, const float (&v)[4]) {
    float4* p = reinterpret_cast<float4*>(&At[swz(col, row)]);
    float4 t = *p;
    t.x += v[0]; t.y += v[1]; t.z += v[2]; t.w += v[3];
    *p = t;
}
// rows row..row+3 of column col of a row-major [rows][ld] HBM buffer (each wave store = 2 x 128 B lines)
__device__ __forceinline__ void g_store_quad(float* __restrict__ base, size_t grow0, int ld, int row, int col, const float (&v)[4]) {
    float* p = base + (grow0 + row) * (size_t)ld + col;
    // streamed once (read back by a later kernel): non-temporal, so the activation stream does not evict the weights from L2
    __builtin_nontemporal_store(v[0], p); __builtin_nontemporal_store(v[1], p + ld);
    __builtin_nontemporal_store(v[2], p + 2 * ld); __builtin_nontemporal_store(v[3], p + 3 * ld);
}
__device__ __forceinline__ void g_load_quad(const float* __restrict__ base, size_t grow0, int ld, int row, int col, float (&v)[4]) {
    const float* p = base + (grow0 + row) * (size_t)ld + col;
    v[0] = __builtin_nontemporal_load(p); v[1] = __builtin_nontemporal_load(p + ld);
    v[2] = __builtin_nontemporal_load(p + 2 * ld); v[3] = __builtin_nontemporal_load(p + 3 * ld);
}

// ---- fragment-ordered activation stacks --------------------------------------------------------------------------------
// A [64 rows][256 columns] tile of a saved activation stack stored in accumulator-fragment order: the quad (4 consecutive rows
// of one column) that a lane holds after the MFMA is ONE float4, and the 64 lanes of a wave access 1 KiB contiguously
// (index ((wave * 16 + (ri * 2 + ni) * 4 + q) * 64 + lane) * 4, rows 32 ri + 8 q + 4 hi .., column 64 wave + 32 ni + lo).
typedef float v4f_frag __attribute__((ext_vector_type(4)));
__device__ __forceinline__ size_t frag_off(size_t grow0, int row, int col) {
    const int ri = row >> 5, q = (row >> 3) & 3, hi = (row >> 2) & 1;
    const int w = col >> 6, ni = (col >> 5) & 1, lo = col & 31;
    return grow0 * 256 + (size_t)(((w * 16 + (ri * 2 + ni) * 4 + q) * 64 + hi * 32 + lo) * 4);
}
__device__ __forceinline__ void g_store_quad_f(float* __restrict__ base, size_t grow0, int row, int col, const float (&v)[4]) {
    const v4f_frag t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f_frag*>(base + frag_off(grow0, row, col)));
}
__device__ __forceinline__ void g_load_quad_f(const float* __restrict__ base, size_t grow0, int row, int col, float (&v)[4]) {
    const v4f_frag t = __builtin_nontemporal_load(reinterpret_cast<const v4f_frag*>(base + frag_off(grow0, row, col)));
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <int RTC, int NTC>
__device__ __forceinline__ void prefetch_quads_f(float (&buf)[RTC * NTC * 4][4], const float* __restrict__ base, size_t grow0, int rt0, int nt0,
                                                 int lane) {
    const int lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ri = 0; ri < RTC; ++ri)
#pragma unroll
        for (int ni = 0; ni < NTC; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                g_load_quad_f(base, grow0, (rt0 + ri) * 32 + 8 * q + 4 * hi, (nt0 + ni) * 32 + lo, buf[(ri * NTC + ni) * 4 + q]);
}
template <int NI>
__device__ __forceinline__ void prefetch_half_f(float (&buf)[8][4], const float* __restrict__ base, size_t grow0, int nt0, int lane) {
    const int lo = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int q = 0; q < 4; ++q) g_load_quad_f(base, grow0, ri * 32 + 8 * q + 4 * hi, (nt0 + NI) * 32 + lo, buf[ri * 4 + q]);
}

// Load a [64][NC] row-major HBM tile into rows 0..NC-1 of the k-major LDS tile (rows grow0.. of ``src``, leading dim ld).
template <int NC>
__device__ __forceinline__ void load_tile(float* At, const float* __restrict__ src, size_t grow0, int ld, int tid) {
    const int r = tid >> 2, c4 = tid & 3;          // 4 threads per row, NC/16 float4 each
    const float* p = src + (grow0 + r) * (size_t)ld;
#pragma unroll 4
    for (int jj = 0; jj < NC / 16; ++jj) {
        const int c = 4 * (c4 + 4 * jj);
        const float4 v = *reinterpret_cast<const float4*>(p + c);
        At[swz(c + 0, r)] = v.x; At[swz(c + 1, r)] = v.y; At[swz(c + 2, r)] = v.z; At[swz(c + 3, r)] = v.w;
    }
}
__device__ __forceinline__ void load_tile_256(float* At, const float* __restrict__ src, size_t grow0, int ld, int tid) {
    load_tile<256>(At, src, grow0, ld, tid);
}

// ---- ReLU masks of the deformation network as bit words ------------------------------------------------
// deform_fwd (tile = 32 points, rows [value, tangent] per point) writes one uint32 per thread and layer: bit 2*qi + v is the
// mask of value row v (quad rows 0 / 2) of quad qi = (ri*2 + ni)*4 + q.  A 64-point tile of the one-row-per-point sweeps
// (VJP, tangent) covers two such tiles; its lane (lo, hi) needs, per layer, the words of producer lanes (lo, 0) and (lo, 1)
// of both: point 32*ri_c + 8*q_c + 4*hi + i is producer row 2*(8*q_c + 4*hi + i) = quad (q_c>>1, ., 2*(q_c&1) + hi) of
// producer lane hi' = i>>1, value row i&1.
struct MaskWords { unsigned w[2][2]; };      // [ri_c][producer hi]
__device__ __forceinline__ MaskWords load_mask_words(const unsigned* __restrict__ Ml, int tile64, int wave, int lane) {
    MaskWords m;
    const unsigned* p = Ml + (size_t)(2 * tile64) * 256 + wave * 64 + (lane & 31);
    m.w[0][0] = p[0]; m.w[0][1] = p[32]; m.w[1][0] = p[256]; m.w[1][1] = p[256 + 32];
    return m;
}
// mask of element i of consumer quad qi = (ri_c*2 + ni)*4 + q_c for a lane with hi = lane>>5
// v if the mask of element i of consumer quad qi is set, else +0 (keep_if_bit on the producer word that mask_bit reads)
__device__ __forceinline__ float mask_keep(float v, const MaskWords& m, int qi, int i, int hi) {
    const int ri_c = qi >> 3, ni = (qi >> 2) & 1, q_c = qi & 3;
    const int pq = ((q_c >> 1) * 2 + ni) * 4 + 2 * (q_c & 1) + hi;
    return keep_if_bit(v, m.w[ri_c][i >> 1], 2 * pq + (i & 1));
}
__device__ __forceinline__ bool mask_bit(const MaskWords& m, int qi, int i, int hi) {
    const int ri_c = qi >> 3, ni = (qi >> 2) & 1, q_c = qi & 3;
    const int pq = ((q_c >> 1) * 2 + ni) * 4 + 2 * (q_c & 1) + hi;
    return (m.w[ri_c][i >> 1] >> (2 * pq + (i & 1))) & 1u;
}

// ---- activations ---------------------------------------------------------------------------------
// nn.Softplus(beta=100, threshold=20)  (reference endosurf.py:771):  z if 100z > 20 else log1p(exp(100z))/100.
// Evaluated branch-free as max(z,0) + log(1 + exp(-|100z|))/100 on the hardware exp/log units: identical above
// the threshold to fp32 rounding, absolute error <= ~1e-9 below it (the accurate libm chain costs ~130 VALU
// instructions per element, which would rival the MFMA time of a 256x256 layer).
__device__ __forceinline__ float softplus100(float z) {
    // raw v_exp_f32 / v_log_f32 (base 2): the logarithm's argument is in (1, 2], so the range handling __logf carries (compare, ldexp,
    // select, a 4-instruction correction: 13 of its 20 VALU instructions) has nothing to do; 64 elements per lane and layer
    const float e = __builtin_amdgcn_exp2f(-144.26950408889634f * fabsf(z));
    return fmaf(0.006931471805599453f, __builtin_amdgcn_logf(1.f + e), relu1(z));
}
// softplus'(z) = sigmoid(100 z) recovered from s = softplus(z):  1 - exp(-100 s)   (series where that cancels)
__device__ __forceinline__ float softplus100_grad_from_s(float s) {
    const float x = 100.f * s;
    return x < 0.02f ? x * (1.f - x * (0.5f - x * (1.f / 6.f))) : 1.f - __expf(-x);
}

// The same function on a quad, as PACKED fp32 arithmetic (v_pk_mul / v_pk_fma / v_pk_add: two elements per VALU instruction; same
// operations in the same order as the scalar form, so the values are identical): 5 VALU instructions per element instead of 8.
typedef float f32x2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2p softplus100_grad_from_s2(f32x2p s) {
    const f32x2p x = s * 100.f;
    const f32x2p t = x * -1.4426950408889634f;
    f32x2p e;
    e[0] = __builtin_amdgcn_exp2f(t[0]); e[1] = __builtin_amdgcn_exp2f(t[1]);
    const f32x2p big = 1.f - e;
    const f32x2p ser = x * (1.f - x * (0.5f - x * (1.f / 6.f)));
    f32x2p r;
    r[0] = x[0] < 0.02f ? ser[0] : big[0];
    r[1] = x[1] < 0.02f ? ser[1] : big[1];
    return r;
}
__device__ __forceinline__ void softplus100_grad_from_s4(const float (&s)[4], float (&d)[4]) {
    const f32x2p a = softplus100_grad_from_s2(f32x2p{s[0], s[1]}), b = softplus100_grad_from_s2(f32x2p{s[2], s[3]});
    d[0] = a[0]; d[1] = a[1]; d[2] = b[0]; d[3] = b[1];
}

// ---- frequency encoding (reference src/renderer/encoder.py:40-54) ----------------------------------
// element ``idx`` of [x, sin(2^0 x), cos(2^0 x), ...] for a D-dim input with value x_c of its coordinate:
// layout index = D + (2*i + fn)*D + c  for frequency i, fn (0 sin / 1 cos), coordinate c.
__device__ __forceinline__ int enc_index(int D, int i, int fn, int c) { return D + (2 * i + fn) * D + c; }

// out[NOUT][row] = sum_k Wrows[i][k] * At[k][row] (+ bias handled by caller): VALU path for tiny-N layers.
// Partial sums of the 4 waves go to scr[(part*NOUT + i)*64 + row]; caller barriers and reduces.
template <int NOUT>
__device__ __forceinline__ void smalln_partial(const float* At, const float* __restrict__ Wrows, int ldw, float* scr, int tid) {
    const int row = tid & 63;
    const int part = __builtin_amdgcn_readfirstlane(tid >> 6);
    float s[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) s[i] = 0.f;
    const int k0 = part * 64;
#pragma unroll 8
    for (int kk = 0; kk < 64; ++kk) {
        const float a = At[swz(k0 + kk, row)];
#pragma unroll
        for (int i = 0; i < NOUT; ++i) s[i] = fmaf(Wrows[i * ldw + k0 + kk], a, s[i]);
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) scr[(part * NOUT + i) * 64 + row] = s[i];
}
template <int NOUT>
__device__ __forceinline__ float smalln_reduce(const float* scr, int i, int row) {
    return (scr[(0 * NOUT + i) * 64 + row] + scr[(1 * NOUT + i) * 64 + row]) + (scr[(2 * NOUT + i) * 64 + row] + scr[(3 * NOUT + i) * 64 + row]);
}

// ---- point source: where a kernel gets its query points from -----------------------------------------
// mode 0: explicit x[M][3], t[M] (or t[0] if t_scalar);  optional dirs[M][3]
// mode 1: ray samples: point i -> ray i / n, sample i % n:  x = o + d/(d.z+1e-6) * z[ray*ldz + s], t = rays[ray][8],
//         dir = rays[ray][3:6]   (reference endosurf.py:66, 87, 153)
// mode 2: the first M_split points are ray samples (as mode 1), the remaining M - M_split are explicit (x, t) without dirs:
//         lets the training step evaluate its auxiliary points (errorondepth / surface neighbours) in the render launch
struct PointSrc {
    const float* x;
    const float* t;
    const float* dirs;
    const float* rays;
    const float* z;
    int mode, t_scalar, n_per_ray, ldz;
    int M, M_split;
};
__device__ __forceinline__ void load_point(const PointSrc& s, int i_in, float (&x)[3], float& t, float (&d)[3]) {
    int i = i_in;
    if (i >= s.M) { x[0] = x[1] = x[2] = 0.f; t = 0.f; d[0] = d[1] = 0.f; d[2] = 1.f; return; }
    if (s.mode == 0 || (s.mode == 2 && i >= s.M_split)) {
        if (s.mode == 2) i -= s.M_split;
        x[0] = s.x[3 * (size_t)i]; x[1] = s.x[3 * (size_t)i + 1]; x[2] = s.x[3 * (size_t)i + 2];
        t = s.t[s.t_scalar ? 0 : i];
        if (s.dirs) { d[0] = s.dirs[3 * (size_t)i]; d[1] = s.dirs[3 * (size_t)i + 1]; d[2] = s.dirs[3 * (size_t)i + 2]; }
        else { d[0] = d[1] = 0.f; d[2] = 1.f; }
    } else {
        const int ray = i / s.n_per_ray, smp = i - ray * s.n_per_ray;
        const float* r = s.rays + 9 * (size_t)ray;
        const float zz = s.z[(size_t)ray * s.ldz + smp];
        const float inv = r[5] + 1e-6f;
        d[0] = r[3]; d[1] = r[4]; d[2] = r[5];
        x[0] = r[0] + (r[3] / inv) * zz; x[1] = r[1] + (r[4] / inv) * zz; x[2] = r[2] + (r[5] / inv) * zz;
        t = r[8];
    }
}

}  // namespace es
