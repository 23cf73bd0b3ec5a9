// Core of the register-resident split-precision kernels (query_x3r.hip, infer_x3r.hip): the k-step-ordered weight stream, the
// operand fragments and the GEMM of one layer.  See query_x3r.hip for the formulation.
#pragma once
#include "chain_common.h"
#include "x3_common.h"

namespace es {

// segments in the order the kernels walk them: [0, 8) deformation forward, [8, 17) SDF forward (query_x3r.hip, infer_x3r.hip value
// passes), [17, 25) deformation reverse DR7 .. DR0 (the VJP sweep of infer_x3r.hip), [25, 35) the geometry-feature layer and the SDF
// reverse sweep (the encoding part of the skip layer's adjoint before its hidden part: both read the same operand), [35, 46) colour
constexpr int XR_SEGS[] = {DF0, DF1, DF2, DF3, DF4, DF5, DF6, DF7, SF0, SF1, SF2, SF3, SF4M, SF4A, SF5, SF6, SF7,
                           DR7, DR6, DR5, DR4, DR3, DR2, DR1, DR0,
                           SF8F, SR7, SR6, SR5, SR4A, SR4M, SR3, SR2, SR1, SR0,
                           CF0S, CF0F, CF1, CF2, CF3, CF4H, CF4S, CF4F, CF5, CF6, CF7,
                           // [46, 57) colour reverse sweep (train_x3r.hip): the three parts of the skip layer's input adjoint read the same operand
                           CR7, CR6, CR5, CR4F, CR4S, CR4H, CR3, CR2, CR1, CR0F, CR0S};
constexpr int XR_COUNT = sizeof(XR_SEGS) / sizeof(int);
constexpr int xr_kg(int i) { return 2 * cdiv(SEGS[XR_SEGS[i]].kreal, 32); }      // k-steps, even (the stream works in pairs)
constexpr int xr_chunk0(int i) {
    int c = 0;
    for (int k = 0; k < i; ++k) c += xr_kg(k);
    return c;
}
constexpr int XR_CHUNKS = xr_chunk0(XR_COUNT);
constexpr int XR_PAD_CHUNKS = 4;                          // a stream reads up to 3 k-steps past its end
constexpr int XR_SDF_CHUNK0 = xr_chunk0(8);               // first k-step of SF0
constexpr int XR_QUERY_CHUNKS = xr_chunk0(17);            // 236: end of the query stream
constexpr int XR_DR_CHUNK0 = xr_chunk0(17);               // first k-step of DR7
constexpr int XR_SI_CHUNK0 = xr_chunk0(25);               // first k-step of SF8F (then SR7 ...)
constexpr int XR_C_CHUNK0 = xr_chunk0(35);                // first k-step of CF0S
constexpr int XR_CR_CHUNK0 = xr_chunk0(46);               // first k-step of CR7
constexpr int XR_SDF_FWD_CHUNKS = XR_QUERY_CHUNKS - XR_SDF_CHUNK0;      // 120 k-steps of SF0 .. SF7
constexpr int XR_CHUNK_UNITS = 8 * 3;                     // 1 KB units (64 lanes x 16 B) per k-step
constexpr int XR_CHUNK_BYTES = XR_CHUNK_UNITS * 1024;
constexpr int XR_THREADS = 256;
constexpr int XR_ENC_LD = 68;                             // floats per column row of the encoding scratch (conflict-free b32 / b128 reads)
constexpr int XR_RING = 4;                                // k-steps resident in LDS
static_assert(XR_SDF_CHUNK0 % 2 == 0 && XR_DR_CHUNK0 % 2 == 0 && XR_SI_CHUNK0 % 2 == 0 && XR_C_CHUNK0 % 2 == 0 && XR_CHUNKS % 2 == 0, "k-step pairs");
static_assert(XR_SEGS[46] == CR7 && XR_SEGS[56] == CR0S && XR_COUNT == 57, "segment order the training kernels hard-code");
static_assert(LAYER_N[NET_D][3] == 204 && LAYER_N[NET_S][7] == 256 && SEGS[SF4A].kreal == 39, "shapes the kernels hard-code");

// position j (0..7) of lane half hi in k-step-local order -> k offset inside the 16-wide step
__host__ __device__ constexpr int xr_kperm(int hi, int j) { return j < 4 ? 4 * hi + j : 8 + 4 * hi + (j - 4); }

// ---- the weight stream -----------------------------------------------------------------------------------------------------------
// K-step k lives in ring slot k % 4; k-steps travel in pairs (2c, 2c+1).  Every wave moves 6 of the 24 1-KB units of a k-step by
// direct loads.  The timeline, with each k-step split into its two accumulator groups:
//     k-step 2c,   group 0:  read the fragments of (2c, group 1);    MFMAs, with the 6 pieces of k-step 2c+2 issued between them
//     k-step 2c,   group 1:  read the fragments of (2c+1, group 0);  MFMAs, with the 6 pieces of k-step 2c+3 issued between them
//     k-step 2c+1, group 0:  read the fragments of (2c+1, group 1);  MFMAs;  all pieces landed (vmcnt 0), BARRIER
//     k-step 2c+1, group 1:  read the fragments of (2c+2, group 0);  MFMAs
// A pair's slots are rewritten one barrier after their last read and read one barrier after they landed; the fragments needed right
// after a barrier are requested before the MFMAs that follow it.  A bare s_barrier behind an explicit vmcnt(0): a kernel's own loads
// and stores between two barriers (side streams, saved stacks, masks) are waited for as well, so they are issued early in a pair.
struct FragA { u32x4 p[4][3]; };
struct WStream {
    const u32x4* g;          // k-step 0 of the packed buffer
    unsigned char* ring;
    int k, wave, lane;       // k = the (logical) k-step being computed
    // logical -> packed k-step: a kernel walks up to three ranges of the packed order back to back: [.., e0) -> b0 + k,
    // [e0, e1) -> b1 + (k - e0), [e1, ..) -> b2 + (k - e1)
    int e0 = 0x7fffffff, e1 = 0x7fffffff, b0 = 0, b1 = 0, b2 = 0;
    FragA a0;                // fragments of (k, group 0), read ahead
    __device__ __forceinline__ int packed_step(int kk) const { return kk < e0 ? b0 + kk : (kk < e1 ? b1 + (kk - e0) : b2 + (kk - e1)); }
    __device__ __forceinline__ void piece(int kk, int i) {
#ifndef XR_NO_DMA
        const u32x4* src = g + ((size_t)packed_step(kk) * XR_CHUNK_UNITS + wave * 6 + i) * 64 + lane;
        unsigned char* dst = ring + (kk & (XR_RING - 1)) * XR_CHUNK_BYTES + (wave * 6 + i) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
#endif
    }
    // nst: vector-memory STORES the kernel issued after the last load of the pair that is landing (the sink of the even k-step: it runs
    // after that k-step's staging slots, and the odd k-step issues nothing before its barrier).  vmcnt counts loads and stores of gfx9 in
    // issue order, so vmcnt(nst) waits for every load while the youngest stores keep draining over the next k-step instead of being
    // acknowledged within half of one.
    int nst = 0;
    __device__ __forceinline__ void landed_barrier() {
        if (nst == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else if (nst == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifndef XR_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
    }
    __device__ __forceinline__ void read_group(FragA& a, int kk, int grp) const {
        const u32x4* A = reinterpret_cast<const u32x4*>(ring + (kk & (XR_RING - 1)) * XR_CHUNK_BYTES) + lane;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int p = 0; p < 3; ++p) a.p[f][p] = A[((4 * grp + f) * 3 + p) * 64];
    }
    __device__ __forceinline__ void start() {
#pragma unroll
        for (int i = 0; i < 6; ++i) { piece(k, i); piece(k + 1, i); }
        landed_barrier();
        read_group(a0, k, 0);
    }
};

#ifdef XR_PROFILE        // dev builds only (tools/dev/xr_profile.sh): cycle stamps of block 0 / wave 0
extern __device__ long long xr_prof[512];
#define XR_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) xr_prof[i] = __builtin_readcyclecounter(); } while (0)
#define XR_ADD(i, v) do { if (blockIdx.x == 0 && threadIdx.x == 0) xr_prof[i] += (v); } while (0)
#else
#define XR_STAMP(i) do {} while (0)
#define XR_ADD(i, v) do {} while (0)
#endif

struct FragB { u32x4 h, m, l; };
// SINK: sink(s, v) receives the 8 fp32 operand elements of k-step s (element j <-> k offset xr_kperm(hi, j)) once, right after they
// were built -- the training kernels stream them out as the saved layer inputs / adjoints of the weight-gradient GEMMs
struct NoSink { __device__ __forceinline__ void operator()(int, const float (&)[8]) const {} };
template <class VAL, class SINK>
__device__ __forceinline__ void build_frag(FragB& b, VAL&& val, int s, SINK&& sink) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned hh, mm, ll;
        v[2 * j] = val(s, 2 * j); v[2 * j + 1] = val(s, 2 * j + 1);
        split_pair(v[2 * j], v[2 * j + 1], hh, mm, ll);
        b.h[j] = hh; b.m[j] = mm; b.l[j] = ll;
    }
    sink(s, v);
}

// C[fb] += W[32 fb .. +31][k-steps 0 .. KG) . B, with B's k-step s operand = split(val(s, 0..7)); the operand of k-step s+1 is built
// between the MFMAs of k-step s.  MFMA order: term-major over groups of 4 accumulators -- consecutive MFMAs never share an
// accumulator (an instruction issued between two MFMAs of one accumulate chain costs ~40 cycles, between independent ones ~6), the
// six partial products of an accumulator still arrive smallest first.  KG is even and ws.k is even on entry.
struct NoSide { __device__ __forceinline__ void operator()(int, int, int) const {} };
// LATE (values that depend on data landing at this k-step's barrier): all 8 operand elements are built in group 1, two per slot;
// otherwise 8 of the 12 (group, term) slots build one element each.  STAGE: this k-step issues the direct loads of the next pair.
// side(kk, t, sloc): slot t of the k-step that stages the loads of k-step kk (logical) = k-step sloc of the running GEMM (compile-time after
// unrolling; sloc >= KG: the first k-steps of whatever follows it).  NF: feature blocks of the group that are computed (4, or 2 with NG = 0).
template <int G, bool MFMA, bool LATE, bool STAGE, int NF, class VAL, class SIDE>
__device__ __forceinline__ void mfma_group(f32x16 (&C)[8], WStream& ws, const FragA& a, const FragB& b, FragB& nb, float (&v)[8], VAL&& val, int snext,
                                           bool more, SIDE&& side) {
    // smallest terms first: (l,h) (m,m) (h,l) | (m,h) (h,m) | (h,h)
    constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const u32x4 bt = TB[t] == 0 ? b.h : (TB[t] == 1 ? b.m : b.l);
#ifndef XR_NO_MFMA
        if (MFMA)
#pragma unroll
            for (int f = 0; f < NF; ++f)
                C[4 * G + f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.p[f][TA[t]]), __builtin_bit_cast(bf16x8, bt),
                                                                      C[4 * G + f], 0, 0, 0);
#endif
        if (STAGE) {
            ws.piece(ws.k + 2 + G, t);                     // one direct load per four MFMAs
            side(ws.k + 2 + G, t, snext + 1 + G);          // a kernel's own per-k-step operand stream (same slots, same barriers)
        }
#ifndef XR_NO_VALU
        if (more && t < 4) {
            constexpr int NK = LATE ? (G == 1 ? 2 : 0) : 1;
            const int k0 = LATE ? 2 * t : 4 * G + t;
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
                const int k = k0 + kk;
                v[k] = val(snext, k);
                if (k & 1) {
                    unsigned hh, mm, ll;
                    split_pair(v[k - 1], v[k], hh, mm, ll);
                    nb.h[k >> 1] = hh; nb.m[k >> 1] = mm; nb.l[k >> 1] = ll;
                }
            }
        }
#endif
#ifndef XR_NO_PIN
        // keep each term's fillers next to its four MFMAs: left alone, the scheduler gathers the MFMAs into long runs and the VALU work
        // into blocks of ~40 instructions between two of them (measured: 36.3 k -> 33.7 k cycles per 256-wide layer; one barrier per MFMA
        // with the fillers dealt out by hand: 36.5 k)
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}
template <int NG, bool LATE, bool ODD, class VAL, class SIDE, class SINK>
__device__ __forceinline__ void kstep_r(f32x16 (&C)[8], WStream& ws, FragB& b, VAL&& val, SIDE&& side, SINK&& sink, int snext, bool more) {
    FragB nb = b;
    float v[8];
    FragA a1;
    ws.read_group(a1, ws.k, 1);
    mfma_group<0, true, LATE && ODD, !ODD, (NG == 0 ? 2 : 4)>(C, ws, ws.a0, b, nb, v, val, snext, more, side);
    if (ODD) ws.landed_barrier();
    ws.read_group(ws.a0, ws.k + 1, 0);
    mfma_group<1, NG == 2, LATE && ODD, !ODD, 4>(C, ws, a1, b, nb, v, val, snext, more, side);
    if (more) sink(snext, v);
    b = nb;
    ++ws.k;
}
// NG = 1: only accumulator group 0 (features 0 .. 127) is computed, NG = 0: only its first two blocks (features 0 .. 63) -- the stream,
// its barriers and the operand build are unchanged.
// LATE: val(s, .) of an EVEN k-step s reads data that lands with the barrier inside k-step s - 1 (a side stream): those operands are
// built after that barrier.
// NST: vector-memory store instructions one call of ``sink`` issues (0: unknown / none -> the barriers wait for everything)
template <int KG, int NG = 2, bool LATE = false, int NST = 0, class VAL, class SIDE, class SINK>
__device__ __forceinline__ void gemm_rs(f32x16 (&C)[8], WStream& ws, VAL&& val, SIDE&& side, SINK&& sink) {
    static_assert(KG % 2 == 0, "k-step pairs");
    static_assert(NST == 0 || NST == 2 || NST == 4, "store counts landed_barrier knows");
    ws.nst = NST;
    FragB b;
    build_frag(b, val, 0, sink);
#pragma unroll
    for (int sp = 0; sp < KG / 2; ++sp) {
        kstep_r<NG, LATE, false>(C, ws, b, val, side, sink, 2 * sp + 1, true);
        kstep_r<NG, LATE, true>(C, ws, b, val, side, sink, 2 * sp + 2, 2 * sp + 2 < KG);
    }
    ws.nst = 0;
}
template <int KG, int NG = 2, bool LATE = false, class VAL, class SIDE>
__device__ __forceinline__ void gemm_r(f32x16 (&C)[8], WStream& ws, VAL&& val, SIDE&& side) { gemm_rs<KG, NG, LATE, 0>(C, ws, val, side, NoSink()); }
template <int KG, int NG = 2, class VAL>
__device__ __forceinline__ void gemm_r(f32x16 (&C)[8], WStream& ws, VAL&& val) { gemm_r<KG, NG, false>(C, ws, val, NoSide()); }

// accumulators start from the layer's bias: register 4 q + i of block b is feature 32 b + 8 q + 4 hi + i
__device__ __forceinline__ void init8(f32x16 (&C)[8], const float* bl, int hi) {
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(bl + 32 * b + 8 * q + 4 * hi);
            C[b][4 * q] = v.x; C[b][4 * q + 1] = v.y; C[b][4 * q + 2] = v.z; C[b][4 * q + 3] = v.w;
        }
}
__device__ __forceinline__ void copy8(f32x16 (&P)[8], const f32x16 (&C)[8]) {
#pragma unroll
    for (int b = 0; b < 8; ++b) P[b] = C[b];
}
// copy8 with the copy PINNED to the accumulation registers (AGPR class, through an empty asm with an "a" operand).  A kernel that keeps
// the previous layer's 128 values next to 128 live accumulators has 256 of its 512 registers in those two arrays; left to the
// allocator, part of P is homed in the architectural half, which then has no room for the operand fragments (96 + 24 registers) and
// whole 16-register accumulator tuples go to scratch (round 4: k_deform_vjp_x3r<SAVE> 57 -> 0 spilled registers).  P's elements are
// read one at a time (v_accvgpr_read) where the next layer's operand is built.
__device__ __forceinline__ void copy8_acc(f32x16 (&P)[8], const f32x16 (&C)[8]) {
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t = C[b][r];
            asm("" : "+a"(t));
            P[b][r] = t;
        }
}

// softplus'(z) = sigmoid(100 z) recovered from s = softplus(z): 1 - exp(-100 s), series where that cancels (chain_common.h
// softplus100_grad_from_s on the raw exp unit).  The select is written as v_cmp + v_cndmask: left to the compiler, the ternary becomes a
// divergent branch per element in the middle of the MFMA stream.
__device__ __forceinline__ float dphi_from_s(float s) {
    const float x = 100.f * s;
    const float big = 1.f - __builtin_amdgcn_exp2f(-1.4426950408889634f * x);
    const float ser = x * (1.f - x * (0.5f - x * (1.f / 6.f)));
    float r;
    asm("v_cmp_gt_f32 vcc, 0x3ca3d70a, %3\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(r) : "v"(big), "v"(ser), "v"(x) : "vcc");      // 0.02 > x ? ser : big
    return r;
}

// four consecutive features of one row of a row-major [rows][256] stack.  A row's 128-B line is completed by four such stores (two lane
// halves x two pieces x two k-steps): ES_X3R_NT_STORES=1 (dev builds) marks them non-temporal like the fp32 kernels' stream-outs; the
// default leaves them to the L2's write combining (measured: the non-temporal form wrote 1.5-1.66x the bytes at ~2.3 TB/s)
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
    const v4f_frag t = {a, b, c, d};
#ifdef ES_X3R_NT_STORES
    __builtin_nontemporal_store(t, reinterpret_cast<v4f_frag*>(p));
#else
    *reinterpret_cast<v4f_frag*>(p) = t;
#endif
}
// the operand of k-step s (8 values: features 16 s + 4 hi .. + 3 and 16 s + 8 + 4 hi .. + 3) into a row whose base already holds + 4 hi
__device__ __forceinline__ void st_kstep(float* row_hi, int s, const float (&v)[8]) {
    st4(row_hi + 16 * s, v[0], v[1], v[2], v[3]);
    st4(row_hi + 16 * s + 8, v[4], v[5], v[6], v[7]);
}

// Coalescing sink of the training kernels' saved stacks.  A lane owns a ROW of a row-major [rows][ld] stack and gets 16 of its columns per
// k-step (two 16-B pieces, the other lane half's between them): stored directly, every instruction touches 64 different cache lines with
// 16 B each.  Instead the operands of a k-step PAIR (32 rows x 32 columns of the wave = one full 128-B line per row) are collected in an
// LDS tile of the wave and written out as four stores of 8 complete lines each (lane -> row 8 i + lane / 8, 16-B chunk lane % 8).  Same
// wave writes and reads the tile: LDS operations of a wave execute in order, no barrier.  The flush runs in the sink of the pair's odd
// operand = at the end of an EVEN k-step, so its four stores are the youngest memory operations at the next barrier (gemm_rs NST = 4).
constexpr int XR_TILE_LD = 36;                                   // floats per tile row (16-B aligned, shifts consecutive rows by 4 banks)
constexpr int XR_TILE_BYTES = 4 * 32 * XR_TILE_LD * 4;            // one tile per wave
struct RowTile {
    float* t;           // this wave's tile [32][XR_TILE_LD]
    int rowidx;         // this lane's row among the wave's 32 consecutive stack rows
    int hi, lane;
    // Flush mapping: store i of a flush covers rows 4 (lane / 8) + i, 16-B chunk lane % 8 -- 8 complete 128-B lines per store, and the four
    // stores of a flush (and the pairs of a whole layer) differ by compile-time byte offsets only (i LD 4 + 128 (s / 2) < 4096, the
    // immediate field of a global store): ONE 32-bit lane offset against the wave-uniform base instead of four 64-bit addresses per lane
    // (round 4: the address registers were what the saving kernels spilled).
    template <int LD>
    __device__ __forceinline__ void put(int s, const float (&v)[8], float* wave_base) const {      // wave_base = &stack[first row of the wave][0]
        static_assert(3 * LD * 4 + 128 * 7 < 4096, "immediate offsets of the flush");
        float* p = t + rowidx * XR_TILE_LD + 16 * (s & 1) + 4 * hi;
        *reinterpret_cast<v4f_frag*>(p) = v4f_frag{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<v4f_frag*>(p + 8) = v4f_frag{v[4], v[5], v[6], v[7]};
        if (s & 1) {
            const unsigned r = 4u * ((unsigned)lane >> 3), c = 4u * ((unsigned)lane & 7u);
            const float* tp = t + r * XR_TILE_LD + c;
            float* gp = wave_base + (r * (unsigned)LD + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const v4f_frag x = *reinterpret_cast<const v4f_frag*>(tp + i * XR_TILE_LD);
                *reinterpret_cast<v4f_frag*>(gp + i * LD + 32 * (s >> 1)) = x;
            }
        }
    }
};

// mask bit of register r of feature block b inside the 128-bit word of a (layer, point, lane half)
__device__ __forceinline__ void mask_set(u32x4& mk, int b, int r, bool m) { mk[b >> 1] |= (m ? 1u : 0u) << ((b & 1) * 16 + r); }
// mask_set whose or is DONE before the next statement: the empty volatile asm on the word is an anchor the scheduler cannot move the
// or (and hence the comparison's 64-bit lane mask, an SGPR pair) past.  Left to itself the compiler deferred the 32 ors of a k-step and
// kept their lane masks alive: 371-387 SGPR spills (v_writelane / v_readlane + hazard nops) in k_deform_jvp_x3r.
__device__ __forceinline__ void mask_set_now(u32x4& mk, int b, int r, bool m) {
    unsigned w = mk[b >> 1] | ((m ? 1u : 0u) << ((b & 1) * 16 + r));
    asm volatile("" : "+v"(w));
    mk[b >> 1] = w;
}
__device__ __forceinline__ bool mask_get(const u32x4& mk, int b, int r) { return (mk[b >> 1] >> ((b & 1) * 16 + r)) & 1u; }

// the element of the VALUE column of this lane's point: lanes 0-15 of a lane half keep their own, lanes 16-31 get their partner's
// (v_permlane16_swap_b32 exchanges the odd 16-lane rows of its first operand with the even rows of its second)
__device__ __forceinline__ float value_row(float z) {
    return __uint_as_float(__builtin_amdgcn_permlane16_swap(__float_as_uint(z), __float_as_uint(z), false, false)[0]);
}

}  // namespace es
