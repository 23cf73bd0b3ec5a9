// K6: hand-written backward of the fused point evaluation (the reference gets it from autograd with
// create_graph=True double-backward, endosurf.py:598, 616, 640-656).  Given adjoints of (sdf, g_o, rgb) per point:
//   color_bwd   reverse sweep through ColorNetwork -> adjoints of its pre-activations (for the weight-gradient GEMMs)
//               and of its inputs: x_c (via enc10), g_c, d_c -> v = J d (via normalize), feat
//   sdf_bwd     (i) forward tangent sweep along gbar_c: the adjoint of the reverse-mode input gradient g_c is a
//               forward-mode directional derivative; yields tau_l and the second-order terms 100(1-phi')rho_l pi_l
//               (softplus''); (ii) ordinary reverse sweep of the value pass seeded with [sdfbar, featbar]
//   deform_tan  forward tangent sweep of the deformation network along gbar_o: g_o = J^T g_c is linear in g_c with
//               adjoint J gbar_o (consumed by sdf_bwd) and linear in every W_l (pairs (tau_l, r_l) with the VJP sweep's r_l)
//   deform_bwd  reverse sweep on 2 rows per point (value row seeded with xbar_c, J d row with vbar); ReLU'' = 0
// Weight gradients are formed afterwards by wgrad.hip from the streamed (input, adjoint) pairs.
#include <type_traits>

#include "chain_common.h"
#include "encode.h"
#include "launch.h"
#include "tabs.h"
#include "timing.h"
#include "workspace.h"
#include "point_bwd_bodies.h"

namespace es {

#ifdef ES_PROFILE_BWD        // dev builds only: cycle stamps of block 0 / thread 0 inside sdf_bwd_tile (tools/dev/bwd_profile.py)
__device__ long long b_prof[256];
extern "C" int es_debug_b_profile(long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(b_prof), sizeof(long long) * (n < 256 ? n : 256)); }
#endif

// Two-segment launches as in point_fwd.hip: the tail's two dependent stages ride in the halves of the main deformation launch
//   colour_bwd(main) | sdf_bwd(main) | sdf_bwd(tail) + deform_bwd(main, 1st half) | deform_bwd(tail) + deform_bwd(main, 2nd half)
enum BwdBody { BB_NONE = 0, BB_COLOR, BB_SDF, BB_DEFORM, BB_TAN, BB_TAN_SDF, BB_DEFORM_HALF };
template <int B>
__device__ __forceinline__ void bwd_body(const BwdArgs& a, int tile) {
    if constexpr (B == BB_COLOR) color_bwd_tile(a, tile);
    else if constexpr (B == BB_SDF) sdf_bwd_tile(a, tile);
    else if constexpr (B == BB_DEFORM) deform_bwd_tile(a, tile);
    else if constexpr (B == BB_DEFORM_HALF) deform_bwd_tile<true>(a, tile);      // 16-point tiles (the tail's stand-alone launch)
    else if constexpr (B == BB_TAN) deform_tan_tile(a, tile);
    else if constexpr (B == BB_TAN_SDF) {      // both stages of a colour-less tile in one workgroup (J gbar_o goes through the workspace)
        deform_tan_tile(a, tile);
        __syncthreads();
        sdf_bwd_tile(a, tile);
    }
}
template <int B0, int B1>
__global__ __launch_bounds__(NTHREADS, 2) void k_point_bwd(BwdArgs a, int n0, int t0, int t1) {
    if constexpr (B0 != BB_NONE) {
        if ((int)blockIdx.x < n0) { bwd_body<B0>(a, t0 + blockIdx.x); return; }
    }
    bwd_body<B1>(a, t1 + (int)blockIdx.x - n0);
}
constexpr int bwd_lds(int b) {
    return b == BB_COLOR ? CBWD_LDS_BYTES : (b == BB_SDF ? SBWD_LDS_BYTES : (b == BB_DEFORM || b == BB_DEFORM_HALF ? DBWD_LDS_BYTES : (b == BB_TAN || b == BB_TAN_SDF ? LEAN_LDS_BYTES : 0)));
}
template <int B0, int B1>
static int launch_bwd(const BwdArgs& a, int n0, int t0, int n1, int t1, hipStream_t st) {
    constexpr int lds = bwd_lds(B0) > bwd_lds(B1) ? bwd_lds(B0) : bwd_lds(B1);
    static DeviceOnce attr_done;
    if (attr_done.first()) {
        if (int e = allow_big_lds(k_point_bwd<B0, B1>, lds)) return e;
        attr_done.done();
    }
    if (n0 + n1 <= 0) return ST_OK;
    hipLaunchKernelGGL((k_point_bwd<B0, B1>), dim3(n0 + n1), dim3(NTHREADS), lds, st, a, n0, t0, t1);
    return ST_OK;
}

// train_x3r.hip / query_x3.hip
int deform_tan_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, const float* d_go, hipStream_t st, int m_rows = 0);
int deform_bwd_x3r_with_tail(const BwdArgs& ba, const void* packed_r, int m_main, hipStream_t st);
int color_bwd_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, bool deform, int m_color, const float* d_rgb,
                  hipStream_t st);
int deform_bwd_x3r(const void* packed_r, const float* weff, float* ws, const WsLayout& L, int M, int m_color, hipStream_t st);
const void* packed_x3r_part(const void* packed_x3);

int point_backward_chains(const PointSrc& src, const float* packed, const float* weff, float* ws, int flags, int m_color,
                          const float* d_sdf, const float* d_go, const float* d_rgb, hipStream_t st, const void* packed_x3) {
    if (src.M <= 0) return ST_OK;
    BwdArgs a;
    a.src = src; a.tb = make_tabs(); a.packed = reinterpret_cast<const float4*>(packed); a.weff = weff; a.ws = ws;
    a.L = ws_layout(src.M, flags); a.flags = flags; a.d_sdf = d_sdf; a.d_go = d_go; a.d_rgb = d_rgb;
    a.M_color = (flags & PF_COLOR) ? (m_color > 0 ? m_color : src.M) : 0;
    const int Mp = a.L.Mp, Mcp = round_up64(a.M_color);
    const bool deform = flags & PF_DEFORM;
    if (flags & PF_X3_CHAIN) {
        // the workspace comes from the split-precision training chain (point_fwd.hip): its backward on the register-resident core
        // (train_x3r.hip): colour reverse sweep | deformation tangent sweep | SDF backward (fp32 kernel) | deformation reverse sweep
        if (!packed_x3) return fail(ST_BAD_ARG, "point_backward_chains", "PF_X3_CHAIN needs the split weights (es_pack_x3)");
        const void* pr = packed_x3r_part(packed_x3);
        if (deform && aux_tail(flags, a.M_color, src.M) && a.M_color % 128 == 0) {
            // the forward's tail arrangement (point_fwd.hip) mirrored: the tail on the fp32 family, its tangent + SDF-backward stages at the
            // head of this family's deformation reverse sweep (train_x3r.hip k_deform_bwd_x3r_tail):
            //   colour_bwd(main) | tan(main) | sdf_bwd(main, fp32) | [tan + sdf_bwd](tail, fp32) + deform_bwd(main) | deform_bwd(tail, fp32)
            const int Mc = a.M_color;
            if (int e = color_bwd_x3r(src, pr, weff, ws, a.L, deform, Mc, d_rgb, st)) return e;
            if (int e = deform_tan_x3r(src, pr, weff, ws, a.L, d_go, st, Mc)) return e;
            { ScopedTimer tm(KID_SDF_BWD, Mc, st); if (int e = launch_bwd<BB_NONE, BB_SDF>(a, 0, 0, Mc / TM, 0, st)) return e; }
            if (int e = deform_bwd_x3r_with_tail(a, pr, Mc, st)) return e;
            { ScopedTimer tm(KID_DEFORM_BWD, Mp - Mc, st); if (int e = launch_bwd<BB_NONE, BB_DEFORM_HALF>(a, 0, 0, (Mp - Mc) / 16, Mc / 16, st)) return e; }
            return hip_last("point_backward_chains");
        }
        if (flags & PF_COLOR) { if (int e = color_bwd_x3r(src, pr, weff, ws, a.L, deform, a.M_color, d_rgb, st)) return e; }
        if (deform) { if (int e = deform_tan_x3r(src, pr, weff, ws, a.L, d_go, st)) return e; }
        { ScopedTimer tm(KID_SDF_BWD, src.M, st); if (int e = launch_bwd<BB_NONE, BB_SDF>(a, 0, 0, Mp / TM, 0, st)) return e; }
        if (deform) { if (int e = deform_bwd_x3r(pr, weff, ws, a.L, src.M, a.M_color, st)) return e; }
        return hip_last("point_backward_chains");
    }
    if (deform && aux_tail(flags, a.M_color, src.M)) {
        // tail stages mixed into the main launches as in point_fwd.hip:
        //   colour_bwd(main) | tan(main) | sdf_bwd(main) | [tan + sdf_bwd](tail) + deform_bwd(main) | deform_bwd(tail)
        const int Mc = a.M_color;
        { ScopedTimer tm(KID_COLOR_BWD, Mc, st); if (int e = launch_bwd<BB_NONE, BB_COLOR>(a, 0, 0, Mc / TM, 0, st)) return e; }
        { ScopedTimer tm(KID_DEFORM_TAN, Mc, st); if (int e = launch_bwd<BB_NONE, BB_TAN>(a, 0, 0, Mc / TM, 0, st)) return e; }
        { ScopedTimer tm(KID_SDF_BWD, Mc, st); if (int e = launch_bwd<BB_NONE, BB_SDF>(a, 0, 0, Mc / TM, 0, st)) return e; }
        { ScopedTimer tm(KID_DEFORM_BWD, src.M, st);
          if (int e = launch_bwd<BB_TAN_SDF, BB_DEFORM>(a, (Mp - Mc) / TM, Mc / TM, Mc / 32, 0, st)) return e;
          if (int e = launch_bwd<BB_NONE, BB_DEFORM_HALF>(a, 0, 0, (Mp - Mc) / 16, Mc / 16, st)) return e; }      // half-height tiles
        return hip_last("point_backward_chains");
    }
    if (!deform && aux_tail(flags, a.M_color, src.M)) {
        // no deformation network: sdf_bwd(tail) + colour_bwd(main) | sdf_bwd(main)   (see point_fwd.hip: the tail's tiles ride at the
        // head of the colour launch instead of adding a third round to the SDF launch; they do not depend on the colour backward)
        const int Mc = a.M_color;
        { ScopedTimer tm(KID_COLOR_BWD, Mc, st); if (int e = launch_bwd<BB_SDF, BB_COLOR>(a, (Mp - Mc) / TM, Mc / TM, Mc / TM, 0, st)) return e; }
        { ScopedTimer tm(KID_SDF_BWD, Mc, st); if (int e = launch_bwd<BB_NONE, BB_SDF>(a, 0, 0, Mc / TM, 0, st)) return e; }
        return hip_last("point_backward_chains");
    }
    if (deform) { ScopedTimer tm(KID_DEFORM_TAN, src.M, st); if (int e = launch_bwd<BB_NONE, BB_TAN>(a, 0, 0, Mp / TM, 0, st)) return e; }
    if (flags & PF_COLOR) { ScopedTimer tm(KID_COLOR_BWD, a.M_color, st); if (int e = launch_bwd<BB_NONE, BB_COLOR>(a, 0, 0, Mcp / TM, 0, st)) return e; }
    { ScopedTimer tm(KID_SDF_BWD, src.M, st); if (int e = launch_bwd<BB_NONE, BB_SDF>(a, 0, 0, Mp / TM, 0, st)) return e; }
    if (deform) { ScopedTimer tm(KID_DEFORM_BWD, src.M, st); if (int e = launch_bwd<BB_NONE, BB_DEFORM>(a, 0, 0, Mp / 32, 0, st)) return e; }
    return hip_last("point_backward_chains");
}

}  // namespace es
