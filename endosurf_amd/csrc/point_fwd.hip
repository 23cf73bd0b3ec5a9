// K4 forward: per-point network evaluation of EndoSurfNet.forward (reference endosurf.py:660-689) and
// get_sdf_grad_from_observed_space (:581-601), restructured so that no network pass is repeated:
//   deform_fwd  DeformNetwork (:724-738) on 2 rows per point: value + forward-mode tangent along the ray direction
//               -> x_c and J d (the reference builds the full 3x3 Jacobian with 3 autograd sweeps, :621-658, only to form J d)
//   sdf_fwd     SDFNetwork (:773-786) value pass (sdf, 256 features) + analytic reverse sweep for g_c = d sdf / d x_c (:603-619)
//   deform_vjp  reverse sweep of the deformation network for the covector g_c -> g_o = J^T g_c (identical to :581-601 up
//               to rounding)
//   color_fwd   ColorNetwork (:828-842) on [enc10(x_c), g_c, enc4(normalize(J d)), feat] -> sigmoid rgb
// With PF_SAVE the layer inputs / reverse adjoints are streamed to the workspace for the backward pass.
#include "chain_common.h"
#include "encode.h"
#include "launch.h"
#include "tabs.h"
#include "timing.h"
#include "workspace.h"
#include "point_fwd_bodies.h"

namespace es {

// One launch = two segments of tiles, possibly of different networks: blocks [0, n0) run body B0 on tiles t0.., the rest run
// body B1 on tiles t1...  Used for the short colour-less tail of a training batch (errorondepth / surface-neighbour points):
// launched with its own network's main tiles it adds a nearly empty third round of the 512 workgroup slots to the SDF
// kernels (1024 + 48 tiles), which costs a full tile time.  Every main launch of this workload is a whole number of rounds,
// so extra tiles always cost something -- least inside a launch of MANY short rounds: the tail's two dependent stages
// (deform, then SDF) are therefore mixed into the two halves of the 8-round deformation launch of the main tiles.
enum FwdBody { FB_NONE = 0, FB_DEFORM, FB_SDF, FB_COLOR, FB_VJP, FB_SDF_VJP, FB_DEFORM_HALF, FB_SDF_VJP_HALF, FB_SDF_HALF };
template <int B>
__device__ __forceinline__ void fwd_body(const FwdArgs& a, int tile) {
    if constexpr (B == FB_DEFORM) deform_fwd_tile(a, tile);
    else if constexpr (B == FB_DEFORM_HALF) deform_fwd_tile<true>(a, tile);      // 16-point tiles (the tail's stand-alone launch)
    else if constexpr (B == FB_SDF) sdf_fwd_tile(a, tile);
    else if constexpr (B == FB_COLOR) color_fwd_tile(a, tile);
    else if constexpr (B == FB_VJP) deform_vjp_tile(a, tile);
    else if constexpr (B == FB_SDF_VJP) {      // both stages of a colour-less tile in one workgroup (g_c goes through the workspace)
        sdf_fwd_tile(a, tile);
        __syncthreads();
        deform_vjp_tile(a, tile);
    } else if constexpr (B == FB_SDF_VJP_HALF) {      // the same on 32-row half tiles (stand-alone pieces of a tail: es_point_forward_rows)
        sdf_fwd_tile<true>(a, tile);
        __syncthreads();
        deform_vjp_tile<true>(a, tile);
    } else if constexpr (B == FB_SDF_HALF) sdf_fwd_tile<true>(a, tile);
}
template <int B0, int B1>
__global__ __launch_bounds__(NTHREADS, 2) void k_point_fwd(FwdArgs a, int n0, int t0, int t1) {
    if constexpr (B0 != FB_NONE) {
        if ((int)blockIdx.x < n0) { fwd_body<B0>(a, t0 + blockIdx.x); return; }
    }
    fwd_body<B1>(a, t1 + (int)blockIdx.x - n0);
}
constexpr int fwd_lds(int b) { return b == FB_COLOR ? CFWD_LDS_BYTES : (b == FB_NONE ? 0 : LEAN_LDS_BYTES); }   // SBWD <= LEAN
template <int B0, int B1>
static int launch_fwd(const FwdArgs& a, int n0, int t0, int n1, int t1, hipStream_t st) {
    constexpr int lds = fwd_lds(B0) > fwd_lds(B1) ? fwd_lds(B0) : fwd_lds(B1);
    static DeviceOnce attr_done;
    if (attr_done.first()) {
        if (int e = allow_big_lds(k_point_fwd<B0, B1>, lds)) return e;
        attr_done.done();
    }
    if (n0 + n1 <= 0) return ST_OK;
    hipLaunchKernelGGL((k_point_fwd<B0, B1>), dim3(n0 + n1), dim3(NTHREADS), lds, st, a, n0, t0, t1);
    return ST_OK;
}

// -------------------------------------------------------------------------------------------------------------
// infer_x3r.hip / query_x3.hip
int deform_jvp_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, bool save, hipStream_t st);
int deform_vjp_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, bool save, hipStream_t st, int m_rows = 0);
int deform_jvp_x3r_with_tail(const FwdArgs& fa, const void* packed_r, int m_main, hipStream_t st);
int sdf_fwd_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, bool deform, bool color, hipStream_t st);
int color_fwd_x3r(const PointSrc& src, const void* packed_r, const float* weff, float* ws, const WsLayout& L, bool deform, int Mcp, bool save, hipStream_t st);
const void* packed_x3r_part(const void* packed_x3);

// packed_x3 (nullable): the split weights of es_pack_x3; with PF_X3 and without PF_SAVE the deformation- and SDF-network launches of a
// no-grad evaluation run in split precision (opt-in)
int point_forward(const PointSrc& src, const float* packed, const float* weff, float* ws, int flags, int m_color, hipStream_t st,
                  const void* packed_x3) {
    if (src.M <= 0) return ST_OK;
    FwdArgs a;
    a.src = src; a.tb = make_tabs(); a.packed = reinterpret_cast<const float4*>(packed); a.weff = weff; a.ws = ws;
    a.L = ws_layout(src.M, flags); a.flags = flags;
    a.M_color = (flags & PF_COLOR) ? (m_color > 0 ? m_color : src.M) : 0;
    const int Mp = a.L.Mp, Mcp = round_up64(a.M_color);
    const bool deform = flags & PF_DEFORM;
    if ((flags & PF_X3) && !(flags & PF_SAVE) && packed_x3) {
        // opt-in split-precision inference: deformation value + tangent | SDF value + features + reverse sweep | colour | VJP
        const void* pr = packed_x3r_part(packed_x3);
        if (deform) { if (int e = deform_jvp_x3r(src, pr, weff, ws, a.L, false, st)) return e; }
        if (int e = sdf_fwd_x3r(src, pr, weff, ws, a.L, deform, (flags & PF_COLOR) != 0, st)) return e;
        if (flags & PF_COLOR) { if (int e = color_fwd_x3r(src, pr, weff, ws, a.L, deform, Mcp, false, st)) return e; }
        return deform ? deform_vjp_x3r(src, pr, weff, ws, a.L, false, st) : hip_last("point_forward");
    }
    if ((flags & PF_X3_CHAIN) && (flags & PF_SAVE) && packed_x3) {
        // opt-in split-precision TRAINING chain: all four launches on the register-resident core, keeping what the backward needs in
        // the fp32 kernels' buffers (row-major stacks; the ReLU mask words are this family's: PF_X3_CHAIN tells the backward); the SDF
        // network stays on the fp32 kernels (see infer_x3r.hip)
        const void* pr = packed_x3r_part(packed_x3);
        if (deform && aux_tail(flags, a.M_color, src.M) && a.M_color % 128 == 0) {
            // colour-less tail behind a block-aligned main part (the fused training batch): the tail goes through the fp32 family, its
            // dependent stages hidden in this family's 4-round launch (infer_x3r.hip k_deform_jvp_x3r_tail):
            //   deform(tail, fp32) | [sdf + vjp](tail, fp32) + jvp(main) | sdf(main, fp32) | colour(main) | vjp(main)
            const int Mc = a.M_color;
            { ScopedTimer tm(KID_DEFORM_FWD, Mp - Mc, st); if (int e = launch_fwd<FB_NONE, FB_DEFORM_HALF>(a, 0, 0, (Mp - Mc) / 16, Mc / 16, st)) return e; }
            if (int e = deform_jvp_x3r_with_tail(a, pr, Mc, st)) return e;
            { ScopedTimer tm(KID_SDF_FWD, Mc, st); if (int e = launch_fwd<FB_NONE, FB_SDF>(a, 0, 0, Mc / TM, 0, st)) return e; }
            if (int e = color_fwd_x3r(src, pr, weff, ws, a.L, deform, Mc, true, st)) return e;
            return deform_vjp_x3r(src, pr, weff, ws, a.L, true, st, Mc);
        }
        if (deform) { if (int e = deform_jvp_x3r(src, pr, weff, ws, a.L, true, st)) return e; }
        { ScopedTimer tm(KID_SDF_FWD, src.M, st); if (int e = launch_fwd<FB_NONE, FB_SDF>(a, 0, 0, Mp / TM, 0, st)) return e; }
        if (flags & PF_COLOR) { if (int e = color_fwd_x3r(src, pr, weff, ws, a.L, deform, Mcp, true, st)) return e; }
        return deform ? deform_vjp_x3r(src, pr, weff, ws, a.L, true, st) : hip_last("point_forward");
    }
    if (deform && aux_tail(flags, a.M_color, src.M)) {
        // main tiles [0, Mc), colour-less tail [Mc, Mp).  Every main launch is a whole number of rounds of the 512 workgroup
        // slots, so the tail's few tiles (whose three stages depend on each other) cost least where they overlap a launch of
        // many short rounds or a launch of longer tiles:
        //   deform(tail) | [sdf + vjp](tail) + deform(main) | sdf(main) | colour(main) | vjp(main)
        const int Mc = a.M_color;
        { ScopedTimer tm(KID_DEFORM_FWD, src.M, st);
          if (int e = launch_fwd<FB_NONE, FB_DEFORM_HALF>(a, 0, 0, (Mp - Mc) / 16, Mc / 16, st)) return e;      // at one tile's latency: half-height tiles
          if (int e = launch_fwd<FB_SDF_VJP, FB_DEFORM>(a, (Mp - Mc) / TM, Mc / TM, Mc / 32, 0, st)) return e; }
        { ScopedTimer tm(KID_SDF_FWD, Mc, st); if (int e = launch_fwd<FB_NONE, FB_SDF>(a, 0, 0, Mc / TM, 0, st)) return e; }
        { ScopedTimer tm(KID_COLOR_FWD, Mc, st); if (int e = launch_fwd<FB_NONE, FB_COLOR>(a, 0, 0, Mc / TM, 0, st)) return e; }
        { ScopedTimer tm(KID_DEFORM_VJP, Mc, st); if (int e = launch_fwd<FB_NONE, FB_VJP>(a, 0, 0, Mc / TM, 0, st)) return e; }
        return hip_last("point_forward");
    }
    if (!deform && aux_tail(flags, a.M_color, src.M)) {
        // no deformation network (base_d*k1 configs): sdf(main) | sdf(tail) + colour(main).  The main launches are whole rounds of
        // the 512 workgroup slots; the tail's 48 SDF tiles in the SDF launch would add a third, nearly empty round (+0.6 ms), at the
        // head of the colour launch they only take 48 slots away from its first two rounds (+0.06 ms)
        const int Mc = a.M_color;
        { ScopedTimer tm(KID_SDF_FWD, Mc, st); if (int e = launch_fwd<FB_NONE, FB_SDF>(a, 0, 0, Mc / TM, 0, st)) return e; }
        { ScopedTimer tm(KID_COLOR_FWD, Mc, st); if (int e = launch_fwd<FB_SDF, FB_COLOR>(a, (Mp - Mc) / TM, Mc / TM, Mc / TM, 0, st)) return e; }
        return hip_last("point_forward");
    }
    if (deform) { ScopedTimer tm(KID_DEFORM_FWD, src.M, st); if (int e = launch_fwd<FB_NONE, FB_DEFORM>(a, 0, 0, Mp / 32, 0, st)) return e; }
    { ScopedTimer tm(KID_SDF_FWD, src.M, st); if (int e = launch_fwd<FB_NONE, FB_SDF>(a, 0, 0, Mp / TM, 0, st)) return e; }
    if (flags & PF_COLOR) { ScopedTimer tm(KID_COLOR_FWD, a.M_color, st); if (int e = launch_fwd<FB_NONE, FB_COLOR>(a, 0, 0, Mcp / TM, 0, st)) return e; }
    if (deform) { ScopedTimer tm(KID_DEFORM_VJP, src.M, st); if (int e = launch_fwd<FB_NONE, FB_VJP>(a, 0, 0, Mp / TM, 0, st)) return e; }
    return hip_last("point_forward");
}


// Rows [row0, row0 + nrows) of a workspace laid out for ALL of src.M points (ABI v8, es_point_forward_rows): the forward of a workspace
// that is filled piece by piece.  The reference trainer calls renderer(rays), errorondepth and surface_neighbour_error one after the other
// (trainer_endosurf.py:130, :140, :155) and back-propagates once: the render's workspace is laid out with room for the colour-less points
// of the two later calls behind its samples, each call evaluates its own rows when it is made, and ONE es_point_backward over the whole
// workspace -- with the tail's stages mixed into the main launches (point_bwd.hip) -- replaces three separate backward chains, two of them
// latency-bound (16 - 32 workgroups) launches at a tile's full latency each.
//   row0 == 0, nrows == m_color:  the main part (colour points): deform | sdf | colour | vjp over its tiles only
//   row0 >= m_color:              a piece of the colour-less tail (row0, nrows multiples of 64): deform | [sdf + vjp], both on half-height tiles
// fp32 family only.
int point_forward_rows(const PointSrc& src, const float* packed, const float* weff, float* ws, int flags, int m_color, int row0, int nrows,
                       hipStream_t st) {
    if (src.M <= 0 || nrows <= 0) return ST_OK;
    FwdArgs a;
    a.src = src; a.tb = make_tabs(); a.packed = reinterpret_cast<const float4*>(packed); a.weff = weff; a.ws = ws;
    a.L = ws_layout(src.M, flags); a.flags = flags;
    a.M_color = (flags & PF_COLOR) ? (m_color > 0 ? m_color : src.M) : 0;
    const int Mc = a.M_color;
    const bool deform = flags & PF_DEFORM;
    if (row0 + nrows > a.L.Mp) return fail(ST_BAD_ARG, "point_forward_rows", "rows beyond the workspace");
    if (row0 == 0 && nrows == Mc && Mc % 64 == 0) {
        if (deform) { ScopedTimer tm(KID_DEFORM_FWD, Mc, st); if (int e = launch_fwd<FB_NONE, FB_DEFORM>(a, 0, 0, Mc / 32, 0, st)) return e; }
        { ScopedTimer tm(KID_SDF_FWD, Mc, st); if (int e = launch_fwd<FB_NONE, FB_SDF>(a, 0, 0, Mc / TM, 0, st)) return e; }
        { ScopedTimer tm(KID_COLOR_FWD, Mc, st); if (int e = launch_fwd<FB_NONE, FB_COLOR>(a, 0, 0, Mc / TM, 0, st)) return e; }
        if (deform) { ScopedTimer tm(KID_DEFORM_VJP, Mc, st); if (int e = launch_fwd<FB_NONE, FB_VJP>(a, 0, 0, Mc / TM, 0, st)) return e; }
        return hip_last("point_forward_rows");
    }
    if (row0 < Mc || row0 % 64 != 0 || nrows % 64 != 0)
        return fail(ST_BAD_ARG, "point_forward_rows", "rows: either the whole colour part [0, m_color) or a 64-aligned piece of the colour-less tail");
    if (deform) {
        { ScopedTimer tm(KID_DEFORM_FWD, nrows, st); if (int e = launch_fwd<FB_NONE, FB_DEFORM_HALF>(a, 0, 0, nrows / 16, row0 / 16, st)) return e; }
        { ScopedTimer tm(KID_SDF_FWD, nrows, st); if (int e = launch_fwd<FB_NONE, FB_SDF_VJP_HALF>(a, 0, 0, nrows / 32, row0 / 32, st)) return e; }
    } else {
        ScopedTimer tm(KID_SDF_FWD, nrows, st);
        if (int e = launch_fwd<FB_NONE, FB_SDF_HALF>(a, 0, 0, nrows / 32, row0 / 32, st)) return e;
    }
    return hip_last("point_forward_rows");
}

// The VJP sweep of the deformation network on its own, for the covector the caller has written to WS_GC of a workspace that a forward
// of the SAME points has filled (the sweep reads that forward's ReLU mask words): WS_GO = J^T c, WS_CURV = the encoding-curvature sums
// against the same adjoint.  Nothing is saved (PF_SAVE is ignored: the forward's saved VJP adjoints stay as they are).
int point_vjp(const PointSrc& src, const float* packed, const float* weff, float* ws, int flags, hipStream_t st) {
    if (src.M <= 0) return ST_OK;
    FwdArgs a;
    a.src = src; a.tb = make_tabs(); a.packed = reinterpret_cast<const float4*>(packed); a.weff = weff; a.ws = ws;
    a.L = ws_layout(src.M, flags); a.flags = flags & ~PF_SAVE;
    a.M_color = 0;
    ScopedTimer tm(KID_DEFORM_VJP, src.M, st);
    if (int e = launch_fwd<FB_NONE, FB_VJP>(a, 0, 0, a.L.Mp / TM, 0, st)) return e;
    return hip_last("point_vjp");
}

// ColorNetwork.forward(x, n, d, geo_feat) (reference endosurf.py:828-842) on explicit inputs: the caller has written x -> WS_XC,
// n -> WS_GC, geo_feat -> WS_FEAT of a (PF_COLOR, no deformation) workspace and passes d as the point source's view directions; only
// the colour body runs, with the direction taken as given (EndoSurfNet.forward normalises J d before it calls the colour network,
// the colour network itself does not).  rgb -> WS_RGB.
int color_forward(const PointSrc& src, const float* packed, const float* weff, float* ws, hipStream_t st) {
    if (src.M <= 0) return ST_OK;
    FwdArgs a;
    a.src = src; a.tb = make_tabs(); a.packed = reinterpret_cast<const float4*>(packed); a.weff = weff; a.ws = ws;
    a.flags = PF_COLOR | PF_RAW_DIR;
    a.L = ws_layout(src.M, PF_COLOR); a.M_color = src.M;
    ScopedTimer tm(KID_COLOR_FWD, src.M, st);
    if (int e = launch_fwd<FB_NONE, FB_COLOR>(a, 0, 0, a.L.Mp / TM, 0, st)) return e;
    return hip_last("color_forward");
}

}  // namespace es
