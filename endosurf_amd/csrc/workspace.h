// Layout of the per-call point-evaluation workspace (one caller-owned fp32 device buffer).
//
// All per-point tensors are row-major [Mp][ld] with Mp = M rounded up to a multiple of 128 (two 64-row tiles of the fp32 kernels = one
// 128-point block of the register-resident split-precision kernels) so that tiles never
// need row guards; rows >= M carry finite junk in forward buffers and exact zeros in every adjoint buffer.
// The four [8][Mp][256] stacks of the SDF kernels (WS_S_ACT, WS_S_RHO, WS_S_TAU, WS_S_ZB) are NOT row-major: each [64 x 256] tile
// is stored in accumulator-fragment order (chain_common.h frag_off) because the SDF epilogues load and store them per quad.
// Deformation-network value/JVP buffers have 2 rows per point: row 2p = value, row 2p+1 = tangent along the ray direction d
// (J d); the VJP sweep (J^T g_c) and the backward's tangent sweep (J gbar_o) have 1 row per point.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>

namespace es {

constexpr int PF_DEFORM = 1;   // deformation network present (use_deform)
constexpr int PF_COLOR = 2;    // evaluate the colour network (render_core) — off for errorondepth / surface_neighbour_error
constexpr int PF_SAVE = 4;     // keep activations for the backward pass (training)
constexpr int PF_X3_CHAIN = 32; // OPT-IN: the workspace belongs to the split-precision TRAINING chain (infer_x3r.hip with SAVE / train_x3r.hip): its
                                // ReLU mask words are in that family's layout, so the backward must run that family's kernels.  No effect on offsets
constexpr int PF_RAW_DIR = 16;  // colour-only evaluation on explicit inputs (es_color_forward): the view direction is used as given, not normalised
constexpr int PF_X3 = 8;       // OPT-IN: weight-gradient GEMMs in split precision (3 x bf16 planes, wgrad.hip); no effect on layouts

enum WsBuf : int {
    // forward outputs
    WS_XC, WS_V, WS_SDF, WS_FEAT, WS_GC, WS_GO, WS_RGB,      // WS_V = J d (3 per point)
    // forward saves
    WS_S_ACT,      // [8][Mp][256]  s_1..s_8 (softplus outputs; always written: the SDF reverse sweep needs them); fragment order
    WS_D_U0,       // [2Mp][64]     deform encoding rows (52 valid)
    WS_D_U,        // [8][2Mp][256] u_1..u_8 (for the weight-gradient GEMMs)
    WS_D_MASK,     // [8][Mp/32][256] uint32: ReLU masks of the value rows, one word per (32-point tile, thread) in the
                   //               accumulator-fragment order of deform_fwd (bit 2*quad + {0,1}); always written
    WS_D_R,        // [8][Mp][256]  VJP sweep: adjoints r_0..r_7 of the pre-activations for the covector g_c
    WS_S_S0,       // [Mp][64]      enc6(x_c) (39 valid)
    WS_S_RHO,      // [8][Mp][256]  d sdf / d z_0..7; fragment order
    WS_S_ADJEPS,   // [Mp][64]      d sdf / d enc6(x_c)
    WS_C_IN,       // [Mp][128]     colour input small part (93 valid)
    WS_C_H,        // [8][Mp][256]  h_1..h_8
    WS_C_MASK,     // [8][Mp/64][512] uint32: ReLU masks of h_1..h_8, two words per (tile, thread) in accumulator-fragment
                   //               order (bit 4*quad + row); colour_fwd and colour_bwd share the tile geometry
    // backward buffers
    WS_C_Y,        // [8][Mp][256]  adjoints of colour pre-activations y_0..7
    WS_C_Y8,       // [Mp][4]
    WS_FEATBAR,    // [Mp][256]
    WS_XCBAR_C, WS_GCBAR_C, WS_VBAR_C,     // adjoints of x_c, g_c and v = J d from the colour network
    WS_S_TAU0,     // [Mp][64]
    WS_S_TAU,      // [8][Mp][256]  tau_1..tau_8; fragment order
    WS_S_ZB,       // [8][Mp][256]  second-order terms, overwritten in place by the adjoints of z_0..7; fragment order
    WS_XCBAR,      // [Mp][3]       the adjoint of x_c over all paths (public id ES_WS_XCBAR: the point adjoint reads it, es_point_vjp)
    WS_JU,         // [Mp][3]       J gbar_o (adjoint of g_c through g_o = J^T g_c)
    WS_D_T0,       // [Mp][64]      tangent sweep along gbar_o: encoding tangent (52 valid)
    WS_D_T,        // [8][Mp][256]  tau_1..tau_8
    WS_D_A,        // [8][2Mp][256] adjoints of deform pre-activations a_0..7 (value row, J d row)
    WS_D_A8,       // [2Mp][4]
    WS_C_SBAR,     // [Mp][128]     adjoint of the colour input's small part (93 valid)
    WS_CURV,       // [Mp][3]  c_j = sum_i 4^i (ebar_sin(i,j) sin(2^i x_j) + ebar_cos(i,j) cos(2^i x_j)), ebar = the adjoint of the deformation
                   //          network's encoding input for the covector in WS_GC: minus the diagonal of sum_k g_c[k] d2 x_c[k] / dx^2 (the
                   //          encodings act per coordinate).  Written by the VJP sweep; the second-order term of d g_o / d x (es_point_vjp)
    WS_TBAR,       // [Mp]     the VJP sweep's adjoint of the TIME input: <c, d x_c / d t> for the covector c in WS_GC (public id ES_WS_TBAR)
    WS_COUNT
};
static_assert(WS_XCBAR == 27 && WS_CURV == 34 && WS_TBAR == 35 && WS_VBAR_C == 23,
              "public buffer ids of include/endosurf_hip.h (ES_WS_XCBAR, ES_WS_CURV, ES_WS_TBAR, ES_WS_VBAR)");

struct WsLayout {
    size_t off[WS_COUNT + 1];
    int Mp;
};

inline int round_up64(int m) { return (m + 63) / 64 * 64; }
inline int round_up_rows(int m) { return (m + 127) / 128 * 128; }      // rows of every workspace buffer

inline WsLayout ws_layout(int M, int flags) {
    WsLayout L;
    const size_t Mp = (size_t)round_up_rows(M);
    L.Mp = (int)Mp;
    const bool def = flags & PF_DEFORM, col = flags & PF_COLOR, save = flags & PF_SAVE;
    size_t sz[WS_COUNT] = {0};
    sz[WS_XC] = Mp * 3; sz[WS_V] = Mp * 3; sz[WS_SDF] = Mp; sz[WS_GC] = Mp * 3; sz[WS_GO] = Mp * 3;
    sz[WS_D_MASK] = def ? 8 * (Mp / 32) * 256 : 0;
    sz[WS_CURV] = def ? Mp * 3 : 0;
    sz[WS_TBAR] = def ? Mp : 0;
    sz[WS_FEAT] = col ? Mp * 256 : 0; sz[WS_RGB] = col ? Mp * 3 : 0;
    sz[WS_C_IN] = col ? Mp * 128 : 0;      // always: the colour kernel re-stages it at the skip layer
    sz[WS_S_ACT] = 8 * Mp * 256;
    if (save) {
        if (def) {
            sz[WS_D_U0] = 2 * Mp * 64; sz[WS_D_U] = 8 * 2 * Mp * 256; sz[WS_D_A] = 8 * 2 * Mp * 256; sz[WS_D_A8] = 2 * Mp * 4;
            sz[WS_D_R] = 8 * Mp * 256; sz[WS_JU] = Mp * 3; sz[WS_D_T0] = Mp * 64; sz[WS_D_T] = 8 * Mp * 256;
        }
        sz[WS_S_S0] = Mp * 64; sz[WS_S_RHO] = 8 * Mp * 256; sz[WS_S_ADJEPS] = Mp * 64;
        sz[WS_S_TAU0] = Mp * 64; sz[WS_S_TAU] = 8 * Mp * 256; sz[WS_S_ZB] = 8 * Mp * 256;
        sz[WS_XCBAR] = Mp * 3;
        if (col) {
            sz[WS_C_SBAR] = Mp * 128; sz[WS_C_H] = 8 * Mp * 256; sz[WS_C_MASK] = 8 * (Mp / 64) * 512; sz[WS_C_Y] = 8 * Mp * 256; sz[WS_C_Y8] = Mp * 4;
            sz[WS_FEATBAR] = Mp * 256; sz[WS_XCBAR_C] = Mp * 3; sz[WS_GCBAR_C] = Mp * 3; sz[WS_VBAR_C] = Mp * 3;
        }
    }
    size_t o = 0;
    for (int i = 0; i < WS_COUNT; ++i) { L.off[i] = o; o += (sz[i] + 63) / 64 * 64; }   // keep every buffer 256-B aligned
    L.off[WS_COUNT] = o;
    return L;
}

// a colour-less tail [m_color, M) behind a tile-aligned main part (the fused training batch): see point_fwd.hip
inline bool aux_tail(int flags, int m_color, int M) {
    return (flags & PF_COLOR) && m_color > 0 && m_color < M && m_color % 64 == 0;
}

}  // namespace es
