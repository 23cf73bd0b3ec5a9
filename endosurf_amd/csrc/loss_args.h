// Argument block of the fused training-loss kernel (mirrors es_loss_args of include/endosurf_hip.h).
#pragma once
namespace es {
struct LossArgs {
    // renderer outputs
    const float* color_map; const float* depth_map; const float* eik; const float* aux_sdf; const float* aux_go;
    // batch
    const float* rays; const float* eod_pts; const float* color_gt; const float* depth_gt; const float* mask; const float* cmask;
    const unsigned char* valid_sn;
    int N;
    float w_color, w_depth, w_sdf, w_angle, w_eik, w_sn;
    // outputs: terms[8] = {color, depth, sdf, angle, eikonal, surf_neig, total, n_valid}; adjoints for d total = 1
    float* terms; float* g_color; float* g_depth; float* g_eik; float* g_aux_sdf; float* g_aux_go;
};
}  // namespace es
