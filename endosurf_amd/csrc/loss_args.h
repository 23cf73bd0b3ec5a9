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
    // data-parallel training with EXACT big-batch normalisers, both nullable:
    //   den_out[4]     pass 1 only: write this rank's sums {cmask, inside, valid * mask, n_valid} and return (they are all-reduced next)
    //   den_global[4]  use these (summed over the ranks) instead of the local sums, and scale every term and adjoint by ``world``: the mean
    //                  over the ranks of the per-rank loss / gradient is then the loss / gradient of the concatenated batch
    float* den_out; const float* den_global; float world;
    float* total_out;      // nullable: the total once more, in storage of its own (the differentiable output of the autograd node)
};
}  // namespace es
