// Argument block of the compositing kernels (mirrors es_composite_args of include/endosurf_hip.h).
#pragma once
namespace es {
struct CompositeArgs {
    const float* rays; const float* z; int ldz; const float* sdf; const float* g_o; const float* rgb;
    const float* variance;   // deviation_network.variance (device scalar)
    int N, S; float sample_dist, cos_anneal;
    // forward outputs
    float* color; float* depth; float* weights; float* cdf; float* weight_max; float* eik_acc;   // eik_acc[2] = {sum relax*err, sum relax}
    int* wmax_idx;
    // backward inputs (adjoints of the forward outputs) and outputs
    const float* g_color; const float* g_depth; const float* g_weights; const float* g_cdf; const float* g_wmax;
    const float* g_gradients_o; const float* g_eik; const float* eik_den;   // g_eik: scalar adj of gradient_o_error; eik_den = sum relax + 1e-6
    float* d_sdf; float* d_go; float* d_rgb; float* d_invs_acc;            // d_invs_acc[1]: adj of inv_s (atomic)
    // deterministic mode (nullable): [N][2] floats; the batch sums (eik_acc / d_invs_acc) are then formed from per-ray partials
    // in a fixed order instead of with fp32 atomics
    float* ray_part;
    // nullable: cos_anneal read from device memory instead (a captured training step changes it between replays)
    const float* cos_anneal_dev;
    // nullable (round 4): forward -> an own-storage copy of the ray samples' g_o rows [N*S][3] (the renderer's ``gradients_o`` output);
    // backward -> n_aux extra rows appended to d_sdf / d_go behind the N*S sample rows: the adjoints of the auxiliary points that were
    // evaluated in the render's launches (NULL pointers = zero rows)
    float* go_copy;
    const float* g_aux_sdf; const float* g_aux_go; int n_aux;
};
}  // namespace es
