// extern "C" surface of libendosurf_hip.so (declared in include/endosurf_hip.h).
#include "../../include/endosurf_hip.h"

#include "arch.h"
#include "chain_common.h"
#include "launch.h"
#include "loss_args.h"
#include "ray_args.h"
#include "workspace.h"

namespace es {
int weightnorm_pack(const float* params, float* weff, float* packed, int use_deform, hipStream_t st);
int weightnorm_backward(const float* params, const float* dweff, float* dparams, int use_deform, hipStream_t st);
int weightnorm_backward_layers(const float* params, const float* dweff, float* dparams, int first_layer, int n_layers, hipStream_t st);
int query_sdf(const PointSrc& src, const float* packed, const float* weff, float* sdf_out, int use_deform, hipStream_t st,
              int ld_out = 0, const int* ray_done = nullptr, int tile_points = 0);
int march_progress(const float* sdf, int N, int n, int n_valid, float tau, int* done, hipStream_t st);
size_t packed_x3_bytes();
int pack_x3(const float* weff, void* packed_x3, int use_deform, hipStream_t st);
int query_sdf_x3(const PointSrc& src, const void* packed_x3, const float* weff, float* sdf_out, int use_deform, hipStream_t st, int ld_out,
                 const int* ray_done);
int variance_terms(const float* variance, const float* d_invs_acc, float* s_val, float* d_var, hipStream_t st);

int point_forward(const PointSrc& src, const float* packed, const float* weff, float* ws, int flags, int m_color, hipStream_t st,
                  const void* packed_x3 = nullptr);
int point_forward_rows(const PointSrc& src, const float* packed, const float* weff, float* ws, int flags, int m_color, int row0, int nrows,
                       hipStream_t st);
int eod_points(const float* rays, const float* depth_gt, const float* mask, int N, float* x, float* t, float* inside, hipStream_t st);
int sn_points(const float* rays, const float* mask, const float* d_i, const float* u, float rad, int N, float* x, float* t, unsigned char* valid,
              hipStream_t st);
int eod_loss(const float* rays, const float* pts, const float* mask, const float* sdf, const float* go, int N, float* out, float* inside, hipStream_t st);
int eod_loss_bwd(const float* rays, const float* inside, const float* sdf, const float* go, const float* out, const float* g_sdf_err,
                 const float* g_ang_err, int N, float* d_sdf, float* d_go, hipStream_t st);
int sn_loss(const float* g, const unsigned char* valid, int N, float* out, hipStream_t st);
int sn_loss_bwd(const float* g, const unsigned char* valid, const float* out, const float* g_loss, int N, float* d_g, hipStream_t st);
int copy2(float* da, const float* sa, long long na, float* db, const float* sb, long long nb, hipStream_t st);
int color_forward(const PointSrc& src, const float* packed, const float* weff, float* ws, hipStream_t st);
int point_vjp(const PointSrc& src, const float* packed, const float* weff, float* ws, int flags, hipStream_t st);
int point_backward_chains(const PointSrc& src, const float* packed, const float* weff, float* ws, int flags, int m_color,
                          const float* d_sdf, const float* d_go, const float* d_rgb, hipStream_t st, const void* packed_x3 = nullptr);
int point_wgrad(int M, float* ws, int flags, int m_color, const float* d_sdf, float* dweff, float* det, hipStream_t st, int net_mask = 7);
size_t wgrad_det_floats();
int gemm_atb(const float* X, const float* dA, int M, float* out, int x3, float* det, hipStream_t st);
int train_loss(const LossArgs& a, hipStream_t st);
int train_aux_points(const float* rays, const float* depth_gt, const float* mask, const float* d_i, const float* u, float rad, int N,
                     float* x, float* t, unsigned char* valid, hipStream_t st);
int train_schedule(double* state, double lr_init, double n_iter, double warm_up_end, double lr_alpha, double beta1, double beta2, float grad_scale,
                   double anneal_end, float* scal, hipStream_t st);
int adam_step_dev(float* p, const float* g, float* m, float* v, long long n, float beta1, float beta2, float eps, const float* scal,
                  const float* g_extra, long long extra_index, hipStream_t st);
int adam_step(float* p, const float* g, float* m, float* v, long long n, float beta1, float beta2, float eps, float step_size,
              float bc2_sqrt, float grad_scale, const float* g_extra, long long extra_index, hipStream_t st);
int uniform(float* out, long long n, unsigned long long seed, unsigned long long subseq, const double* subseq_dev, hipStream_t st);
int scale(float* out, const float* in, long long n, const float* s, hipStream_t st);
int zero(void* p, long long nbytes, hipStream_t st);
int render_finish(const float* eik_acc, const float* aux_sdf_ws, const float* aux_go_ws, int n_aux, float* eik, float* eik_den, float* aux_sdf,
                  float* aux_go, hipStream_t st);
static_assert(sizeof(es_loss_args) == sizeof(LossArgs), "es_loss_args must mirror es::LossArgs");
int ray_setup(const float* rays, const float* u, int N, int n, float sample_dist, int lin_mode, float* z, int ldz, float* near_out,
              float* far_out, hipStream_t st);
int upsample_step(const float* rays, const float* z_in, int ld_in, const float* sdf_in, int ld_sdf, int N, int n, int n_imp,
                  float inv_s, float* z_new, float* z_out, int ld_out, int* src_idx, hipStream_t st);
int merge_sdf(const float* sdf_in, int ld_in, const float* sdf_new, int n_imp, const int* src_idx, int ld_out, int N, int n,
              float* sdf_out, hipStream_t st);
int mid_z(const float* z, int ldz, int N, int S, float sample_dist, float* mid, hipStream_t st);
int composite(const CompositeArgs& a, int backward, hipStream_t st);
int march_find(const float* sdf, const float* dprop, int N, int n, float tau, float* state, int* flags, float* d_pred, hipStream_t st);
int secant_points(const float* rays, const float* d_pred, int N, float* x, float* t, hipStream_t st);
int secant_update(const float* sdf_mid, int N, float tau, float* state, float* d_pred, hipStream_t st);
int march_finish(const float* d_pred, const int* flags, int N, float* d_out, hipStream_t st);
static_assert(sizeof(es_composite_args) == sizeof(CompositeArgs), "es_composite_args must mirror es::CompositeArgs");

static_assert(sizeof(es_points) == sizeof(PointSrc), "es_points must mirror es::PointSrc");
static inline PointSrc to_src(const es_points* p) {
    PointSrc s;
    s.x = p->x; s.t = p->t; s.dirs = p->dirs; s.rays = p->rays; s.z = p->z;
    s.mode = p->mode; s.t_scalar = p->t_scalar; s.n_per_ray = p->n_per_ray; s.ldz = p->ldz; s.M = p->M; s.M_split = p->M_split;
    return s;
}
static inline int check_src(const es_points* p) {
    ES_REQUIRE(p != nullptr, "es_points is null");
    ES_REQUIRE(p->M >= 0, "negative point count");
    if (p->M == 0) return ST_OK;
    if (p->mode == 0) {
        ES_REQUIRE(p->x && p->t, "mode 0 needs x and t");
    } else {
        ES_REQUIRE(p->mode == 1 || p->mode == 2, "unknown point-source mode");
        ES_REQUIRE(p->rays && p->z && p->n_per_ray > 0 && p->ldz >= p->n_per_ray, "modes 1/2 need rays, z, n_per_ray <= ldz");
        if (p->mode == 2) ES_REQUIRE(p->M_split >= 0 && p->M_split <= p->M && (p->M_split == p->M || (p->x && p->t)), "mode 2 needs M_split <= M and x, t");
    }
    return ST_OK;
}
}  // namespace es

using namespace es;

extern "C" {

int es_abi_version(void) { return ES_ABI_VERSION; }
const char* es_last_error(void) { return last_error_buf(); }
int es_init(void) { return init_tables(); }

int64_t es_param_floats(void) { return PARAM_FLOATS; }
int64_t es_param_variance_off(void) { return PARAM_VARIANCE_OFF; }
int es_param_layout(int net, int layer, int64_t* bias_off, int64_t* g_off, int64_t* v_off, int* out_dim, int* in_dim) {
    ES_REQUIRE(net >= 0 && net < NETS && layer >= 0 && layer < LAYERS, "net/layer out of range");
    int64_t off = 0;
    for (int n = 0; n < NETS; ++n)
        for (int l = 0; l < LAYERS; ++l) {
            if (n == net && l == layer) {
                if (bias_off) *bias_off = off;
                if (g_off) *g_off = off + LAYER_N[n][l];
                if (v_off) *v_off = off + 2 * LAYER_N[n][l];
                if (out_dim) *out_dim = LAYER_N[n][l];
                if (in_dim) *in_dim = LAYER_K[n][l];
                return ST_OK;
            }
            off += (int64_t)LAYER_N[n][l] * (2 + LAYER_K[n][l]);
        }
    return ST_BAD_ARG;
}
int64_t es_weff_floats(void) { return WEFF_FLOATS; }
int es_weff_layout(int net, int layer, int64_t* w_off, int64_t* b_off) {
    ES_REQUIRE(net >= 0 && net < NETS && layer >= 0 && layer < LAYERS, "net/layer out of range");
    int64_t off = 0;
    for (int n = 0; n < NETS; ++n)
        for (int l = 0; l < LAYERS; ++l) {
            if (n == net && l == layer) {
                if (w_off) *w_off = off;
                if (b_off) *b_off = off + (int64_t)LAYER_N[n][l] * LAYER_K[n][l];
                return ST_OK;
            }
            off += (int64_t)LAYER_N[n][l] * (1 + LAYER_K[n][l]);
        }
    return ST_BAD_ARG;
}
int64_t es_packed_floats(void) { return (int64_t)PACKED_TOTAL_FLOATS; }

int es_weightnorm_pack(const float* params, float* weff, float* packed, int use_deform, void* stream) {
    ES_REQUIRE(params && weff && packed, "null buffer");
    return weightnorm_pack(params, weff, packed, use_deform, (hipStream_t)stream);
}
int es_weightnorm_backward(const float* params, const float* dweff, float* dparams, int use_deform, void* stream) {
    ES_REQUIRE(params && dweff && dparams, "null buffer");
    return weightnorm_backward(params, dweff, dparams, use_deform, (hipStream_t)stream);
}

int es_weightnorm_backward_layers(const float* params, const float* dweff, float* dparams, int first_layer, int n_layers, void* stream) {
    ES_REQUIRE(params && dweff && dparams, "null buffer");
    ES_REQUIRE(first_layer >= 0 && n_layers >= 0 && first_layer + n_layers <= NETS * LAYERS, "layers [first, first + n) of the 27 (network x 9 + layer)");
    return weightnorm_backward_layers(params, dweff, dparams, first_layer, n_layers, (hipStream_t)stream);
}

int es_query_sdf(const es_points* pts, const float* packed, const float* weff, float* sdf_out, int use_deform, void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(packed && weff && (sdf_out || pts->M == 0), "null buffer");
    return query_sdf(to_src(pts), packed, weff, sdf_out, use_deform, (hipStream_t)stream);
}
int es_query_sdf_tiles(const es_points* pts, const float* packed, const float* weff, float* sdf_out, int use_deform, int tile_points,
                       void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(packed && weff && (sdf_out || pts->M == 0), "null buffer");
    ES_REQUIRE(tile_points == 0 || tile_points == 16 || tile_points == 32 || tile_points == 64, "tile_points: 0 (by batch size), 16, 32 or 64");
    return query_sdf(to_src(pts), packed, weff, sdf_out, use_deform, (hipStream_t)stream, 0, nullptr, tile_points);
}
int es_query_sdf_rays(const es_points* pts, const float* packed, const float* weff, float* sdf_out, int ld_out, const int* ray_done,
                      int use_deform, void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(packed && weff && (sdf_out || pts->M == 0), "null buffer");
    ES_REQUIRE(pts->mode == 1 && pts->n_per_ray >= 1 && ld_out >= pts->n_per_ray, "es_query_sdf_rays takes ray samples (mode 1), ld_out >= n_per_ray");
    return query_sdf(to_src(pts), packed, weff, sdf_out, use_deform, (hipStream_t)stream, ld_out, ray_done);
}
int64_t es_packed_x3_bytes(void) { return (int64_t)packed_x3_bytes(); }
int es_pack_x3(const float* weff, void* packed_x3, int use_deform, void* stream) {
    ES_REQUIRE(weff && packed_x3, "null buffer");
    return pack_x3(weff, packed_x3, use_deform, (hipStream_t)stream);
}
int es_query_sdf_x3(const es_points* pts, const void* packed_x3, const float* weff, float* sdf_out, int ld_out, const int* ray_done,
                    int use_deform, void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(packed_x3 && weff && (sdf_out || pts->M == 0), "null buffer");
    ES_REQUIRE(ld_out == 0 || (pts->mode == 1 && ld_out >= pts->n_per_ray), "ld_out > 0 needs ray samples (mode 1), ld_out >= n_per_ray");
    ES_REQUIRE(ray_done == nullptr || pts->mode == 1, "ray_done needs ray samples (mode 1)");
    return query_sdf_x3(to_src(pts), packed_x3, weff, sdf_out, use_deform, (hipStream_t)stream, ld_out, ray_done);
}
int es_variance_terms(const float* variance, const float* d_invs_acc, float* s_val, float* d_var, void* stream) {
    ES_REQUIRE(variance && (s_val || d_var) && (!d_var || d_invs_acc), "es_variance_terms arguments");
    return variance_terms(variance, d_invs_acc, s_val, d_var, (hipStream_t)stream);
}
int es_march_progress(const float* sdf, int N, int n, int n_valid, float tau, int* done, void* stream) {
    if (N == 0) return ST_OK;          // an empty ray batch: nothing to do, whatever the (possibly null) buffers
    ES_REQUIRE(sdf && done && n >= 2 && n_valid >= 1 && n_valid <= n, "es_march_progress arguments");
    return march_progress(sdf, N, n, n_valid, tau, done, (hipStream_t)stream);
}


int es_ray_setup(const float* rays, const float* u, int N, int n, float sample_dist, int lin_mode, float* z, int ldz,
                 float* near_out, float* far_out, void* stream) {
    if (N == 0) return ST_OK;          // an empty ray batch: nothing to do, whatever the (possibly null) buffers
    ES_REQUIRE(rays && z && N >= 0 && n >= 1 && ldz >= n, "es_ray_setup arguments");
    return ray_setup(rays, u, N, n, sample_dist, lin_mode, z, ldz, near_out, far_out, (hipStream_t)stream);
}
int es_upsample_step(const float* rays, const float* z_in, int ld_in, const float* sdf_in, int ld_sdf, int N, int n, int n_imp,
                     float inv_s, float* z_new, float* z_out, int ld_out, int32_t* src_idx, void* stream) {
    if (N == 0) return ST_OK;          // an empty ray batch: nothing to do, whatever the (possibly null) buffers
    ES_REQUIRE(rays && z_in && sdf_in && z_new && z_out && src_idx && ld_in >= n && ld_sdf >= n, "es_upsample_step arguments");
    return upsample_step(rays, z_in, ld_in, sdf_in, ld_sdf, N, n, n_imp, inv_s, z_new, z_out, ld_out, src_idx, (hipStream_t)stream);
}
int es_merge_sdf(const float* sdf_in, int ld_in, const float* sdf_new, int n_imp, const int32_t* src_idx, int ld_out, int N, int n,
                 float* sdf_out, void* stream) {
    if (N == 0) return ST_OK;          // an empty ray batch: nothing to do, whatever the (possibly null) buffers
    ES_REQUIRE(sdf_in && sdf_new && src_idx && sdf_out && sdf_out != sdf_in, "es_merge_sdf arguments (out must not alias in)");
    return merge_sdf(sdf_in, ld_in, sdf_new, n_imp, src_idx, ld_out, N, n, sdf_out, (hipStream_t)stream);
}
int es_mid_z(const float* z, int ldz, int N, int S, float sample_dist, float* mid, void* stream) {
    if (N == 0) return ST_OK;          // an empty ray batch: nothing to do, whatever the (possibly null) buffers
    ES_REQUIRE(z && mid && ldz >= S && S >= 1, "es_mid_z arguments");
    return mid_z(z, ldz, N, S, sample_dist, mid, (hipStream_t)stream);
}
static inline const CompositeArgs& as_comp(const es_composite_args* a) { return *reinterpret_cast<const CompositeArgs*>(a); }
int es_composite_forward(const es_composite_args* a, void* stream) {
    if (a && a->N == 0) return ST_OK;      // an empty ray batch
    ES_REQUIRE(a && a->rays && a->z && a->sdf && a->g_o && a->rgb && a->variance, "es_composite_forward inputs");
    ES_REQUIRE(a->color && a->depth && a->weights && a->cdf && a->weight_max && a->eik_acc && a->wmax_idx, "es_composite_forward outputs");
    return composite(as_comp(a), 0, (hipStream_t)stream);
}
int es_composite_backward(const es_composite_args* a, void* stream) {
    if (a && a->N == 0) return ST_OK;      // an empty ray batch
    ES_REQUIRE(a && a->rays && a->z && a->sdf && a->g_o && a->rgb && a->variance, "es_composite_backward inputs");
    ES_REQUIRE(a->n_aux >= 0, "es_composite_backward: negative n_aux");
    ES_REQUIRE(a->g_color && a->g_depth && a->g_eik && a->eik_den && a->d_sdf && a->d_go && a->d_rgb && a->d_invs_acc,
               "es_composite_backward adjoints");
    return composite(as_comp(a), 1, (hipStream_t)stream);
}
int es_march_find(const float* sdf, const float* dprop, int N, int n, float tau, float* state, int32_t* flags, float* d_pred, void* stream) {
    if (N == 0) return ST_OK;          // an empty ray batch: nothing to do, whatever the (possibly null) buffers
    ES_REQUIRE(sdf && dprop && state && flags && d_pred && n >= 2, "es_march_find arguments");
    return march_find(sdf, dprop, N, n, tau, state, flags, d_pred, (hipStream_t)stream);
}
int es_secant_points(const float* rays, const float* d_pred, int N, float* x, float* t, void* stream) {
    if (N == 0) return ST_OK;          // an empty ray batch: nothing to do, whatever the (possibly null) buffers
    ES_REQUIRE(rays && d_pred && x && t, "es_secant_points arguments");
    return secant_points(rays, d_pred, N, x, t, (hipStream_t)stream);
}
int es_secant_update(const float* sdf_mid, int N, float tau, float* state, float* d_pred, void* stream) {
    if (N == 0) return ST_OK;          // an empty ray batch: nothing to do, whatever the (possibly null) buffers
    ES_REQUIRE(sdf_mid && state && d_pred, "es_secant_update arguments");
    return secant_update(sdf_mid, N, tau, state, d_pred, (hipStream_t)stream);
}
int es_march_finish(const float* d_pred, const int32_t* flags, int N, float* d_out, void* stream) {
    if (N == 0) return ST_OK;          // an empty ray batch: nothing to do, whatever the (possibly null) buffers
    ES_REQUIRE(d_pred && flags && d_out, "es_march_finish arguments");
    return march_finish(d_pred, flags, N, d_out, (hipStream_t)stream);
}


int64_t es_point_workspace_floats(int M, int flags) { return M <= 0 ? 0 : (int64_t)ws_layout(M, flags).off[WS_COUNT]; }
int64_t es_point_workspace_offset(int M, int flags, int buffer_id) {
    if (M <= 0 || buffer_id < 0 || buffer_id >= WS_COUNT) return -1;
    return (int64_t)ws_layout(M, flags).off[buffer_id];
}
static inline int check_mcolor(const es_points* pts, int flags, int m_color) {
    if (!(flags & ES_PF_COLOR) || m_color <= 0 || m_color == pts->M) return ST_OK;
    ES_REQUIRE(m_color < pts->M && m_color % 64 == 0, "m_color must be a multiple of 64 (tile aligned) or cover all points");
    ES_REQUIRE(pts->mode != 2 || m_color <= pts->M_split, "colour points must be ray samples");
    return ST_OK;
}
int es_point_forward(const es_points* pts, const float* packed, const float* weff, float* ws, int flags, int m_color, void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(packed && weff && (ws || pts->M == 0), "null buffer");
    ES_REQUIRE(!(flags & ES_PF_COLOR) || pts->mode != 0 || pts->dirs, "colour evaluation needs view directions");
    if (int e = check_mcolor(pts, flags, m_color)) return e;
    return point_forward(to_src(pts), packed, weff, ws, flags, m_color, (hipStream_t)stream);
}
int es_point_forward_rows(const es_points* pts, const float* packed, const float* weff, float* ws, int flags, int m_color, int row0, int nrows,
                          void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(packed && weff && (ws || pts->M == 0), "null buffer");
    ES_REQUIRE(!(flags & (ES_PF_X3 | ES_PF_X3_CHAIN)), "es_point_forward_rows: fp32 family only");
    ES_REQUIRE(!(flags & ES_PF_COLOR) || pts->mode != 0 || pts->dirs, "colour evaluation needs view directions");
    ES_REQUIRE(row0 >= 0 && nrows >= 0, "es_point_forward_rows: negative row range");
    if (int e = check_mcolor(pts, flags, m_color)) return e;
    return point_forward_rows(to_src(pts), packed, weff, ws, flags, m_color, row0, nrows, (hipStream_t)stream);
}
int es_eod_points(const float* rays, const float* depth_gt, const float* mask, int N, float* x, float* t, float* inside, void* stream) {
    ES_REQUIRE(N >= 0 && (N == 0 || (rays && depth_gt && x && t)), "es_eod_points buffers");
    ES_REQUIRE(inside == nullptr || mask != nullptr || N == 0, "es_eod_points: the inside mask needs the ray mask");
    return eod_points(rays, depth_gt, mask, N, x, t, inside, (hipStream_t)stream);
}
int es_sn_points(const float* rays, const float* mask, const float* d_i, const float* u, float rad, int N, float* x, float* t,
                 unsigned char* valid, void* stream) {
    ES_REQUIRE(N >= 0 && (N == 0 || (rays && mask && d_i && u && x && t && valid)), "es_sn_points buffers");
    return sn_points(rays, mask, d_i, u, rad, N, x, t, valid, (hipStream_t)stream);
}
int es_eod_loss(const float* rays, const float* pts, const float* mask, const float* sdf, const float* g_o, int N, float* out, float* inside,
                void* stream) {
    ES_REQUIRE(N >= 0 && out && (N == 0 || (rays && pts && mask && sdf && g_o && inside)), "es_eod_loss buffers");
    return eod_loss(rays, pts, mask, sdf, g_o, N, out, inside, (hipStream_t)stream);
}
int es_eod_loss_backward(const float* rays, const float* inside, const float* sdf, const float* g_o, const float* out, const float* g_sdf_err,
                         const float* g_ang_err, int N, float* d_sdf, float* d_go, void* stream) {
    ES_REQUIRE(N >= 0 && (N == 0 || (rays && inside && sdf && g_o && out && d_sdf && d_go)), "es_eod_loss_backward buffers");
    return eod_loss_bwd(rays, inside, sdf, g_o, out, g_sdf_err, g_ang_err, N, d_sdf, d_go, (hipStream_t)stream);
}
int es_sn_loss(const float* g, const unsigned char* valid, int N, float* out, void* stream) {
    ES_REQUIRE(N >= 0 && out && (N == 0 || (g && valid)), "es_sn_loss buffers");
    return sn_loss(g, valid, N, out, (hipStream_t)stream);
}
int es_sn_loss_backward(const float* g, const unsigned char* valid, const float* out, const float* g_loss, int N, float* d_g, void* stream) {
    ES_REQUIRE(N >= 0 && (N == 0 || (g && valid && out && g_loss && d_g)), "es_sn_loss_backward buffers");
    return sn_loss_bwd(g, valid, out, g_loss, N, d_g, (hipStream_t)stream);
}
int es_copy2(float* da, const float* sa, long long na, float* db, const float* sb, long long nb, void* stream) {
    ES_REQUIRE(na >= 0 && nb >= 0 && (na == 0 || (da && sa)) && (nb == 0 || (db && sb)), "es_copy2 buffers");
    return copy2(da, sa, na, db, sb, nb, (hipStream_t)stream);
}
int es_point_vjp(const es_points* pts, const float* packed, const float* weff, float* ws, int flags, void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(packed && weff && (ws || pts->M == 0), "null buffer");
    ES_REQUIRE(flags & ES_PF_DEFORM, "es_point_vjp is the reverse sweep of the deformation network (ES_PF_DEFORM)");
    ES_REQUIRE(!(flags & ES_PF_X3_CHAIN), "the workspace must come from es_point_forward (fp32 family: its ReLU mask words)");
    return point_vjp(to_src(pts), packed, weff, ws, flags, (hipStream_t)stream);
}
int es_color_forward(const es_points* pts, const float* packed, const float* weff, float* ws, void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(packed && weff && (ws || pts->M == 0), "null buffer");
    ES_REQUIRE(pts->mode == 0 && pts->dirs, "es_color_forward takes explicit points with their view directions");
    return color_forward(to_src(pts), packed, weff, ws, (hipStream_t)stream);
}
int es_point_forward_x3(const es_points* pts, const float* packed, const void* packed_x3, const float* weff, float* ws, int flags, int m_color,
                        void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(packed && packed_x3 && weff && (ws || pts->M == 0), "null buffer");
    ES_REQUIRE(!(flags & ES_PF_COLOR) || pts->mode != 0 || pts->dirs, "colour evaluation needs view directions");
    if (int e = check_mcolor(pts, flags, m_color)) return e;
    // no-grad evaluation: ES_PF_X3; with ES_PF_SAVE: the split-precision TRAINING chain (ES_PF_X3_CHAIN; backward: es_point_backward_x3)
    const int mode = (flags & ES_PF_SAVE) ? ES_PF_X3_CHAIN : ES_PF_X3;
    return point_forward(to_src(pts), packed, weff, ws, (flags & ~(ES_PF_X3 | ES_PF_X3_CHAIN)) | mode, m_color, (hipStream_t)stream, packed_x3);
}
int es_point_backward_x3(const es_points* pts, const float* packed, const void* packed_x3, const float* weff, float* ws, int flags, int m_color,
                         const float* d_sdf, const float* d_go, const float* d_rgb, float* dweff, float* wg_scratch, void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(flags & ES_PF_SAVE, "es_point_backward_x3 needs a workspace produced by es_point_forward_x3 with ES_PF_SAVE");
    ES_REQUIRE(packed && packed_x3 && weff && dweff && (pts->M == 0 || (ws && d_sdf && d_go)), "null buffer");
    ES_REQUIRE(!(flags & ES_PF_COLOR) || d_rgb || pts->M == 0, "colour adjoint missing");
    if (int e = check_mcolor(pts, flags, m_color)) return e;
    flags |= ES_PF_X3_CHAIN | ES_PF_X3;          // chain kernels of the split-precision family, weight-gradient GEMMs in split precision
    if (int e = point_backward_chains(to_src(pts), packed, weff, ws, flags, m_color, d_sdf, d_go, d_rgb, (hipStream_t)stream, packed_x3)) return e;
    return point_wgrad(pts->M, ws, flags, m_color, d_sdf, dweff, wg_scratch, (hipStream_t)stream);
}

int es_point_backward(const es_points* pts, const float* packed, const float* weff, float* ws, int flags, int m_color,
                      const float* d_sdf, const float* d_go, const float* d_rgb, float* dweff, void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(flags & ES_PF_SAVE, "es_point_backward needs a workspace produced with ES_PF_SAVE");
    ES_REQUIRE(packed && weff && dweff && (pts->M == 0 || (ws && d_sdf && d_go)), "null buffer");
    ES_REQUIRE(!(flags & ES_PF_COLOR) || d_rgb || pts->M == 0, "colour adjoint missing");
    if (int e = check_mcolor(pts, flags, m_color)) return e;
    if (int e = point_backward_chains(to_src(pts), packed, weff, ws, flags, m_color, d_sdf, d_go, d_rgb, (hipStream_t)stream)) return e;
    return point_wgrad(pts->M, ws, flags, m_color, d_sdf, dweff, nullptr, (hipStream_t)stream);
}
int es_point_backward_stages(const es_points* pts, const float* packed, const float* weff, float* ws, int flags, int m_color,
                             const float* d_sdf, const float* d_go, const float* d_rgb, float* dweff, float* wg_scratch, int stages, void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(flags & ES_PF_SAVE, "es_point_backward_stages needs a workspace produced with ES_PF_SAVE");
    ES_REQUIRE(!(flags & ES_PF_X3_CHAIN), "the staged backward is the fp32 family's (a workspace of es_point_forward)");
    ES_REQUIRE(packed && weff && dweff && (pts->M == 0 || (ws && d_sdf && d_go)), "null buffer");
    ES_REQUIRE(!(flags & ES_PF_COLOR) || d_rgb || pts->M == 0, "colour adjoint missing");
    ES_REQUIRE(stages > 0 && stages < 16, "stages: ES_BWD_CHAINS | ES_BWD_WGRAD_DEFORM | ES_BWD_WGRAD_SDF | ES_BWD_WGRAD_COLOR");
    if (int e = check_mcolor(pts, flags, m_color)) return e;
    if (stages & ES_BWD_CHAINS)
        if (int e = point_backward_chains(to_src(pts), packed, weff, ws, flags, m_color, d_sdf, d_go, d_rgb, (hipStream_t)stream)) return e;
    if (stages >> 1) return point_wgrad(pts->M, ws, flags, m_color, d_sdf, dweff, wg_scratch, (hipStream_t)stream, stages >> 1);
    return ST_OK;
}
int64_t es_wgrad_scratch_floats(void) { return (int64_t)wgrad_det_floats(); }
int es_gemm_atb(const float* X, const float* dA, int M, float* out, int split_precision, float* wg_scratch, void* stream) {
    ES_REQUIRE(X && dA && out && M > 0 && M % 64 == 0, "es_gemm_atb: X [M][256], dA [M][256], out [256][256], M a multiple of 64");
    return gemm_atb(X, dA, M, out, split_precision, wg_scratch, (hipStream_t)stream);
}
int es_point_backward_det(const es_points* pts, const float* packed, const float* weff, float* ws, int flags, int m_color,
                          const float* d_sdf, const float* d_go, const float* d_rgb, float* dweff, float* wg_scratch, void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(flags & ES_PF_SAVE, "es_point_backward needs a workspace produced with ES_PF_SAVE");
    ES_REQUIRE(packed && weff && dweff && (pts->M == 0 || (ws && d_sdf && d_go)), "null buffer");
    ES_REQUIRE(!(flags & ES_PF_COLOR) || d_rgb || pts->M == 0, "colour adjoint missing");
    if (int e = check_mcolor(pts, flags, m_color)) return e;
    if (int e = point_backward_chains(to_src(pts), packed, weff, ws, flags, m_color, d_sdf, d_go, d_rgb, (hipStream_t)stream)) return e;
    return point_wgrad(pts->M, ws, flags, m_color, d_sdf, dweff, wg_scratch, (hipStream_t)stream);
}

int es_train_loss(const es_loss_args* a, void* stream) {
    ES_REQUIRE(a && a->color_map && a->depth_map && a->eik && a->aux_sdf && a->aux_go && a->rays && a->eod_pts && a->color_gt &&
               a->depth_gt && a->mask && a->cmask && a->valid_sn, "es_train_loss inputs");
    ES_REQUIRE(a->den_out || (a->terms && a->g_color && a->g_depth && a->g_eik && a->g_aux_sdf && a->g_aux_go), "es_train_loss outputs");
    ES_REQUIRE(!a->den_global || a->world >= 1.f, "es_train_loss: den_global needs the world size");
    return train_loss(*reinterpret_cast<const LossArgs*>(a), (hipStream_t)stream);
}

int es_zero(void* p, long long nbytes, void* stream) {
    ES_REQUIRE((p || nbytes == 0) && nbytes >= 0, "es_zero arguments");
    return zero(p, nbytes, (hipStream_t)stream);
}
int es_uniform(float* out, long long n, unsigned long long seed, unsigned long long subsequence, const double* subsequence_dev, void* stream) {
    ES_REQUIRE((out || n == 0) && n >= 0, "es_uniform arguments");
    return uniform(out, n, seed, subsequence, subsequence_dev, (hipStream_t)stream);
}
int es_scale(float* out, const float* in, long long n, const float* s, void* stream) {
    ES_REQUIRE(((out && in) || n == 0) && n >= 0 && s, "es_scale arguments");
    return scale(out, in, n, s, (hipStream_t)stream);
}
int es_render_finish(const float* eik_acc, const float* aux_sdf_ws, const float* aux_go_ws, int n_aux, float* eik, float* eik_den, float* aux_sdf,
                     float* aux_go, void* stream) {
    ES_REQUIRE(eik_acc && eik && eik_den && n_aux >= 0 && (n_aux == 0 || (aux_sdf_ws && aux_go_ws && aux_sdf && aux_go)), "es_render_finish arguments");
    return render_finish(eik_acc, aux_sdf_ws, aux_go_ws, n_aux, eik, eik_den, aux_sdf, aux_go, (hipStream_t)stream);
}

int es_train_aux_points(const float* rays, const float* depth_gt, const float* mask, const float* d_i, const float* u, float rad, int N,
                        float* x, float* t, unsigned char* valid, void* stream) {
    if (N == 0) return ST_OK;          // an empty ray batch: nothing to do, whatever the (possibly null) buffers
    ES_REQUIRE(rays && depth_gt && mask && d_i && u && x && t && valid && N >= 0, "es_train_aux_points buffers");
    return train_aux_points(rays, depth_gt, mask, d_i, u, rad, N, x, t, valid, (hipStream_t)stream);
}

int es_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float beta1, float beta2, float eps,
                 float step_size, float bc2_sqrt, float grad_scale, const float* g_extra, long long extra_index, void* stream) {
    ES_REQUIRE(params && grad && exp_avg && exp_avg_sq && n >= 0, "es_adam_step buffers");
    ES_REQUIRE(g_extra == nullptr || (extra_index >= 0 && extra_index < n), "es_adam_step extra gradient index");
    return adam_step(params, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, step_size, bc2_sqrt, grad_scale, g_extra, extra_index,
                     (hipStream_t)stream);
}
int es_train_schedule(double* state, double lr_init, double n_iter, double warm_up_end, double lr_alpha, double beta1, double beta2,
                      float grad_scale, double anneal_end, float* scal, void* stream) {
    ES_REQUIRE(state && scal && n_iter > warm_up_end && warm_up_end >= 0, "es_train_schedule arguments");
    return train_schedule(state, lr_init, n_iter, warm_up_end, lr_alpha, beta1, beta2, grad_scale, anneal_end, scal, (hipStream_t)stream);
}
int es_adam_step_dev(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float beta1, float beta2, float eps,
                     const float* scal, const float* g_extra, long long extra_index, void* stream) {
    ES_REQUIRE(params && grad && exp_avg && exp_avg_sq && scal && n >= 0, "es_adam_step_dev buffers");
    ES_REQUIRE(g_extra == nullptr || (extra_index >= 0 && extra_index < n), "es_adam_step_dev extra gradient index");
    return adam_step_dev(params, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, scal, g_extra, extra_index, (hipStream_t)stream);
}

}  // extern "C"

// ---- whole-stage calls (the per-kernel entry points above chained like the reference's render_rays / render_core) ----
static inline size_t up64(size_t n) { return (n + 63) / 64 * 64; }
int64_t es_sample_scratch_floats(int N, int n_samples, int n_importance, int up_sample_steps) {
    if (N <= 0 || n_samples <= 0) return 0;
    const size_t S = (size_t)n_samples + (size_t)(n_importance > 0 ? n_importance : 0);
    const size_t n_imp = (up_sample_steps > 0 && n_importance > 0) ? (size_t)(n_importance / up_sample_steps) : 0;
    // z ping-pong [N][S], sdf x3 ([N][n_samples], 2 x [N][S]), merge permutation [N][S] (int32), new depths / their sdf [N][n_imp]
    return (int64_t)(up64((size_t)N * S) * 4 + up64((size_t)N * n_samples) + 2 * up64((size_t)N * n_imp));
}
int es_sample_z(const float* rays, const float* u_perturb, int N, int n_samples, int n_importance, int up_sample_steps, int upsample,
                const float* packed, const float* weff, int use_deform, float* z_out, float* scratch, void* stream) {
    if (N == 0) return ST_OK;
    ES_REQUIRE(rays && z_out && packed && weff && N >= 0 && n_samples >= 2, "es_sample_z arguments");
    hipStream_t st = (hipStream_t)stream;
    const bool do_up = upsample && n_importance > 0 && up_sample_steps > 0;
    const int S = n_samples + (do_up ? n_importance : 0);
    const float sample_dist = 2.0f / (float)n_samples;
    if (!do_up) return ray_setup(rays, u_perturb, N, n_samples, sample_dist, 0, z_out, S, nullptr, nullptr, st);
    ES_REQUIRE(scratch != nullptr && n_importance % up_sample_steps == 0, "es_sample_z needs scratch and n_importance divisible by up_sample_steps");
    const int n_imp = n_importance / up_sample_steps;
    const size_t NS = up64((size_t)N * S);
    float* zbuf[2] = {z_out, scratch};                       // ping-pong; the result must land in z_out
    float* sdf_a = scratch + NS;
    float* sdf_b = scratch + 2 * NS;
    int* src = reinterpret_cast<int*>(scratch + 3 * NS);
    float* sdf_c0 = scratch + 4 * NS;
    float* z_new = sdf_c0 + up64((size_t)N * n_samples);
    float* sdf_new = z_new + up64((size_t)N * n_imp);
    int cur = up_sample_steps % 2;                            // so that after up_sample_steps swaps the current buffer is z_out
    if (int e = ray_setup(rays, u_perturb, N, n_samples, sample_dist, 0, zbuf[cur], S, nullptr, nullptr, st)) return e;
    PointSrc ps{};
    ps.rays = rays; ps.mode = 1; ps.t_scalar = 0;
    ps.z = zbuf[cur]; ps.n_per_ray = n_samples; ps.ldz = S; ps.M = N * n_samples;
    if (int e = query_sdf(ps, packed, weff, sdf_c0, use_deform, st)) return e;
    const float* sdf_c = sdf_c0;
    int ld_sdf = n_samples, n = n_samples;
    for (int i = 0; i < up_sample_steps; ++i) {
        if (int e = upsample_step(rays, zbuf[cur], S, sdf_c, ld_sdf, N, n, n_imp, 64.f * (float)(1 << i), z_new, zbuf[cur ^ 1], S, src, st)) return e;
        if (i + 1 != up_sample_steps) {
            ps.z = z_new; ps.n_per_ray = n_imp; ps.ldz = n_imp; ps.M = N * n_imp;
            if (int e = query_sdf(ps, packed, weff, sdf_new, use_deform, st)) return e;
            float* dst = sdf_c != sdf_a ? sdf_a : sdf_b;
            if (int e = merge_sdf(sdf_c, ld_sdf, sdf_new, n_imp, src, S, N, n, dst, st)) return e;
            sdf_c = dst; ld_sdf = S;
        }
        cur ^= 1;
        n += n_imp;
    }
    return ST_OK;
}

int64_t es_render_scratch_floats(int N, int S) {
    if (N <= 0 || S <= 0) return 0;
    const size_t P = (size_t)N * S;
    return (int64_t)(up64(P) + up64(P) + 2 * up64(3 * P));     // mid | d_sdf | d_go | d_rgb
}
static int render_points(const es_render_args* a, PointSrc& ps, int& flags) {
    ES_REQUIRE(a && a->c.rays && a->c.z && a->c.variance && a->ws && a->scratch && a->c.N >= 0 && a->c.S >= 1 && a->c.ldz >= a->c.S,
               "es_render arguments");
    ps = PointSrc{};
    ps.rays = a->c.rays; ps.z = a->scratch; ps.mode = 1; ps.n_per_ray = a->c.S; ps.ldz = a->c.S; ps.M = a->c.N * a->c.S;
    flags = (a->flags & (ES_PF_DEFORM | ES_PF_SAVE | ES_PF_X3)) | ES_PF_COLOR;      // ES_PF_X3: opt-in split-precision weight gradients
    return ST_OK;
}
static CompositeArgs render_composite_args(const es_render_args* a, int flags) {
    CompositeArgs c = as_comp(&a->c);
    const WsLayout L = ws_layout(a->c.N * a->c.S, flags);
    c.sdf = a->ws + L.off[WS_SDF]; c.g_o = a->ws + L.off[WS_GO]; c.rgb = a->ws + L.off[WS_RGB];
    const size_t P = (size_t)a->c.N * a->c.S;
    c.d_sdf = a->scratch + up64(P); c.d_go = c.d_sdf + up64(P); c.d_rgb = c.d_go + up64(3 * P);
    c.n_aux = 0; c.g_aux_sdf = nullptr; c.g_aux_go = nullptr;      // (the whole-stage calls evaluate the ray samples only: scratch holds N*S rows)
    return c;
}
int es_render_forward(const es_render_args* a, const float* packed, const float* weff, void* stream) {
    if (a && a->c.N == 0) return ST_OK;      // an empty ray batch
    PointSrc ps; int flags;
    if (int e = render_points(a, ps, flags)) return e;
    ES_REQUIRE(packed && weff, "null weights");
    ES_REQUIRE(a->c.color && a->c.depth && a->c.weights && a->c.cdf && a->c.weight_max && a->c.eik_acc && a->c.wmax_idx, "es_render_forward outputs");
    if (a->c.N == 0) return ST_OK;
    hipStream_t st = (hipStream_t)stream;
    if (int e = mid_z(a->c.z, a->c.ldz, a->c.N, a->c.S, a->c.sample_dist, a->scratch, st)) return e;
    if ((flags & ES_PF_X3) && (flags & ES_PF_SAVE) && a->packed_x3) flags = (flags & ~ES_PF_X3) | ES_PF_X3_CHAIN;      // training chain
    if (int e = point_forward(ps, packed, weff, a->ws, flags, 0, st, a->packed_x3)) return e;
    return composite(render_composite_args(a, flags), 0, st);
}
int es_render_backward(const es_render_args* a, const float* packed, const float* weff, float* dweff, void* stream) {
    if (a && a->c.N == 0) return ST_OK;      // an empty ray batch
    PointSrc ps; int flags;
    if (int e = render_points(a, ps, flags)) return e;
    ES_REQUIRE(packed && weff && dweff, "null weights / gradient buffer");
    ES_REQUIRE(flags & ES_PF_SAVE, "es_render_backward needs a forward run with ES_PF_SAVE");
    ES_REQUIRE(a->c.g_color && a->c.g_depth && a->c.g_eik && a->c.eik_den && a->c.d_invs_acc, "es_render_backward adjoints");
    if (a->c.N == 0) return ST_OK;
    hipStream_t st = (hipStream_t)stream;
    const CompositeArgs c = render_composite_args(a, flags);
    if (int e = composite(c, 1, st)) return e;
    if ((flags & ES_PF_X3) && a->packed_x3) flags |= ES_PF_X3_CHAIN;       // the forward of the same arguments ran the split-precision chain
    if (int e = point_backward_chains(ps, packed, weff, a->ws, flags, 0, c.d_sdf, c.d_go, c.d_rgb, st, a->packed_x3)) return e;
    return point_wgrad(ps.M, a->ws, flags, 0, c.d_sdf, dweff, a->wg_scratch, st);
}

int64_t es_march_scratch_floats(int N, int n_steps) {
    if (N <= 0 || n_steps <= 0) return 0;
    // proposals + sdf [N][n_steps], bracket state [N][4], flags / done (int32) [N], d_pred [N], secant points x [N][3], t [N], f_mid [N]
    return (int64_t)(2 * up64((size_t)N * n_steps) + up64((size_t)N * 4) + 5 * up64((size_t)N) + up64((size_t)N * 3));
}
int es_ray_marching(const float* rays, int N, int n_steps, int n_secant, float tau, int block, const float* packed, const float* weff,
                    int use_deform, float* d_out, float* scratch, void* stream) {
    if (N == 0) return ST_OK;
    ES_REQUIRE(rays && packed && weff && d_out && N >= 0 && n_steps >= 2 && n_secant >= 0, "es_ray_marching arguments");
    ES_REQUIRE(scratch != nullptr, "es_ray_marching needs scratch");
    hipStream_t st = (hipStream_t)stream;
    const size_t NP = up64((size_t)N * n_steps), N1 = up64((size_t)N);
    float* dprop = scratch;
    float* sdf = dprop + NP;
    float* state = sdf + NP;
    int* flags = reinterpret_cast<int*>(state + up64((size_t)N * 4));
    int* done = flags + N1;
    float* d_pred = reinterpret_cast<float*>(done + N1);
    float* t = d_pred + N1;
    float* f_mid = t + N1;
    float* x = f_mid + N1;
    if (int e = ray_setup(rays, nullptr, N, n_steps, 0.f, 1, dprop, n_steps, nullptr, nullptr, st)) return e;
    PointSrc ps{};
    ps.rays = rays; ps.mode = 1;
    if (block > 0 && n_steps % block == 0 && n_steps > block) {
        if (hipMemsetAsync(sdf, 0, (size_t)N * n_steps * sizeof(float), st) != hipSuccess) return hip_last("es_ray_marching memset");
        for (int b = 0; b < n_steps / block; ++b) {          // skipped proposals read as 0: no sign change
            ps.z = dprop + (size_t)b * block; ps.n_per_ray = block; ps.ldz = n_steps; ps.M = N * block;
            if (int e = query_sdf(ps, packed, weff, sdf + (size_t)b * block, use_deform, st, n_steps, b ? done : nullptr)) return e;
            if (b + 1 < n_steps / block)
                if (int e = march_progress(sdf, N, n_steps, (b + 1) * block, tau, done, st)) return e;
        }
    } else {
        ps.z = dprop; ps.n_per_ray = n_steps; ps.ldz = n_steps; ps.M = N * n_steps;
        if (int e = query_sdf(ps, packed, weff, sdf, use_deform, st)) return e;
    }
    if (int e = march_find(sdf, dprop, N, n_steps, tau, state, flags, d_pred, st)) return e;
    PointSrc pm{};
    pm.x = x; pm.t = t; pm.mode = 0; pm.n_per_ray = 1; pm.ldz = 1; pm.M = N;
    for (int i = 0; i < n_secant; ++i) {
        if (int e = secant_points(rays, d_pred, N, x, t, st)) return e;
        if (int e = query_sdf(pm, packed, weff, f_mid, use_deform, st)) return e;
        if (int e = secant_update(f_mid, N, tau, state, d_pred, st)) return e;
    }
    return march_finish(d_pred, flags, N, d_out, st);
}
