// extern "C" surface of libendosurf_hip.so (declared in include/endosurf_hip.h).
#include "../../include/endosurf_hip.h"

#include "arch.h"
#include "chain_common.h"
#include "launch.h"

namespace es {
int weightnorm_pack(const float* params, float* weff, float* packed, int use_deform, hipStream_t st);
int weightnorm_backward(const float* params, const float* dweff, float* dparams, int use_deform, hipStream_t st);
int query_sdf(const PointSrc& src, const float* packed, const float* weff, float* sdf_out, int use_deform, hipStream_t st);

static_assert(sizeof(es_points) == sizeof(PointSrc), "es_points must mirror es::PointSrc");
static inline PointSrc to_src(const es_points* p) {
    PointSrc s;
    s.x = p->x; s.t = p->t; s.dirs = p->dirs; s.rays = p->rays; s.z = p->z;
    s.mode = p->mode; s.t_scalar = p->t_scalar; s.n_per_ray = p->n_per_ray; s.ldz = p->ldz; s.M = p->M;
    return s;
}
static inline int check_src(const es_points* p) {
    ES_REQUIRE(p != nullptr, "es_points is null");
    ES_REQUIRE(p->M >= 0, "negative point count");
    if (p->M == 0) return ST_OK;
    if (p->mode == 0) {
        ES_REQUIRE(p->x && p->t, "mode 0 needs x and t");
    } else {
        ES_REQUIRE(p->mode == 1, "unknown point-source mode");
        ES_REQUIRE(p->rays && p->z && p->n_per_ray > 0 && p->ldz >= p->n_per_ray, "mode 1 needs rays, z, n_per_ray <= ldz");
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
int64_t es_packed_floats(void) { return (int64_t)PACKED_FLOATS; }

int es_weightnorm_pack(const float* params, float* weff, float* packed, int use_deform, void* stream) {
    ES_REQUIRE(params && weff && packed, "null buffer");
    return weightnorm_pack(params, weff, packed, use_deform, (hipStream_t)stream);
}
int es_weightnorm_backward(const float* params, const float* dweff, float* dparams, int use_deform, void* stream) {
    ES_REQUIRE(params && dweff && dparams, "null buffer");
    return weightnorm_backward(params, dweff, dparams, use_deform, (hipStream_t)stream);
}

int es_query_sdf(const es_points* pts, const float* packed, const float* weff, float* sdf_out, int use_deform, void* stream) {
    if (int e = check_src(pts)) return e;
    ES_REQUIRE(packed && weff && (sdf_out || pts->M == 0), "null buffer");
    return query_sdf(to_src(pts), packed, weff, sdf_out, use_deform, (hipStream_t)stream);
}

}  // extern "C"
