#include "timing.h"

#include <vector>

#include "../../include/endosurf_hip.h"
#include "launch.h"

namespace es {

struct TimedLaunch { int kid; long long rows; hipEvent_t a, b; };
static bool g_timing_on = false;
static std::vector<TimedLaunch> g_launches;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void timing_begin(int kid, long long rows, hipStream_t st) {
    if (!g_timing_on) return;
    TimedLaunch t{kid, rows, get_event(), get_event()};
    if (!t.a || !t.b) return;
    hipEventRecord(t.a, st);
    g_launches.push_back(t);
}
void timing_end(int kid, hipStream_t st) {
    if (!g_timing_on || g_launches.empty()) return;
    TimedLaunch& t = g_launches.back();
    if (t.kid == kid) hipEventRecord(t.b, st);
}

}  // namespace es

using namespace es;

extern "C" {

int es_timing_enable(int on) {
    g_timing_on = on != 0;
    return ST_OK;
}

// Drains the recorded launches (synchronising on their events): for i < min(n, capacity): kid[i], rows[i], ms[i].
int es_timing_drain(int capacity, int* kid, long long* rows, float* ms, int* n_out) {
    int n = 0;
    for (auto& t : g_launches) {
        float v = 0.f;
        if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&v, t.a, t.b) == hipSuccess && n < capacity) {
            kid[n] = t.kid; rows[n] = t.rows; ms[n] = v; ++n;
        }
        g_pool.push_back(t.a); g_pool.push_back(t.b);
    }
    g_launches.clear();
    if (n_out) *n_out = n;
    (void)hipGetLastError();
    return ST_OK;
}

const char* es_kernel_name(int kid) {
    static const char* names[KID_COUNT] = {"k_query_sdf", "k_deform_fwd", "k_sdf_fwd", "k_color_fwd", "k_color_bwd", "k_sdf_bwd",
                                           "k_deform_bwd", "k_wgrad[deform]", "k_wgrad[sdf]", "k_wgrad[color]", "k_wgrad_small",
                                           "k_query_sdf[later marching blocks: tiles of finished rays exit]", "k_deform_vjp", "k_deform_tan", "k_query_sdf16", "k_query_sdf_x3", "k_wgrad_x3[deform]", "k_wgrad_x3[sdf]", "k_wgrad_x3[color]",
                                           "k_deform_fwd_x3", "k_sdf_fwd_x3", "k_color_fwd_x3", "k_deform_vjp_x3",
                                           "k_deform_tan_x3", "k_deform_bwd_x3", "k_color_bwd_x3", "k_sdf_bwd_x3"};
    return kid >= 0 && kid < KID_COUNT ? names[kid] : "?";
}

}  // extern "C"
