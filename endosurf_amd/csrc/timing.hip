#include "timing.h"

#include <vector>

#include "../../include/endosurf_hip.h"
#include "device_table.h"
#include "launch.h"

namespace es {

// One timer state per DEVICE (the launches of a device come from one host thread: include/endosurf_hip.h): switch, recorded launches and
// the pool of events (events belong to the device they were created on).  Two engines on two GPUs of one process time independently.
struct TimedLaunch { int kid; long long rows; hipEvent_t a, b; };
struct TimingState {
    bool on = false;
    std::vector<TimedLaunch> launches;
    std::vector<hipEvent_t> pool;
};
static DeviceTable<TimingState> g_timing;
static int g_any_on = 0;          // number of devices with timers on: the launch path of an untimed process skips the device query

static TimingState& cur_state() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return g_timing.at(dev);
}
static hipEvent_t get_event(TimingState& s) {
    if (!s.pool.empty()) { hipEvent_t e = s.pool.back(); s.pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void timing_begin(int kid, long long rows, hipStream_t st) {
    if (!g_any_on) return;
    TimingState& s = cur_state();
    if (!s.on) return;
    TimedLaunch t{kid, rows, get_event(s), get_event(s)};
    if (!t.a || !t.b) return;
    hipEventRecord(t.a, st);
    s.launches.push_back(t);
}
void timing_end(int kid, hipStream_t st) {
    if (!g_any_on) return;
    TimingState& s = cur_state();
    if (!s.on || s.launches.empty()) return;
    TimedLaunch& t = s.launches.back();
    if (t.kid == kid) hipEventRecord(t.b, st);
}

}  // namespace es

using namespace es;

extern "C" {

// the CURRENT device's timers (callers run under hipSetDevice / torch.cuda.device of their engine)
int es_timing_enable(int on) {
    TimingState& s = cur_state();
    const bool want = on != 0;
    if (want != s.on) g_any_on += want ? 1 : -1;
    s.on = want;
    return ST_OK;
}

// Drains the launches recorded on the current device (synchronising on their events): for i < min(n, capacity): kid[i], rows[i], ms[i].
int es_timing_drain(int capacity, int* kid, long long* rows, float* ms, int* n_out) {
    int n = 0;
    TimingState& s = cur_state();
    auto& g_launches = s.launches;
    auto& g_pool = s.pool;
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
