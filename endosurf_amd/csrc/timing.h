// Opt-in per-kernel timing with HIP events on the launch stream (used by bench.py for the roofline line).
// Disabled by default: es_timing_enable(0) => zero overhead and graph-capturable launches.
#pragma once
#include <hip/hip_runtime.h>

namespace es {

enum KernelId : int {
    KID_QUERY = 0, KID_DEFORM_FWD, KID_SDF_FWD, KID_COLOR_FWD, KID_COLOR_BWD, KID_SDF_BWD, KID_DEFORM_BWD,
    KID_WGRAD_D, KID_WGRAD_S, KID_WGRAD_C, KID_WGRAD_SMALL, KID_QUERY_EXIT, KID_DEFORM_VJP, KID_DEFORM_TAN, KID_QUERY16, KID_QUERY_X3, KID_WGRAD_D_X3, KID_WGRAD_S_X3, KID_WGRAD_C_X3,
    KID_DEFORM_FWD_X3, KID_SDF_FWD_X3, KID_COLOR_FWD_X3, KID_DEFORM_VJP_X3,
    KID_DEFORM_TAN_X3, KID_DEFORM_BWD_X3, KID_COLOR_BWD_X3, KID_SDF_BWD_X3, KID_COUNT
};

void timing_begin(int kid, long long rows, hipStream_t st);
void timing_end(int kid, hipStream_t st);

struct ScopedTimer {
    int kid; hipStream_t st;
    ScopedTimer(int k, long long rows, hipStream_t s) : kid(k), st(s) { timing_begin(k, rows, s); }
    ~ScopedTimer() { timing_end(kid, st); }
};

}  // namespace es
