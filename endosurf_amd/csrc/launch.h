// Host-side launch helpers shared by the translation units of libendosurf_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

namespace es {

constexpr int ST_OK = 0;
constexpr int ST_BAD_ARG = 1;
constexpr int ST_HIP_ERROR = 2;

inline char* last_error_buf() {
    static char buf[512] = {0};
    return buf;
}
inline int fail(int code, const char* what, const char* detail) {
    snprintf(last_error_buf(), 512, "%s: %s", what, detail ? detail : "");
    return code;
}
inline int hip_last(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ST_HIP_ERROR, what, hipGetErrorString(e));
    return ST_OK;
}
#define ES_HIP(call)                                                                      \
    do {                                                                                  \
        const hipError_t e__ = (call);                                                    \
        if (e__ != hipSuccess) return ::es::fail(::es::ST_HIP_ERROR, #call, hipGetErrorString(e__)); \
    } while (0)
#define ES_REQUIRE(cond, msg)                                             \
    do {                                                                  \
        if (!(cond)) return ::es::fail(::es::ST_BAD_ARG, msg, #cond);     \
    } while (0)

// opt a kernel into > 64 KiB of dynamic LDS (once per kernel)
template <class K>
inline int allow_big_lds(K kernel, int bytes) {
    ES_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return ST_OK;
}

// Per-device "has this one-time setup run yet" flag: hipFuncSetAttribute and __constant__ uploads are per device, so a
// process that drives several GPUs (one Engine per device) must repeat them on each.  Keyed by the CURRENT device
// (callers run under hipSetDevice / torch.cuda.device of their engine).  Not thread-safe: one host thread per device handle.
struct DeviceOnce {
    unsigned long long mask[4] = {0, 0, 0, 0};
    int dev = 0;
    bool first() {
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) dev = 0;
        return !((mask[dev >> 6] >> (dev & 63)) & 1ull);
    }
    void done() { mask[dev >> 6] |= 1ull << (dev & 63); }
};

int init_tables();

}  // namespace es
