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

int init_tables();

}  // namespace es
