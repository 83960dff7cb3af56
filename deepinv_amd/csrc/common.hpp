// Common host/device helpers for libdeepinv_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/deepinv_amd.h"

// dynamic LDS of a kernel as a typed pointer (the host emulation used by the CPU tests supplies its own definition)
#ifndef DINV_DYN_LDS
#define DINV_DYN_LDS(T, name)                                                   \
    extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[];  \
    T* name = reinterpret_cast<T*>(name##_raw)
#endif

namespace dinv {

// ---------------------------------------------------------------- error handling
inline char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define DINV_CHECK_HIP(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return ::dinv::fail(100 + (int)_e, "%s failed: %s (%s:%d)", #expr,            \
                                hipGetErrorString(_e), __FILE__, __LINE__);               \
    } while (0)

#define DINV_CHECK_LAUNCH()                                                               \
    do {                                                                                  \
        hipError_t _e = hipGetLastError();                                                \
        if (_e != hipSuccess)                                                             \
            return ::dinv::fail(100 + (int)_e, "kernel launch failed: %s (%s:%d)",        \
                                hipGetErrorString(_e), __FILE__, __LINE__);               \
    } while (0)

#define DINV_REQUIRE(cond, ...)                                                           \
    do {                                                                                  \
        if (!(cond)) return ::dinv::fail(2, __VA_ARGS__);                                 \
    } while (0)

constexpr int kErrArg = 2;

// ---------------------------------------------------------------- small device math
__host__ __device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__host__ __device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__host__ __device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
__host__ __device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__host__ __device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace dinv
