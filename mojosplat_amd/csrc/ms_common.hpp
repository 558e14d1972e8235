// Shared host-side plumbing for libmojosplat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mojosplat_hip.h"

namespace ms {

void set_error(const char *fmt, ...);

#define MS_REQUIRE(cond, code, ...)  \
    do {                             \
        if (!(cond)) {               \
            ms::set_error(__VA_ARGS__); \
            return (code);           \
        }                            \
    } while (0)

#define MS_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            ms::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                   \
            return MS_ERR_HIP;                                                         \
        }                                                                              \
    } while (0)

// after a kernel launch: catches bad launch configurations without synchronising
#define MS_LAUNCH_CHECK() MS_HIP(hipGetLastError())

constexpr float kAlphaThreshold = 1.0f / 255.0f;
constexpr float kMaxAlpha = 0.999f;
constexpr float kTransmittanceStop = 1e-4f;

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace ms
