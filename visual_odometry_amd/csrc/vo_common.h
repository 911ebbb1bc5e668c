// Shared host-side helpers for the C-ABI entry points (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vo_hip.h"

// Records a printf-style message for vo_last_error() and returns `code`.
int vo_fail(int code, const char* fmt, ...);

#define VO_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return vo_fail(VO_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)

// Call right after a kernel launch: turns a rejected launch into VO_ERR_LAUNCH.
#define VO_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess)                                                        \
            return vo_fail(VO_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_));    \
    } while (0)

static inline hipStream_t vo_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t vo_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int VO_WAVE = 64;   // CDNA4 wavefront width
