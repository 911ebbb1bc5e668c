// Shared host-side helpers for the C-ABI entry points (gfx950 only).
#pragma once

#include <hip/hip_fp16.h>
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

// Small counters and maxima are zeroed by a kernel of their own, not by hipMemsetAsync: recorded into a HIP graph the
// memset of a few words did not take effect in stream order on replay (a captured step replayed on new frames kept
// the previous replay's values; tests/test_pipeline_gpu.py::test_a_step_captures_into_a_hip_graph), a kernel node does.
namespace {
__global__ void vo_zero_words_kernel(unsigned* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
}  // namespace
static inline hipError_t vo_zero_words(void* p, int64_t words, hipStream_t st) {
    if (words <= 0) return hipSuccess;
    vo_zero_words_kernel<<<dim3((unsigned)vo_ceil_div(words, 256)), dim3(256), 0, st>>>(static_cast<unsigned*>(p), words);
    return hipGetLastError();
}

constexpr int VO_WAVE = 64;   // CDNA4 wavefront width

// Activation storage of the fused conv-net passes is float32 or float16 (VO_F32 /
// VO_F16); arithmetic is always float32.  Four elements per access: 16 or 8 bytes.
template <typename T> __device__ __forceinline__ float4 vo_ld4(const T* p, int64_t i);
template <> __device__ __forceinline__ float4 vo_ld4<float>(const float* p, int64_t i) {
    return reinterpret_cast<const float4*>(p)[i];
}
template <> __device__ __forceinline__ float4 vo_ld4<__half>(const __half* p, int64_t i) {
    const uint2 raw = reinterpret_cast<const uint2*>(p)[i];
    const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&raw.x));
    const float2 b = __half22float2(*reinterpret_cast<const __half2*>(&raw.y));
    return make_float4(a.x, a.y, b.x, b.y);
}
template <typename T> __device__ __forceinline__ void vo_st4(T* p, int64_t i, float4 v);
template <> __device__ __forceinline__ void vo_st4<float>(float* p, int64_t i, float4 v) {
    reinterpret_cast<float4*>(p)[i] = v;
}
template <> __device__ __forceinline__ void vo_st4<__half>(__half* p, int64_t i, float4 v) {
    const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
    uint2 raw;
    raw.x = *reinterpret_cast<const unsigned*>(&a);
    raw.y = *reinterpret_cast<const unsigned*>(&b);
    reinterpret_cast<uint2*>(p)[i] = raw;
}
__device__ __forceinline__ float vo_ld1(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float vo_ld1(const __half* p, int64_t i) { return __half2float(p[i]); }
__device__ __forceinline__ void vo_st1(float* p, int64_t i, float v) { p[i] = v; }
__device__ __forceinline__ void vo_st1(__half* p, int64_t i, float v) { p[i] = __float2half_rn(v); }

// True when p may be accessed with vo_ld4 / vo_st4 of element size `esize`.
static inline bool vo_vec4_ok(const void* p, size_t esize) {
    return (reinterpret_cast<uintptr_t>(p) & (4 * esize - 1)) == 0;
}
