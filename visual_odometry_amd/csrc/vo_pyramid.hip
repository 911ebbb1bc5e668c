// Correlation pyramid: levels 1..3 from level 0 in ONE pass over level 0
// (stock implementations launch one avg_pool per level and re-read each level).
// HBM-bound: level 0 is read exactly once, levels 1..3 are written once.
//
// 2-D: each lane owns an 8x8 block of one source pixel's (H2 x W2) slice and
//      produces its 4x4 / 2x2 / 1x1 descendants in registers.  Lanes run along
//      x, so a wave reads runs of consecutive 32-byte segments of each row.
// 1-D: each lane owns 8 consecutive floats of one row (pool along x only).
// Level sizes floor-halve (avg_pool, kernel 2, stride 2, no ceil_mode), so a
// trailing odd row/column simply has no descendant.
#include "vo_common.h"

namespace {

struct Dims2 { int h[4], w[4]; };

__global__ __launch_bounds__(256) void pyramid2d_kernel(const float* __restrict__ l0, float* __restrict__ l1,
                                                        float* __restrict__ l2, float* __restrict__ l3,
                                                        int64_t total, Dims2 d, int hb, int wb, int vec_ok) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int bx = (int)(t % wb);
    const int by = (int)((t / wb) % hb);
    const int64_t img = t / ((int64_t)wb * hb);
    const float* src = l0 + img * d.h[0] * d.w[0];

    float a[4][4];   // level-1 block
    const bool full = (4 * bx + 4 <= d.w[1]) && (4 * by + 4 <= d.h[1]);
    if (full && vec_ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* p = src + (int64_t)(8 * by + 2 * r) * d.w[0] + 8 * bx;
            const float4 t0 = *reinterpret_cast<const float4*>(p);
            const float4 t1 = *reinterpret_cast<const float4*>(p + 4);
            const float4 u0 = *reinterpret_cast<const float4*>(p + d.w[0]);
            const float4 u1 = *reinterpret_cast<const float4*>(p + d.w[0] + 4);
            a[r][0] = (((t0.x + t0.y) + u0.x) + u0.y) * 0.25f;
            a[r][1] = (((t0.z + t0.w) + u0.z) + u0.w) * 0.25f;
            a[r][2] = (((t1.x + t1.y) + u1.x) + u1.y) * 0.25f;
            a[r][3] = (((t1.z + t1.w) + u1.z) + u1.w) * 0.25f;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int y = 4 * by + r, x = 4 * bx + c;
                float v = 0.0f;
                if (y < d.h[1] && x < d.w[1]) {
                    const float* p = src + (int64_t)(2 * y) * d.w[0] + 2 * x;
                    v = (((p[0] + p[1]) + p[d.w[0]]) + p[d.w[0] + 1]) * 0.25f;
                }
                a[r][c] = v;
            }
    }
    if (l1) {
        float* dst = l1 + img * d.h[1] * d.w[1];
        if (full && vec_ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<float4*>(dst + (int64_t)(4 * by + r) * d.w[1] + 4 * bx) =
                    make_float4(a[r][0], a[r][1], a[r][2], a[r][3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (4 * by + r < d.h[1] && 4 * bx + c < d.w[1]) dst[(int64_t)(4 * by + r) * d.w[1] + 4 * bx + c] = a[r][c];
        }
    }
    if (!l2) return;
    float b[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
            b[r][c] = (((a[2 * r][2 * c] + a[2 * r][2 * c + 1]) + a[2 * r + 1][2 * c]) + a[2 * r + 1][2 * c + 1]) * 0.25f;
    {
        float* dst = l2 + img * d.h[2] * d.w[2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (2 * by + r < d.h[2] && 2 * bx + c < d.w[2]) dst[(int64_t)(2 * by + r) * d.w[2] + 2 * bx + c] = b[r][c];
    }
    if (!l3) return;
    if (by < d.h[3] && bx < d.w[3])
        l3[img * d.h[3] * d.w[3] + (int64_t)by * d.w[3] + bx] = (((b[0][0] + b[0][1]) + b[1][0]) + b[1][1]) * 0.25f;
}

__global__ __launch_bounds__(256) void pyramid1d_kernel(const float* __restrict__ l0, float* __restrict__ l1,
                                                        float* __restrict__ l2, float* __restrict__ l3,
                                                        int64_t total, int w0, int w1, int w2, int w3, int wb,
                                                        int vec_ok) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int bx = (int)(t % wb);
    const int64_t row = t / wb;
    const float* src = l0 + row * w0 + 8 * bx;
    float a[4];
    const bool full = 4 * bx + 4 <= w1;
    if (full && vec_ok) {
        const float4 t0 = *reinterpret_cast<const float4*>(src);
        const float4 t1 = *reinterpret_cast<const float4*>(src + 4);
        a[0] = (t0.x + t0.y) * 0.5f; a[1] = (t0.z + t0.w) * 0.5f;
        a[2] = (t1.x + t1.y) * 0.5f; a[3] = (t1.z + t1.w) * 0.5f;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = (4 * bx + c < w1) ? (src[2 * c] + src[2 * c + 1]) * 0.5f : 0.0f;
    }
    if (l1) {
        float* dst = l1 + row * w1 + 4 * bx;
        if (full && vec_ok) {
            *reinterpret_cast<float4*>(dst) = make_float4(a[0], a[1], a[2], a[3]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (4 * bx + c < w1) dst[c] = a[c];
        }
    }
    if (!l2) return;
    const float b0 = (a[0] + a[1]) * 0.5f, b1 = (a[2] + a[3]) * 0.5f;
    if (2 * bx < w2) l2[row * w2 + 2 * bx] = b0;
    if (2 * bx + 1 < w2) l2[row * w2 + 2 * bx + 1] = b1;
    if (!l3) return;
    if (bx < w3) l3[row * w3 + bx] = (b0 + b1) * 0.5f;
}

}  // namespace

extern "C" int vo_corr_pyramid_2d(const float* l0, float* l1, float* l2, float* l3, int64_t images, int H2, int W2,
                                  void* stream) {
    VO_REQUIRE(l0, "vo_corr_pyramid_2d: level 0 is null");
    VO_REQUIRE(images > 0 && H2 > 0 && W2 > 0, "vo_corr_pyramid_2d: bad sizes images=%lld H2=%d W2=%d",
               (long long)images, H2, W2);
    VO_REQUIRE(!(l2 && !l1) && !(l3 && !l2), "vo_corr_pyramid_2d: levels must be given without gaps");
    if (!l1) return VO_OK;
    Dims2 d;
    d.h[0] = H2; d.w[0] = W2;
    for (int i = 1; i < 4; ++i) { d.h[i] = d.h[i - 1] / 2; d.w[i] = d.w[i - 1] / 2; }
    const int hb = (int)vo_ceil_div(H2, 8), wb = (int)vo_ceil_div(W2, 8);
    const int64_t total = images * hb * wb;
    const int64_t blocks = vo_ceil_div(total, 256);
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_corr_pyramid_2d: grid too large");
    // 16-byte vector path: every level-0 row and level-1 row start must be 16-byte aligned
    const int vec_ok = (W2 % 4 == 0) && (d.w[1] % 4 == 0) &&
                       ((reinterpret_cast<uintptr_t>(l0) | reinterpret_cast<uintptr_t>(l1)) & 15) == 0;
    pyramid2d_kernel<<<dim3((unsigned)blocks), dim3(256), 0, vo_stream(stream)>>>(l0, l1, l2, l3, total, d, hb, wb, vec_ok);
    VO_CHECK_LAUNCH("vo_corr_pyramid_2d");
    return VO_OK;
}

extern "C" int vo_corr_pyramid_1d(const float* l0, float* l1, float* l2, float* l3, int64_t rows, int W2,
                                  void* stream) {
    VO_REQUIRE(l0, "vo_corr_pyramid_1d: level 0 is null");
    VO_REQUIRE(rows > 0 && W2 > 0, "vo_corr_pyramid_1d: bad sizes rows=%lld W2=%d", (long long)rows, W2);
    VO_REQUIRE(!(l2 && !l1) && !(l3 && !l2), "vo_corr_pyramid_1d: levels must be given without gaps");
    if (!l1) return VO_OK;
    const int w1 = W2 / 2, w2 = w1 / 2, w3 = w2 / 2;
    const int wb = (int)vo_ceil_div(W2, 8);
    const int64_t total = rows * wb;
    const int64_t blocks = vo_ceil_div(total, 256);
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_corr_pyramid_1d: grid too large");
    const int vec_ok = (W2 % 4 == 0) && (w1 % 4 == 0) &&
                       ((reinterpret_cast<uintptr_t>(l0) | reinterpret_cast<uintptr_t>(l1)) & 15) == 0;
    pyramid1d_kernel<<<dim3((unsigned)blocks), dim3(256), 0, vo_stream(stream)>>>(l0, l1, l2, l3, total, W2, w1, w2, w3,
                                                                                 wb, vec_ok);
    VO_CHECK_LAUNCH("vo_corr_pyramid_1d");
    return VO_OK;
}
