// Stereo rectification on the device: the sampling maps, the bilinear remap of uint8
// frames and the 3x3 erosion of the validity mask -- the stage the reference runs with
// OpenCV on the host, one image at a time (stereo/stereo_rectification.py:38-39, 67-68,
// 110-123).  All three are bandwidth kernels: one lane per 4 consecutive output pixels.
#include "vo_common.h"

namespace {

struct RectifyParams {
    double iR[9];        // (P[:3,:3] R)^-1, row-major
    double k[8];         // k1 k2 p1 p2 k3 k4 k5 k6
    double fx, fy, cx, cy;
};

// Plumb-bob model evaluated in float64, stored as float32 (one map pair per camera).
__global__ __launch_bounds__(256) void maps_kernel(RectifyParams p, float* __restrict__ map_x, float* __restrict__ map_y,
                                                   int H, int W) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const double u = (double)(i % W), v = (double)(i / W);
    const double X = (p.iR[0] * u + p.iR[1] * v) + p.iR[2];
    const double Y = (p.iR[3] * u + p.iR[4] * v) + p.iR[5];
    const double Wh = (p.iR[6] * u + p.iR[7] * v) + p.iR[8];
    const double x = X / Wh, y = Y / Wh;
    const double x2 = x * x, y2 = y * y, r2 = x2 + y2, xy2 = 2 * x * y;
    const double kr = (1 + ((p.k[4] * r2 + p.k[1]) * r2 + p.k[0]) * r2) / (1 + ((p.k[7] * r2 + p.k[6]) * r2 + p.k[5]) * r2);
    const double xd = (x * kr + p.k[2] * xy2) + p.k[3] * (r2 + 2 * x2);
    const double yd = (y * kr + p.k[2] * (r2 + 2 * y2)) + p.k[3] * xy2;
    map_x[i] = (float)(p.fx * xd + p.cx);
    map_y[i] = (float)(p.fy * yd + p.cy);
}

__device__ __forceinline__ int tap(const uint8_t* img, int y, int x, int H, int W) {
    return ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? (int)img[(int64_t)y * W + x] : 0;
}

// Fixed-point bilinear remap (5 fractional bits, weights scaled to 2^15, zero border).
// The map is shared by all frames: its fixed-point form is computed once per pixel and
// reused for every frame of the batch.
__global__ __launch_bounds__(256) void remap_kernel(const uint8_t* __restrict__ src, const float* __restrict__ map_x,
                                                    const float* __restrict__ map_y, uint8_t* __restrict__ dst,
                                                    int frames, int H, int W, int Ho, int Wo) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // quad of output pixels
    const int wq = Wo >> 2;
    if (q >= (int64_t)Ho * wq) return;
    const int y = (int)(q / wq), x0 = (int)(q % wq) * 4;
    const int64_t o = (int64_t)y * Wo + x0;
    const float4 mx = *reinterpret_cast<const float4*>(map_x + o), my = *reinterpret_cast<const float4*>(map_y + o);
    const float mxs[4] = {mx.x, mx.y, mx.z, mx.w}, mys[4] = {my.x, my.y, my.z, my.w};
    int ix[4], iy[4], w00[4], w01[4], w10[4], w11[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // round-half-even of map*32; NaN / huge coordinates land outside the image
        const float fx32 = fminf(fmaxf(mxs[i] * 32.0f, -2.0e6f), 2.0e6f), fy32 = fminf(fmaxf(mys[i] * 32.0f, -2.0e6f), 2.0e6f);
        const int sx = (fx32 == fx32) ? __float2int_rn(fx32) : -2000000, sy = (fy32 == fy32) ? __float2int_rn(fy32) : -2000000;
        ix[i] = max(min(sx >> 5, 32767), -32768); iy[i] = max(min(sy >> 5, 32767), -32768);
        const int fx = sx & 31, fy = sy & 31;
        w00[i] = (32 - fy) * (32 - fx) * 32; w01[i] = (32 - fy) * fx * 32;
        w10[i] = fy * (32 - fx) * 32;        w11[i] = fy * fx * 32;
    }
    for (int f = 0; f < frames; ++f) {
        const uint8_t* img = src + (int64_t)f * H * W;
        unsigned packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int acc = w00[i] * tap(img, iy[i], ix[i], H, W) + w01[i] * tap(img, iy[i], ix[i] + 1, H, W) +
                            w10[i] * tap(img, iy[i] + 1, ix[i], H, W) + w11[i] * tap(img, iy[i] + 1, ix[i] + 1, H, W);
            packed |= (unsigned)min(max((acc + (1 << 14)) >> 15, 0), 255) << (8 * i);
        }
        *reinterpret_cast<unsigned*>(dst + (int64_t)f * Ho * Wo + o) = packed;
    }
}

// out = AND of the 3x3 neighbourhood; pixels outside the image do not erode.
__global__ __launch_bounds__(256) void erode_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i % W);
    bool keep = true;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) keep = keep && in[(int64_t)yy * W + xx] != 0;
        }
    out[i] = keep ? 1 : 0;
}

}  // namespace

extern "C" int vo_rectify_maps(const double* K_host, const double* dist_host, int n_dist, const double* R_host,
                               const double* P_host, float* map_x, float* map_y, int H, int W, void* stream) {
    VO_REQUIRE(K_host && R_host && P_host && map_x && map_y, "vo_rectify_maps: null pointer");
    VO_REQUIRE(H > 0 && W > 0 && n_dist >= 0 && n_dist <= 8 && (n_dist == 0 || dist_host),
               "vo_rectify_maps: bad sizes H=%d W=%d n_dist=%d", H, W, n_dist);
    // (P[:3,:3] R)^-1 on the host, by cofactors (P_host is 3x4 row-major)
    double A[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += P_host[r * 4 + k] * R_host[k * 3 + c];
            A[r * 3 + c] = s;
        }
    const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    VO_REQUIRE(det != 0.0 && det == det, "vo_rectify_maps: P[:3,:3] R is singular");
    RectifyParams p;
    p.iR[0] = (A[4] * A[8] - A[5] * A[7]) / det; p.iR[1] = (A[2] * A[7] - A[1] * A[8]) / det; p.iR[2] = (A[1] * A[5] - A[2] * A[4]) / det;
    p.iR[3] = (A[5] * A[6] - A[3] * A[8]) / det; p.iR[4] = (A[0] * A[8] - A[2] * A[6]) / det; p.iR[5] = (A[2] * A[3] - A[0] * A[5]) / det;
    p.iR[6] = (A[3] * A[7] - A[4] * A[6]) / det; p.iR[7] = (A[1] * A[6] - A[0] * A[7]) / det; p.iR[8] = (A[0] * A[4] - A[1] * A[3]) / det;
    for (int i = 0; i < 8; ++i) p.k[i] = i < n_dist ? dist_host[i] : 0.0;
    p.fx = K_host[0]; p.fy = K_host[4]; p.cx = K_host[2]; p.cy = K_host[5];
    const int64_t n = (int64_t)H * W;
    maps_kernel<<<dim3((unsigned)vo_ceil_div(n, 256)), dim3(256), 0, vo_stream(stream)>>>(p, map_x, map_y, H, W);
    VO_CHECK_LAUNCH("vo_rectify_maps");
    return VO_OK;
}

extern "C" int vo_remap_bilinear_u8(const uint8_t* src, const float* map_x, const float* map_y, uint8_t* dst,
                                    int frames, int H, int W, int Ho, int Wo, void* stream) {
    VO_REQUIRE(frames >= 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "vo_remap_bilinear_u8: bad sizes");
    if (frames == 0) return VO_OK;
    VO_REQUIRE(src && map_x && map_y && dst, "vo_remap_bilinear_u8: null pointer");
    if (Wo % 4 != 0) return vo_fail(VO_ERR_UNSUPPORTED, "vo_remap_bilinear_u8: output width %d is not a multiple of 4", Wo);
    VO_REQUIRE(vo_vec4_ok(map_x, 4) && vo_vec4_ok(map_y, 4) && (reinterpret_cast<uintptr_t>(dst) & 3) == 0,
               "vo_remap_bilinear_u8: maps must be 16-byte aligned, dst 4-byte aligned");
    const int64_t quads = (int64_t)Ho * (Wo / 4);
    remap_kernel<<<dim3((unsigned)vo_ceil_div(quads, 256)), dim3(256), 0, vo_stream(stream)>>>(src, map_x, map_y, dst, frames,
                                                                                              H, W, Ho, Wo);
    VO_CHECK_LAUNCH("vo_remap_bilinear_u8");
    return VO_OK;
}

extern "C" int vo_erode3x3_u8(const uint8_t* in, uint8_t* out, int H, int W, void* stream) {
    VO_REQUIRE(in && out && in != out, "vo_erode3x3_u8: null or aliased pointers");
    VO_REQUIRE(H > 0 && W > 0, "vo_erode3x3_u8: bad sizes");
    const int64_t n = (int64_t)H * W;
    erode_kernel<<<dim3((unsigned)vo_ceil_div(n, 256)), dim3(256), 0, vo_stream(stream)>>>(in, out, H, W);
    VO_CHECK_LAUNCH("vo_erode3x3_u8");
    return VO_OK;
}
