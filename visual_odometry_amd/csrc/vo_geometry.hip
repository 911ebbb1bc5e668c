// Disparity->depth and the keypoint stages (lift, 2-D flow, 3-D flow, per-pair
// correspondences).  All of them are tiny gather/elementwise kernels: one lane
// per pixel or per keypoint, lanes running along the contiguous axis.  The
// arithmetic is float64 wherever the reference's NumPy code is float64, so that
// validity masks come out bit-identical and points agree to rounding.
#include "vo_common.h"

namespace {

struct Mat3 { double m[9]; };

// ------------------------------------------------------------------ depth
// stereo/stereo_depth.py:41-47.  One float4 per lane; the tail is scalar.
template <bool F64>
__device__ __forceinline__ float depth_of(float d, double fxB, float fxB32) {
    float a = fabsf(d);
    if (!(a > 0.0f)) return 0.0f;          // |d| > 0 is false for 0 and for NaN
    if (F64) return (float)(fxB / (double)a);
    return fxB32 / a;
}

template <bool F64>
__global__ __launch_bounds__(256) void depth_kernel(const float* __restrict__ disp,
                                                    float* __restrict__ depth, int64_t n,
                                                    double fxB, float fxB32) {
    int64_t n4 = n >> 2;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool aligned = ((reinterpret_cast<uintptr_t>(disp) | reinterpret_cast<uintptr_t>(depth)) & 15) == 0;
    if (aligned) {
        for (int64_t i = tid; i < n4; i += stride) {
            float4 d = reinterpret_cast<const float4*>(disp)[i];
            float4 o;
            o.x = depth_of<F64>(d.x, fxB, fxB32);
            o.y = depth_of<F64>(d.y, fxB, fxB32);
            o.z = depth_of<F64>(d.z, fxB, fxB32);
            o.w = depth_of<F64>(d.w, fxB, fxB32);
            reinterpret_cast<float4*>(depth)[i] = o;
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += stride) depth[i] = depth_of<F64>(disp[i], fxB, fxB32);
    } else {
        for (int64_t i = tid; i < n; i += stride) depth[i] = depth_of<F64>(disp[i], fxB, fxB32);
    }
}

// --------------------------------------------------------------- keypoints
// NumPy index semantics for a truncated coordinate: negative indices down to
// -n wrap around, anything else outside [0, n) has no element (the reference
// would raise IndexError; the kernels then read nothing and emit zeros).
__device__ __forceinline__ bool np_index(double c, int n, int& idx) {
    if (!(c > -9.0e15 && c < 9.0e15)) { idx = 0; return false; }   // NaN / inf
    long long i = (long long)c;                                     // int(): toward zero
    if (i < 0) i += n;
    idx = (int)i;
    return i >= 0 && i < n;
}

// keypoints/keypoints_3d.py:41-48: Z <= 0 -> zeros, else (K^-1 [u v 1])_{xy} * Z.
__device__ __forceinline__ void lift_point(double u, double v, float zf, const Mat3& Ki,
                                           double& X, double& Y, double& Z) {
#pragma clang fp contract(off)
    if (zf <= 0.0f) { X = 0.0; Y = 0.0; Z = 0.0; return; }
    double z = (double)zf;
    double rx = (Ki.m[0] * u + Ki.m[1] * v) + Ki.m[2];
    double ry = (Ki.m[3] * u + Ki.m[4] * v) + Ki.m[5];
    X = rx * z; Y = ry * z; Z = z;
}

__device__ __forceinline__ double clip(double x, double hi) {   // np.clip(x, 0, hi), NaN stays NaN
    return x < 0.0 ? 0.0 : (x > hi ? hi : x);
}

struct Warp2D { double nx, ny; bool valid; };

// keypoints/keypoints_3d_flow.py:36-45.
__device__ __forceinline__ Warp2D warp_point(double x, double y, const float* flow_b, int H, int W) {
    Warp2D r;
    int ix, iy;
    const bool okx = np_index(x, W, ix), oky = np_index(y, H, iy);
    const bool in = okx && oky;
    double fx = 0.0, fy = 0.0;
    if (in) {
        fx = (double)flow_b[(int64_t)iy * W + ix];
        fy = (double)flow_b[(int64_t)H * W + (int64_t)iy * W + ix];
    }
    double nx = x + fx, ny = y + fy;
    r.valid = in && (nx >= 0.0) && (nx < (double)W) && (ny >= 0.0) && (ny < (double)H);
    r.nx = clip(nx, (double)(W - 1));
    r.ny = clip(ny, (double)(H - 1));
    return r;
}

__device__ __forceinline__ float depth_at(const float* depth_b, double x, double y, int H, int W) {
    int ix, iy;
    const bool okx = np_index(x, W, ix), oky = np_index(y, H, iy);
    if (!(okx && oky)) return 0.0f;
    return depth_b[(int64_t)iy * W + ix];
}

__global__ __launch_bounds__(256) void to3d_kernel(const double* __restrict__ kp, int64_t kp_stride,
                                                   const float* __restrict__ depth, Mat3 Ki,
                                                   double* __restrict__ out, int N, int H, int W) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int b = blockIdx.y;
    if (i >= N) return;
    const double* k = kp + b * kp_stride + 2 * (int64_t)i;
    double x = k[0], y = k[1];
    float z = depth_at(depth + (int64_t)b * H * W, x, y, H, W);
    double X, Y, Z;
    lift_point(x, y, z, Ki, X, Y, Z);
    double* o = out + ((int64_t)b * N + i) * 3;
    o[0] = X; o[1] = Y; o[2] = Z;
}

__global__ __launch_bounds__(256) void flow2d_kernel(const double* __restrict__ kp, int64_t kp_stride,
                                                     const float* __restrict__ flow,
                                                     double* __restrict__ next_out,
                                                     uint8_t* __restrict__ valid_out, int N, int H, int W) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int b = blockIdx.y;
    if (i >= N) return;
    const double* k = kp + b * kp_stride + 2 * (int64_t)i;
    Warp2D w = warp_point(k[0], k[1], flow + (int64_t)b * 2 * H * W, H, W);
    double* o = next_out + ((int64_t)b * N + i) * 2;
    o[0] = w.nx; o[1] = w.ny;
    valid_out[(int64_t)b * N + i] = w.valid ? 1 : 0;
}

// MODE 0: compute_3d_flow (p2, valid, optional p1 under the same mask).
// MODE 1: per-pair correspondences for the tracking loop (p1 of every grid
//         point, p2, weight, count of points inside the depth range).
template <int MODE>
__global__ __launch_bounds__(256) void flow3d_kernel(const double* __restrict__ kp, int64_t kp_stride,
                                                     const float* __restrict__ depth1,
                                                     const float* __restrict__ depth2,
                                                     const float* __restrict__ flow, Mat3 Ki,
                                                     double min_depth, double max_depth, double max_dz,
                                                     double* __restrict__ p1_out, double* __restrict__ p2_out,
                                                     uint8_t* __restrict__ mask_out,
                                                     int32_t* __restrict__ n_in_range, int N, int H, int W) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int b = blockIdx.y;
    bool in_range = false;
    if (i < N) {
        const double* k = kp + b * kp_stride + 2 * (int64_t)i;
        double x = k[0], y = k[1];
        const float* d1m = depth1 + (int64_t)b * H * W;
        const float* d2m = depth2 + (int64_t)b * H * W;
        Warp2D w = warp_point(x, y, flow + (int64_t)b * 2 * H * W, H, W);
        float d1 = depth_at(d1m, x, y, H, W);
        float d2 = depth_at(d2m, w.nx, w.ny, H, W);
        bool ok = w.valid && (d1 > 0.0f) && (d2 > 0.0f);        // keypoints_3d_flow.py:73
        double X1, Y1, Z1, X2 = 0.0, Y2 = 0.0, Z2 = 0.0;
        lift_point(x, y, d1, Ki, X1, Y1, Z1);
        if (ok) lift_point(w.nx, w.ny, d2, Ki, X2, Y2, Z2);
        int64_t o = ((int64_t)b * N + i) * 3;
        p2_out[o] = X2; p2_out[o + 1] = Y2; p2_out[o + 2] = Z2;
        if (MODE == 0) {
            if (p1_out) {
                p1_out[o] = ok ? X1 : 0.0; p1_out[o + 1] = ok ? Y1 : 0.0; p1_out[o + 2] = ok ? Z1 : 0.0;
            }
            mask_out[(int64_t)b * N + i] = ok ? 1 : 0;
        } else {
            p1_out[o] = X1; p1_out[o + 1] = Y1; p1_out[o + 2] = Z1;
            in_range = (Z1 >= min_depth) && (Z1 <= max_depth);   // test_camera_tracking.py:180
            double dz = Z2 - Z1;
            bool dz_ok = fabs(dz) <= max_dz;                      // :197-198
            mask_out[(int64_t)b * N + i] = (in_range && ok && dz_ok) ? 1 : 0;
        }
    }
    if (MODE == 1) {
        unsigned long long votes = __ballot(in_range);
        if ((threadIdx.x & (VO_WAVE - 1)) == 0 && votes) atomicAdd(&n_in_range[b], (int)__popcll(votes));
    }
}

// keypoints/keypoints_3d.py:62-68: (u, v) = (K [X/Z, Y/Z, 1])_{xy}; rows with Z <= 0 have no image.
__device__ __forceinline__ bool project_point(double X, double Y, double Z, const Mat3& K, double& u, double& v) {
#pragma clang fp contract(off)
    if (Z <= 0.0) { u = 0.0; v = 0.0; return false; }
    const double xn = X / Z, yn = Y / Z;
    u = (K.m[0] * xn + K.m[1] * yn) + K.m[2];
    v = (K.m[3] * xn + K.m[4] * yn) + K.m[5];
    return true;
}

__global__ __launch_bounds__(256) void to2d_kernel(const double* __restrict__ pts, Mat3 K, double* __restrict__ uv,
                                                   uint8_t* __restrict__ keep, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double u, v;
    const bool ok = project_point(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], K, u, v);
    uv[2 * i] = u; uv[2 * i + 1] = v;
    keep[i] = ok ? 1 : 0;
}

// Multi-frame keypoint tracks (test_keypoints_flow.py:235-310).  One workgroup per
// SEGMENT of iterations between two scheduled seedings; inside it the iterations are
// sequential (each consumes the previous one's re-projected positions) and the
// keypoint slots run across the lanes.  Slot j is always handled by the same lane,
// so the only cross-lane traffic is the per-iteration count of rows that decides
// the "fewer than min_points rows: seed again" rule (:280).
__global__ __launch_bounds__(256) void track_kernel(const double* __restrict__ grid_kp, const float* __restrict__ depth,
                                                    const float* __restrict__ flow, Mat3 K, Mat3 Ki, int first_index,
                                                    int k, int min_points, double* __restrict__ tracks,
                                                    uint8_t* __restrict__ alive, uint8_t* __restrict__ reseeded,
                                                    int T, int N, int H, int W) {
    __shared__ int wave_count[4];
    const int lead = (k - first_index % k) % k;               // iterations before the first scheduled seeding
    int lo, hi;
    if (lead > 0) {
        if (blockIdx.x == 0) { lo = 0; hi = min(lead, T); }
        else { lo = lead + ((int)blockIdx.x - 1) * k; hi = min(lo + k, T); }
    } else { lo = (int)blockIdx.x * k; hi = min(lo + k, T); }
    const int64_t hw = (int64_t)H * W;
    int rows = 0;                                              // rows that entered the previous iteration's result
    for (int i = lo; i < hi; ++i) {
        const bool seed = (i == lo) || (rows < min_points);
        const float* d1m = depth + (int64_t)i * hw;
        int mine = 0;
        for (int j = threadIdx.x; j < N; j += blockDim.x) {
            const int64_t o = (int64_t)i * N + j;
            double x = 0.0, y = 0.0, z = 0.0;
            bool a = false;
            if (seed) {
                x = grid_kp[2 * j]; y = grid_kp[2 * j + 1];
                const float d = depth_at(d1m, x, y, H, W);
                a = d > 0.0f;                                  // :266 keeps Z > 0
                z = (double)d;
            } else if (alive[o - N]) {
                const double px = tracks[3 * (o - N)], py = tracks[3 * (o - N) + 1];
                const Warp2D w = warp_point(px, py, flow + (int64_t)i * 2 * hw, H, W);
                const float d1 = depth_at(d1m, px, py, H, W);
                const float d2 = depth_at(d1m + hw, w.nx, w.ny, H, W);
                if (w.valid && d1 > 0.0f && d2 > 0.0f) {
                    double X, Y, Z;
                    lift_point(w.nx, w.ny, d2, Ki, X, Y, Z);
                    a = project_point(X, Y, Z, K, x, y);
                    z = Z;
                }
            }
            if (!a) { x = 0.0; y = 0.0; z = 0.0; }
            tracks[3 * o] = x; tracks[3 * o + 1] = y; tracks[3 * o + 2] = z;
            alive[o] = a ? 1 : 0;
            mine += a ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, VO_WAVE);
        __syncthreads();                                       // previous iteration's reads of wave_count are done
        if ((threadIdx.x & (VO_WAVE - 1)) == 0) wave_count[threadIdx.x >> 6] = mine;
        __syncthreads();
        rows = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        if (threadIdx.x == 0) reseeded[i] = seed ? 1 : 0;
    }
}

Mat3 load_mat3(const double* host9) {
    Mat3 m;
    for (int i = 0; i < 9; ++i) m.m[i] = host9[i];
    return m;
}

int check_points_args(const char* fn, const void* kp, int batch, int N, int H, int W) {
    if (!kp && N > 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: kp is null", fn);
    if (batch <= 0 || batch > 65535) return vo_fail(VO_ERR_INVALID_ARG, "%s: batch=%d outside [1,65535]", fn, batch);
    if (N < 0 || H <= 0 || W <= 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: bad sizes N=%d H=%d W=%d", fn, N, H, W);
    return VO_OK;
}

}  // namespace

extern "C" int vo_disparity_to_depth(const float* disparity, float* depth, int64_t n, double fxB,
                                     int divide_in_f64, void* stream) {
    VO_REQUIRE(n >= 0, "vo_disparity_to_depth: n=%lld < 0", (long long)n);
    if (n == 0) return VO_OK;                       // an empty map has no storage to point at
    VO_REQUIRE(disparity && depth, "vo_disparity_to_depth: null pointer");
    int64_t blocks = vo_ceil_div(vo_ceil_div(n, 4), 256);
    if (blocks > 2048) blocks = 2048;               // grid-stride beyond 8 blocks per CU
    if (divide_in_f64)
        depth_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, vo_stream(stream)>>>(disparity, depth, n, fxB, (float)fxB);
    else
        depth_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, vo_stream(stream)>>>(disparity, depth, n, fxB, (float)fxB);
    VO_CHECK_LAUNCH("vo_disparity_to_depth");
    return VO_OK;
}

extern "C" int vo_keypoints_to_3d(const double* kp, int64_t kp_batch_stride, const float* depth,
                                  const double* Kinv_host, double* points_out, int batch, int N, int H,
                                  int W, void* stream) {
    int rc = check_points_args("vo_keypoints_to_3d", kp, batch, N, H, W);
    if (rc) return rc;
    if (N == 0) return VO_OK;
    VO_REQUIRE(depth && Kinv_host && points_out, "vo_keypoints_to_3d: null pointer");
    dim3 grid((unsigned)vo_ceil_div(N, 256), (unsigned)batch);
    to3d_kernel<<<grid, dim3(256), 0, vo_stream(stream)>>>(kp, kp_batch_stride, depth, load_mat3(Kinv_host),
                                                          points_out, N, H, W);
    VO_CHECK_LAUNCH("vo_keypoints_to_3d");
    return VO_OK;
}

extern "C" int vo_keypoints_flow_2d(const double* kp, int64_t kp_batch_stride, const float* flow,
                                    double* next_out, uint8_t* valid_out, int batch, int N, int H, int W,
                                    void* stream) {
    int rc = check_points_args("vo_keypoints_flow_2d", kp, batch, N, H, W);
    if (rc) return rc;
    if (N == 0) return VO_OK;
    VO_REQUIRE(flow && next_out && valid_out, "vo_keypoints_flow_2d: null pointer");
    dim3 grid((unsigned)vo_ceil_div(N, 256), (unsigned)batch);
    flow2d_kernel<<<grid, dim3(256), 0, vo_stream(stream)>>>(kp, kp_batch_stride, flow, next_out, valid_out, N, H, W);
    VO_CHECK_LAUNCH("vo_keypoints_flow_2d");
    return VO_OK;
}

extern "C" int vo_keypoints_flow_3d(const double* kp, int64_t kp_batch_stride, const float* depth1,
                                    const float* depth2, const float* flow, const double* Kinv_host,
                                    double* p2_out, uint8_t* valid_out, double* p1_out, int batch, int N,
                                    int H, int W, void* stream) {
    int rc = check_points_args("vo_keypoints_flow_3d", kp, batch, N, H, W);
    if (rc) return rc;
    if (N == 0) return VO_OK;
    VO_REQUIRE(depth1 && depth2 && flow && Kinv_host && p2_out && valid_out, "vo_keypoints_flow_3d: null pointer");
    dim3 grid((unsigned)vo_ceil_div(N, 256), (unsigned)batch);
    flow3d_kernel<0><<<grid, dim3(256), 0, vo_stream(stream)>>>(kp, kp_batch_stride, depth1, depth2, flow,
                                                               load_mat3(Kinv_host), 0.0, 0.0, 0.0, p1_out,
                                                               p2_out, valid_out, nullptr, N, H, W);
    VO_CHECK_LAUNCH("vo_keypoints_flow_3d");
    return VO_OK;
}

extern "C" int vo_pair_correspondences(const double* kp, int64_t kp_batch_stride, const float* depth1,
                                       const float* depth2, const float* flow, const double* Kinv_host,
                                       double min_depth, double max_depth, double max_dz, double* p1_out,
                                       double* p2_out, uint8_t* weight_out, int32_t* n_in_range_out,
                                       int batch, int N, int H, int W, void* stream) {
    int rc = check_points_args("vo_pair_correspondences", kp, batch, N, H, W);
    if (rc) return rc;
    VO_REQUIRE(n_in_range_out, "vo_pair_correspondences: n_in_range_out is null");
    VO_REQUIRE(N == 0 || (depth1 && depth2 && flow && Kinv_host && p1_out && p2_out && weight_out),
               "vo_pair_correspondences: null pointer");
    hipError_t e = vo_zero_words(n_in_range_out, batch, vo_stream(stream));
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "vo_pair_correspondences: zeroing the counters: %s", hipGetErrorString(e));
    if (N == 0) return VO_OK;
    dim3 grid((unsigned)vo_ceil_div(N, 256), (unsigned)batch);
    flow3d_kernel<1><<<grid, dim3(256), 0, vo_stream(stream)>>>(kp, kp_batch_stride, depth1, depth2, flow,
                                                               load_mat3(Kinv_host), min_depth, max_depth,
                                                               max_dz, p1_out, p2_out, weight_out,
                                                               n_in_range_out, N, H, W);
    VO_CHECK_LAUNCH("vo_pair_correspondences");
    return VO_OK;
}

extern "C" int vo_keypoints_to_2d(const double* points, const double* K_host, double* uv_out, uint8_t* keep_out,
                                  int64_t n, void* stream) {
    VO_REQUIRE(n >= 0, "vo_keypoints_to_2d: n=%lld < 0", (long long)n);
    if (n == 0) return VO_OK;
    VO_REQUIRE(points && K_host && uv_out && keep_out, "vo_keypoints_to_2d: null pointer");
    const int64_t blocks = vo_ceil_div(n, 256);
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_keypoints_to_2d: too many points");
    to2d_kernel<<<dim3((unsigned)blocks), dim3(256), 0, vo_stream(stream)>>>(points, load_mat3(K_host), uv_out, keep_out, n);
    VO_CHECK_LAUNCH("vo_keypoints_to_2d");
    return VO_OK;
}

extern "C" int vo_track_keypoints(const double* grid_kp, const float* depth, const float* flow, const double* K_host,
                                  const double* Kinv_host, int first_index, int k, int min_points, double* tracks_out,
                                  uint8_t* alive_out, uint8_t* reseeded_out, int frames, int N, int H, int W,
                                  void* stream) {
    VO_REQUIRE(frames >= 1 && N >= 0 && H > 0 && W > 0, "vo_track_keypoints: bad sizes frames=%d N=%d H=%d W=%d",
               frames, N, H, W);
    VO_REQUIRE(k >= 1 && first_index >= 0 && min_points >= 0, "vo_track_keypoints: k=%d first_index=%d min_points=%d",
               k, first_index, min_points);
    const int T = frames - 1;
    if (T == 0) return VO_OK;
    VO_REQUIRE(reseeded_out && K_host && Kinv_host, "vo_track_keypoints: null pointer");
    VO_REQUIRE(N == 0 || (grid_kp && depth && flow && tracks_out && alive_out), "vo_track_keypoints: null pointer");
    const int lead = (k - first_index % k) % k;
    const int segments = lead >= T ? 1 : (lead > 0 ? 1 : 0) + (int)vo_ceil_div(T - lead, k);
    track_kernel<<<dim3((unsigned)segments), dim3(256), 0, vo_stream(stream)>>>(
        grid_kp, depth, flow, load_mat3(K_host), load_mat3(Kinv_host), first_index, k, min_points, tracks_out, alive_out,
        reseeded_out, T, N, H, W);
    VO_CHECK_LAUNCH("vo_track_keypoints");
    return VO_OK;
}
