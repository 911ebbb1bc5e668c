// Batched weighted Kabsch (camera/camera_svd_xform.py:18-48): one 64-lane
// wavefront per frame pair.  Lanes stride over the points for the two reduction
// passes (centroids, then the centred 3x3 covariance, as the reference does),
// wave-reduce with cross-lane shuffles, then every lane runs the same 3x3
// one-sided Jacobi SVD on the wave-uniform covariance (float64 throughout) and
// lane 0 stores the 4x4 pose.
#include "vo_common.h"
#include "vo_svd3.h"

namespace {

struct Vec3 { double v[3]; };

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, VO_WAVE);
    return x;
}

constexpr int PAIRS_PER_BLOCK = 4;   // 4 waves of 64 lanes

__global__ __launch_bounds__(256) void kabsch_kernel(const double* __restrict__ P, const double* __restrict__ Q,
                                                     const uint8_t* __restrict__ wgt, Vec3 offset,
                                                     double* __restrict__ T, int32_t* __restrict__ n_used,
                                                     int B, int N) {
    const int lane = threadIdx.x & (VO_WAVE - 1);
    const int b = blockIdx.x * PAIRS_PER_BLOCK + (threadIdx.x >> 6);
    if (b >= B) return;                                   // whole wave leaves together
    const double* p = P + (int64_t)b * N * 3;
    const double* q = Q + (int64_t)b * N * 3;
    const uint8_t* w = wgt ? wgt + (int64_t)b * N : nullptr;

    double sw = 0.0, sp[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
    for (int i = lane; i < N; i += VO_WAVE) {
        if (w && !w[i]) continue;
        sw += 1.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { sp[k] += p[3 * i + k]; sq[k] += q[3 * i + k]; }
    }
    sw = wave_sum(sw);
    double cp[3], cq[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { cp[k] = wave_sum(sp[k]) / sw; cq[k] = wave_sum(sq[k]) / sw; }

    double h[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = lane; i < N; i += VO_WAVE) {
        if (w && !w[i]) continue;
        double dp[3], dq[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { dp[k] = p[3 * i + k] - cp[k]; dq[k] = q[3 * i + k] - cq[k]; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) h[r][c] += dp[r] * dq[c];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) h[r][c] = wave_sum(h[r][c]);

    double R[3][3];
    rotation_from_covariance(h, R);

    if (lane == 0) {
        double* o = T + (int64_t)b * 16;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double t = cq[r] - (R[r][0] * cp[0] + R[r][1] * cp[1] + R[r][2] * cp[2]);
            t += R[r][0] * offset.v[0] + R[r][1] * offset.v[1] + R[r][2] * offset.v[2];
            o[4 * r + 0] = R[r][0]; o[4 * r + 1] = R[r][1]; o[4 * r + 2] = R[r][2]; o[4 * r + 3] = t;
        }
        o[12] = 0.0; o[13] = 0.0; o[14] = 0.0; o[15] = 1.0;
        if (n_used) n_used[b] = (int32_t)sw;
    }
}

}  // namespace

extern "C" int vo_kabsch_batch(const double* P, const double* Q, const uint8_t* weight,
                               const double* offset_host, double* T_out, int32_t* n_used_out, int batch,
                               int N, void* stream) {
    VO_REQUIRE(P && Q && T_out, "vo_kabsch_batch: null pointer");
    VO_REQUIRE(batch > 0 && N >= 0, "vo_kabsch_batch: bad sizes batch=%d N=%d", batch, N);
    Vec3 off = {{0.0, 0.0, 0.0}};
    if (offset_host) { off.v[0] = offset_host[0]; off.v[1] = offset_host[1]; off.v[2] = offset_host[2]; }
    unsigned blocks = (unsigned)vo_ceil_div(batch, PAIRS_PER_BLOCK);
    kabsch_kernel<<<dim3(blocks), dim3(64 * PAIRS_PER_BLOCK), 0, vo_stream(stream)>>>(P, Q, weight, off, T_out,
                                                                                      n_used_out, batch, N);
    VO_CHECK_LAUNCH("vo_kabsch_batch");
    return VO_OK;
}
