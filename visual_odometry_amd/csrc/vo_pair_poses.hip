// Geometry half of the tracking loop for a batch of frame pairs in one call
// (test_camera_tracking.py:176-210): grid keypoints -> lift / flow-warp / filters ->
// weighted Kabsch -> 4x4 relative pose + skip flag.  The intermediate
// correspondences (P1, P2, weights, counts) never leave the device; they live in a
// workspace the caller brings.
#include "vo_pair_poses.h"

namespace {

// test_camera_tracking.py:188-190: fewer than min_points rows after the depth-range filter ->
// the pair is skipped; a pair whose final selection is empty has no pose either.
__global__ __launch_bounds__(256) void skip_flags_kernel(const int32_t* __restrict__ n_in,
                                                         const int32_t* __restrict__ n_used, int min_points,
                                                         uint8_t* __restrict__ skipped, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) skipped[b] = (n_in[b] < min_points || n_used[b] < 1) ? 1 : 0;
}

}  // namespace

int vo_pair_poses_run(const char* name, unsigned char* ws, const double* kp, int64_t kp_batch_stride, const float* depth1,
                      const float* depth2, const float* flow, const double* Kinv_host, double min_depth, double max_depth,
                      double max_dz, const double* offset_host, int min_points, double* T_out, uint8_t* skipped_out,
                      int batch, int N, int H, int W, void* stream) {
    const VoPairPosesLayout l = vo_pair_poses_layout(batch, N);
    double* p1 = reinterpret_cast<double*>(ws + l.p1);
    double* p2 = reinterpret_cast<double*>(ws + l.p2);
    int32_t* n_in = reinterpret_cast<int32_t*>(ws + l.n_in);
    int32_t* n_used = reinterpret_cast<int32_t*>(ws + l.n_used);
    uint8_t* weight = ws + l.weight;
    // clears n_in (the only part of the workspace that must start at zero), then writes
    // every element of P1, P2 and weight the fit reads
    int rc = vo_pair_correspondences(kp, kp_batch_stride, depth1, depth2, flow, Kinv_host, min_depth, max_depth, max_dz,
                                     p1, p2, weight, n_in, batch, N, H, W, stream);
    if (rc != VO_OK) return rc;
    rc = vo_kabsch_batch(p1, p2, weight, offset_host, T_out, n_used, batch, N, stream);
    if (rc != VO_OK) return rc;
    skip_flags_kernel<<<dim3((unsigned)vo_ceil_div(batch, 256)), dim3(256), 0, vo_stream(stream)>>>(
        n_in, n_used, min_points, skipped_out, batch);
    VO_CHECK_LAUNCH(name);
    return VO_OK;
}

int vo_pair_poses_check(const char* name, const double* kp, const float* depth1, const float* depth2, const float* flow,
                        const double* Kinv_host, double* T_out, uint8_t* skipped_out, int batch, int N, int H, int W) {
    if (!(kp && depth1 && depth2 && flow && Kinv_host && T_out && skipped_out))
        return vo_fail(VO_ERR_INVALID_ARG, "%s: null pointer", name);
    if (batch <= 0 || N <= 0 || H <= 0 || W <= 0)
        return vo_fail(VO_ERR_INVALID_ARG, "%s: bad sizes batch=%d N=%d H=%d W=%d", name, batch, N, H, W);
    return VO_OK;
}

extern "C" int64_t vo_pair_poses_workspace_bytes(int batch, int N) {
    if (batch <= 0 || N <= 0) return 0;
    return (int64_t)vo_pair_poses_layout(batch, N).total;
}

extern "C" int vo_pair_poses(const double* kp, int64_t kp_batch_stride, const float* depth1, const float* depth2,
                             const float* flow, const double* Kinv_host, double min_depth, double max_depth,
                             double max_dz, const double* offset_host, int min_points, double* T_out,
                             uint8_t* skipped_out, void* workspace, int64_t workspace_bytes, int batch, int N, int H,
                             int W, void* stream) {
    int rc = vo_pair_poses_check("vo_pair_poses", kp, depth1, depth2, flow, Kinv_host, T_out, skipped_out, batch, N, H, W);
    if (rc != VO_OK) return rc;
    VO_REQUIRE(workspace, "vo_pair_poses: workspace is null");
    VO_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "vo_pair_poses: workspace must be 256-byte aligned");
    const VoPairPosesLayout l = vo_pair_poses_layout(batch, N);
    VO_REQUIRE(workspace_bytes >= (int64_t)l.total, "vo_pair_poses: workspace of %lld bytes, %lld needed",
               (long long)workspace_bytes, (long long)l.total);
    return vo_pair_poses_run("vo_pair_poses", static_cast<unsigned char*>(workspace), kp, kp_batch_stride, depth1, depth2, flow,
                         Kinv_host, min_depth, max_depth, max_dz, offset_host, min_points, T_out, skipped_out, batch, N, H,
                         W, stream);
}
