// vo_pair_poses with the workspace inside the library: for callers that must not allocate
// (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
#include <mutex>

#include "vo_pair_poses.h"

namespace {

// Every GPU the library runs on holds its own copy of this array until the process ends.
constexpr int64_t STATIC_MAX_POINTS = VO_PAIR_POSES_STATIC_MAX_POINTS;     // batch * N
constexpr int64_t STATIC_MAX_BATCH = VO_PAIR_POSES_STATIC_MAX_BATCH;
constexpr size_t STATIC_BYTES = 2 * ((size_t)STATIC_MAX_POINTS * 3 * sizeof(double) + 256) +
                                2 * ((size_t)STATIC_MAX_BATCH * sizeof(int32_t) + 256) + (size_t)STATIC_MAX_POINTS + 256;
static __device__ __align__(256) unsigned char scratch[STATIC_BYTES];

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[MAX_DEVICES] = {};      // device address of `scratch` on each GPU, looked up once

// The array's name is a host-side handle, not a device address: the address on the CURRENT device
// comes from hipGetSymbolAddress, once per device.
int scratch_on_current_device(unsigned char** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "vo_pair_poses_static: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= MAX_DEVICES)
        return vo_fail(VO_ERR_UNSUPPORTED, "vo_pair_poses_static: device index %d beyond %d", dev, MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return vo_fail(VO_ERR_LAUNCH, "vo_pair_poses_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
        g_scratch_ptr[dev] = static_cast<unsigned char*>(p);
    }
    *out = g_scratch_ptr[dev];
    return VO_OK;
}

}  // namespace

extern "C" int vo_pair_poses_static(const double* kp, int64_t kp_batch_stride, const float* depth1, const float* depth2,
                                    const float* flow, const double* Kinv_host, double min_depth, double max_depth,
                                    double max_dz, const double* offset_host, int min_points, double* T_out,
                                    uint8_t* skipped_out, int batch, int N, int H, int W, void* stream) {
    int rc = vo_pair_poses_check("vo_pair_poses_static", kp, depth1, depth2, flow, Kinv_host, T_out, skipped_out, batch, N, H, W);
    if (rc != VO_OK) return rc;
    if (batch > STATIC_MAX_BATCH || (int64_t)batch * N > STATIC_MAX_POINTS)
        return vo_fail(VO_ERR_UNSUPPORTED,
                       "vo_pair_poses_static: batch=%d x N=%d exceeds the fixed workspace (at most %lld pairs and %lld "
                       "keypoints in total per call); use vo_pair_poses with a caller workspace",
                       batch, N, (long long)STATIC_MAX_BATCH, (long long)STATIC_MAX_POINTS);
    static_assert(STATIC_BYTES >= 2 * ((size_t)STATIC_MAX_POINTS * 24 + 255) + 2 * ((size_t)STATIC_MAX_BATCH * 4 + 255) +
                                      (size_t)STATIC_MAX_POINTS + 255, "fixed workspace smaller than its largest layout");
    unsigned char* ws = nullptr;
    rc = scratch_on_current_device(&ws);
    if (rc != VO_OK) return rc;
    return vo_pair_poses_run("vo_pair_poses_static", ws, kp, kp_batch_stride, depth1, depth2, flow, Kinv_host, min_depth,
                         max_depth, max_dz, offset_host, min_points, T_out, skipped_out, batch, N, H, W, stream);
}
