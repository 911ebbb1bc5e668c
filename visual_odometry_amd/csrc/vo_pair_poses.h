// Internals shared by the two pair-pose entry points (caller workspace / fixed workspace).
#pragma once

#include "vo_common.h"

// Workspace layout (offsets in bytes, every block 256-byte aligned):
//   P1 (B*N*3 f64) | P2 (B*N*3 f64) | n_in_range (B i32) | n_used (B i32) | weight (B*N u8)
struct VoPairPosesLayout {
    size_t p1, p2, n_in, n_used, weight, total;
};

static inline size_t vo_up256(size_t v) { return (v + 255) & ~(size_t)255; }

static inline VoPairPosesLayout vo_pair_poses_layout(int64_t batch, int64_t N) {
    VoPairPosesLayout l;
    const size_t pts = vo_up256((size_t)batch * (size_t)N * 3 * sizeof(double));
    const size_t cnt = vo_up256((size_t)batch * sizeof(int32_t));
    l.p1 = 0;
    l.p2 = pts;
    l.n_in = 2 * pts;
    l.n_used = l.n_in + cnt;
    l.weight = l.n_used + cnt;
    l.total = l.weight + vo_up256((size_t)batch * (size_t)N);
    return l;
}

// Argument checks common to both entry points (name = the entry point, for messages).
int vo_pair_poses_check(const char* name, const double* kp, const float* depth1, const float* depth2, const float* flow,
                        const double* Kinv_host, double* T_out, uint8_t* skipped_out, int batch, int N, int H, int W);

// The three launches on `ws`, a device buffer of at least vo_pair_poses_layout(batch, N).total bytes.
int vo_pair_poses_run(const char* name, unsigned char* ws, const double* kp, int64_t kp_batch_stride,
                      const float* depth1, const float* depth2, const float* flow, const double* Kinv_host,
                      double min_depth, double max_depth, double max_dz, const double* offset_host, int min_points,
                      double* T_out, uint8_t* skipped_out, int batch, int N, int H, int W, void* stream);
