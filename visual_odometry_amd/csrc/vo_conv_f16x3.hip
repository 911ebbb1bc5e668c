// The split-convolution template (vo_conv_split.h) for float16 pieces: 22 significant bits per operand.
#include "vo_conv_split.h"

namespace vo_f16_pieces {
using Api = SplitApi<PiecesF16>;
int filter(const char* name, const float* weight, void* arranged, int K, int C, int taps, int wx, void* stream) {
    return Api::filter(name, weight, arranged, K, C, taps, wx, stream);
}
int conv3x3(const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale, const float* shift, int relu,
            const float* skip, double* stats, const double* in_stats, float in_eps, int N, int C, int H, int W, int K, int wx, void* stream) {
    return Api::conv3x3(x, arranged, y, y_batch_stride, scale, shift, relu, skip, stats, in_stats, in_eps, N, C, H, W, K, wx, stream);
}
int conv5(const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale, const float* shift, int relu, int N,
          int C, int H, int W, int K, int vertical, int wx, void* stream) {
    return Api::conv5(x, arranged, y, y_batch_stride, scale, shift, relu, N, C, H, W, K, vertical, wx, stream);
}
int conv3x3s2(const float* x, const void* arranged, float* y, const float* scale, const float* shift, int relu, double* stats, int N, int C,
              int H, int W, int K, int wx, void* stream) {
    return Api::conv3x3s2(x, arranged, y, scale, shift, relu, stats, N, C, H, W, K, wx, stream);
}
int conv3x3s2_shortcut(const float* x, const void* arranged, const void* arranged1x1, float* y, float* y2, const float* scale,
                       const float* shift, int relu, double* stats, const float* scale2, const float* shift2, int relu2, double* stats2,
                       int N, int C, int H, int W, int K, int wx, int wx2, void* stream) {
    return Api::conv3x3s2_shortcut(x, arranged, arranged1x1, y, y2, scale, shift, relu, stats, scale2, shift2, relu2, stats2, N, C, H, W, K,
                                   wx, wx2, stream);
}
int conv1x1s2(const float* x, const void* arranged, float* y, const float* scale, const float* shift, int relu, double* stats, int N, int C,
              int H, int W, int K, int wx, void* stream) {
    return Api::conv1x1s2(x, arranged, y, scale, shift, relu, stats, N, C, H, W, K, wx, stream);
}
int conv1x1(const char* name, const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale, const float* shift,
            int relu, int N, int C, int H, int W, int K, int wx, int pixel_major, void* stream) {
    return Api::conv1x1(name, x, arranged, y, y_batch_stride, scale, shift, relu, N, C, H, W, K, wx, pixel_major, stream);
}
int gru_zr(const float* x, const void* arranged, const float* bias, const float* cz, int64_t cz_batch_stride, const float* cr,
           int64_t cr_batch_stride, const float* h, int64_t h_batch_stride, float* z_out, float* rh_out, int N, int Cin, int H, int W, int C,
           int shape, int wx, void* stream) {
    return Api::gru_zr(x, arranged, bias, cz, cz_batch_stride, cr, cr_batch_stride, h, h_batch_stride, z_out, rh_out, N, Cin, H, W, C, shape,
                       wx, stream);
}
int gru_q(const float* x, const float* rh, const void* arranged, const float* bias, const float* cq, int64_t cq_batch_stride, const float* z,
          const float* h, int64_t h_batch_stride, float* h_out, float* h_copy, int64_t copy_batch_stride, int N, int Cin, int H, int W, int C,
          int shape, int wx, void* stream) {
    return Api::gru_q(x, rh, arranged, bias, cq, cq_batch_stride, z, h, h_batch_stride, h_out, h_copy, copy_batch_stride, N, Cin, H, W, C,
                      shape, wx, stream);
}
}  // namespace vo_f16_pieces
