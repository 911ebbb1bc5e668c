// Entry points of the split convolutions (template and arithmetic: vo_conv_split.h).  ``pieces`` picks the 16-bit
// format a float32 operand is cut into: VO_PIECES_BF16 (0) or VO_PIECES_F16 (1); ``weight_exp`` is the power of two
// the weights were multiplied by before they were cut (float16 pieces only; 0 for bf16), the value handed to the
// matching *_filter call.
#include "vo_conv_split.h"

namespace {
using Bf = SplitApi<PiecesBf16>;
inline int bad_pieces(const char* name, int pieces, int wx) {
    if (pieces != VO_PIECES_BF16 && pieces != VO_PIECES_F16) return vo_fail(VO_ERR_INVALID_ARG, "%s: pieces must be VO_PIECES_BF16 or VO_PIECES_F16", name);
    if (pieces == VO_PIECES_BF16 && wx != 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: weight_exp must be 0 with bf16 pieces", name);
    return VO_OK;
}
}  // namespace
namespace F16 = vo_f16_pieces;

#define VO_PIECES_CHECK(name, pieces, wx) do { int rc_ = bad_pieces(name, pieces, wx); if (rc_ != VO_OK) return rc_; } while (0)

extern "C" int vo_conv3x3_bf16x3_filter(const float* weight, void* arranged, int K, int C, int pieces, int weight_exp, void* stream) {
    VO_PIECES_CHECK("vo_conv3x3_bf16x3_filter", pieces, weight_exp);
    return pieces ? F16::filter("vo_conv3x3_bf16x3_filter", weight, arranged, K, C, 9, weight_exp, stream)
                  : Bf::filter("vo_conv3x3_bf16x3_filter", weight, arranged, K, C, 9, 0, stream);
}

extern "C" int vo_conv3x3_bf16x3(const float* x, const void* arranged, float* y, int64_t y_batch_stride,
                                 const float* scale, const float* shift, int relu, const float* skip, double* stats,
                                 const double* in_stats, float in_eps, int N, int C, int H, int W, int K, int pieces, int weight_exp,
                                 void* stream) {
    VO_PIECES_CHECK("vo_conv3x3_bf16x3", pieces, weight_exp);
    return pieces ? F16::conv3x3(x, arranged, y, y_batch_stride, scale, shift, relu, skip, stats, in_stats, in_eps, N, C, H, W, K, weight_exp, stream)
                  : Bf::conv3x3(x, arranged, y, y_batch_stride, scale, shift, relu, skip, stats, in_stats, in_eps, N, C, H, W, K, 0, stream);
}

extern "C" int vo_conv5_bf16x3_filter(const float* weight, void* arranged, int K, int C, int pieces, int weight_exp, void* stream) {
    VO_PIECES_CHECK("vo_conv5_bf16x3_filter", pieces, weight_exp);
    return pieces ? F16::filter("vo_conv5_bf16x3_filter", weight, arranged, K, C, 5, weight_exp, stream)
                  : Bf::filter("vo_conv5_bf16x3_filter", weight, arranged, K, C, 5, 0, stream);
}

extern "C" int vo_conv5_bf16x3(const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale,
                               const float* shift, int relu, int N, int C, int H, int W, int K, int vertical, int pieces, int weight_exp,
                               void* stream) {
    VO_PIECES_CHECK("vo_conv5_bf16x3", pieces, weight_exp);
    return pieces ? F16::conv5(x, arranged, y, y_batch_stride, scale, shift, relu, N, C, H, W, K, vertical, weight_exp, stream)
                  : Bf::conv5(x, arranged, y, y_batch_stride, scale, shift, relu, N, C, H, W, K, vertical, 0, stream);
}

extern "C" int vo_conv3x3s2_bf16x3(const float* x, const void* arranged, float* y, const float* scale, const float* shift,
                                   int relu, double* stats, int N, int C, int H, int W, int K, int pieces, int weight_exp, void* stream) {
    VO_PIECES_CHECK("vo_conv3x3s2_bf16x3", pieces, weight_exp);
    return pieces ? F16::conv3x3s2(x, arranged, y, scale, shift, relu, stats, N, C, H, W, K, weight_exp, stream)
                  : Bf::conv3x3s2(x, arranged, y, scale, shift, relu, stats, N, C, H, W, K, 0, stream);
}

extern "C" int vo_conv3x3s2_shortcut_bf16x3(const float* x, const void* arranged, const void* arranged1x1, float* y, float* y2,
                                            const float* scale, const float* shift, int relu, double* stats,
                                            const float* scale2, const float* shift2, int relu2, double* stats2,
                                            int N, int C, int H, int W, int K, int pieces, int weight_exp, int weight_exp_1x1, void* stream) {
    VO_PIECES_CHECK("vo_conv3x3s2_shortcut_bf16x3", pieces, weight_exp);
    VO_PIECES_CHECK("vo_conv3x3s2_shortcut_bf16x3", pieces, weight_exp_1x1);
    return pieces ? F16::conv3x3s2_shortcut(x, arranged, arranged1x1, y, y2, scale, shift, relu, stats, scale2, shift2, relu2, stats2, N, C, H,
                                            W, K, weight_exp, weight_exp_1x1, stream)
                  : Bf::conv3x3s2_shortcut(x, arranged, arranged1x1, y, y2, scale, shift, relu, stats, scale2, shift2, relu2, stats2, N, C, H,
                                           W, K, 0, 0, stream);
}

extern "C" int vo_conv1x1_bf16x3_filter(const float* weight, void* arranged, int K, int C, int pieces, int weight_exp, void* stream) {
    VO_PIECES_CHECK("vo_conv1x1_bf16x3_filter", pieces, weight_exp);
    return pieces ? F16::filter("vo_conv1x1_bf16x3_filter", weight, arranged, K, C, 1, weight_exp, stream)
                  : Bf::filter("vo_conv1x1_bf16x3_filter", weight, arranged, K, C, 1, 0, stream);
}

extern "C" int vo_conv1x1s2_bf16x3(const float* x, const void* arranged, float* y, const float* scale, const float* shift,
                                   int relu, double* stats, int N, int C, int H, int W, int K, int pieces, int weight_exp, void* stream) {
    VO_PIECES_CHECK("vo_conv1x1s2_bf16x3", pieces, weight_exp);
    return pieces ? F16::conv1x1s2(x, arranged, y, scale, shift, relu, stats, N, C, H, W, K, weight_exp, stream)
                  : Bf::conv1x1s2(x, arranged, y, scale, shift, relu, stats, N, C, H, W, K, 0, stream);
}

extern "C" int vo_gru_conv_zr_bf16x3(const float* x, const void* arranged, const float* bias, const float* cz, int64_t cz_batch_stride,
                                     const float* cr, int64_t cr_batch_stride, const float* h, int64_t h_batch_stride, float* z_out,
                                     float* rh_out, int N, int Cin, int H, int W, int C, int shape, int pieces, int weight_exp, void* stream) {
    VO_PIECES_CHECK("vo_gru_conv_zr_bf16x3", pieces, weight_exp);
    return pieces ? F16::gru_zr(x, arranged, bias, cz, cz_batch_stride, cr, cr_batch_stride, h, h_batch_stride, z_out, rh_out, N, Cin, H, W, C,
                                shape, weight_exp, stream)
                  : Bf::gru_zr(x, arranged, bias, cz, cz_batch_stride, cr, cr_batch_stride, h, h_batch_stride, z_out, rh_out, N, Cin, H, W, C,
                               shape, 0, stream);
}

extern "C" int vo_gru_conv_q_bf16x3(const float* x, const float* rh, const void* arranged, const float* bias, const float* cq,
                                    int64_t cq_batch_stride, const float* z, const float* h, int64_t h_batch_stride, float* h_out,
                                    float* h_copy, int64_t copy_batch_stride, int N, int Cin, int H, int W, int C, int shape,
                                    int pieces, int weight_exp, void* stream) {
    VO_PIECES_CHECK("vo_gru_conv_q_bf16x3", pieces, weight_exp);
    return pieces ? F16::gru_q(x, rh, arranged, bias, cq, cq_batch_stride, z, h, h_batch_stride, h_out, h_copy, copy_batch_stride, N, Cin, H, W,
                               C, shape, weight_exp, stream)
                  : Bf::gru_q(x, rh, arranged, bias, cq, cq_batch_stride, z, h, h_batch_stride, h_out, h_copy, copy_batch_stride, N, Cin, H, W,
                              C, shape, 0, stream);
}

// vo_conv1x1_stream.hip: the same layer with the channel block's weights resident in LDS (same weights image,
// same bits); ``use_template`` = 1 keeps the layer on the template (for comparisons of the two kernels)
bool vo_conv1x1_stream_fits(int C, int64_t P, int K);
int vo_conv1x1_stream_launch(const char* name, const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale,
                             const float* shift, int relu, int N, int C, int64_t P, int K, int pieces, int weight_exp, hipStream_t st);

extern "C" int vo_conv1x1_bf16x3(const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale,
                                 const float* shift, int relu, int N, int C, int H, int W, int K, int pieces, int weight_exp,
                                 int use_template, void* stream) {
    VO_PIECES_CHECK("vo_conv1x1_bf16x3", pieces, weight_exp);
    if (!use_template && H > 0 && W > 0 && vo_conv1x1_stream_fits(C, (int64_t)H * W, K) &&
        ((reinterpret_cast<uintptr_t>(arranged) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && (y_batch_stride & 3) == 0)
        return vo_conv1x1_stream_launch("vo_conv1x1_bf16x3", x, arranged, y, y_batch_stride, scale, shift, relu, N, C, (int64_t)H * W, K,
                                        pieces, weight_exp, vo_stream(stream));
    if (pieces) return F16::conv1x1("vo_conv1x1_bf16x3", x, arranged, y, y_batch_stride, scale, shift, relu, N, C, H, W, K, weight_exp, 0, stream);
    return Bf::conv1x1("vo_conv1x1_bf16x3", x, arranged, y, y_batch_stride, scale, shift, relu, N, C, H, W, K, 0, 0, stream);
}

extern "C" int vo_conv1x1_pixel_major_bf16x3(const float* x, const void* arranged, float* y, const float* scale, const float* shift,
                                             int relu, int N, int C, int H, int W, int K, int pieces, int weight_exp, void* stream) {
    VO_PIECES_CHECK("vo_conv1x1_pixel_major_bf16x3", pieces, weight_exp);
    return pieces ? F16::conv1x1("vo_conv1x1_pixel_major_bf16x3", x, arranged, y, 0, scale, shift, relu, N, C, H, W, K, weight_exp, 1, stream)
                  : Bf::conv1x1("vo_conv1x1_pixel_major_bf16x3", x, arranged, y, 0, scale, shift, relu, N, C, H, W, K, 0, 1, stream);
}

// rows cut into bf16 pieces for vo_corr_volume_2d_presplit: bf16 pieces by definition
extern "C" int vo_conv1x1_split_rows_bf16x3(const float* x, const void* arranged, void* y, const float* scale, const float* shift,
                                            int relu, int N, int C, int H, int W, int K, void* stream) {
    return Bf::conv1x1("vo_conv1x1_split_rows_bf16x3", x, arranged, static_cast<float*>(y), 0, scale, shift, relu, N, C, H, W, K, 0, 2, stream);
}
