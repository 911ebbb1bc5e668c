/*
 * vo_hip.h -- C ABI of the MI355X (gfx950) visual-odometry hot-path library.
 *
 * The reference (VOxFF/visual_odometry) is pure Python; its plugin boundary is a
 * set of abstract base classes, and the arithmetic behind them is NumPy, PyTorch
 * and the RAFT / RAFT-Stereo submodules.  Each entry point below replaces the
 * arithmetic behind one of those interface methods (cited per function), so a
 * Python subclass of the reference's own ABC can forward to it through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - tensors are dense, row-major, float32 unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     all work is enqueued asynchronously on it, nothing synchronises;
 *   - return value: VO_OK (0) or a negative VO_ERR_* code; the text of the last
 *     error on the calling thread is available from vo_last_error();
 *   - the library never allocates, frees or retains caller memory.
 */
#ifndef VO_HIP_H
#define VO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_OK                0
#define VO_ERR_INVALID_ARG  -1   /* null pointer, non-positive size, unsupported shape */
#define VO_ERR_LAUNCH       -2   /* the HIP runtime rejected the launch               */
#define VO_ERR_UNSUPPORTED  -3   /* shape outside what the kernels were built for     */

#define VO_MAX_LEVELS 4          /* corr_levels=4: stereo/stereo_disparity_RAFT.py:40 */

/* Library version as major*10000 + minor*100 + patch; it changes whenever an entry point's signature does.
 * VO_HIP_VERSION is the version this header describes: a caller compares vo_version() with it (or with the
 * value its own binding was written for) before the first call, because a library built from another header
 * would take shifted arguments without any error. */
#define VO_HIP_VERSION 301
int vo_version(void);
/* Message for the last non-zero status returned on this thread ("" if none). */
const char* vo_last_error(void);

/* ---------------------------------------------------------------------------
 * Input stage: (frames,H,W) uint8 gray -> (frames,3,H+pt+pb,W+pl+pr) float32, the gray
 * value replicated into the three channels, borders replicate-padded, and with
 * normalize != 0 mapped to 2*(x/255)-1.  Stands for _loadImage + InputPadder.pad +
 * the models' own normalisation: flow/flow_map_RAFT.py:69-72,91-106,
 * stereo/stereo_disparity_RAFT.py:79-82,97-110.  Padded width must be a multiple of 4.
 * ------------------------------------------------------------------------- */
int vo_prepare_images(const uint8_t* gray, float* out, int frames, int H, int W,
                      int pad_left, int pad_right, int pad_top, int pad_bottom,
                      int normalize, void* stream);
/* The same with ONE output plane per frame, out (frames,1,H+pt+pb,W+pl+pr): the three channels the
 * wrappers build from a gray frame are equal (flow/flow_map_RAFT.py:96,101-102), so the first
 * convolution of every encoder may run on one plane with its filter summed over the input channels
 * (sum_c w[k,c]*x = (sum_c w[k,c])*x): a third of the stems' multiplications. */
int vo_prepare_images_gray(const uint8_t* gray, float* out, int frames, int H, int W,
                           int pad_left, int pad_right, int pad_top, int pad_bottom,
                           int normalize, void* stream);

/* ---------------------------------------------------------------------------
 * Stereo rectification (stereo/stereo_rectification.py; the reference calls OpenCV).
 *   vo_rectify_maps       cv2.initUndistortRectifyMap(K, D, R, P, size, CV_32FC1) (:38-39):
 *                         for every rectified pixel the raw-image position, plumb-bob
 *                         distortion (k1 k2 p1 p2 [k3 k4 k5 k6], n_dist of them), float64
 *                         arithmetic, float32 maps (H,W).  K 3x3, R 3x3, P 3x4 row-major and
 *                         the coefficients are HOST arrays.
 *   vo_remap_bilinear_u8  cv2.remap(img, map_x, map_y, INTER_LINEAR) (:67-68, :113-114) for a
 *                         batch of uint8 frames (frames,H,W) -> (frames,Ho,Wo) sharing one
 *                         map pair (Ho,Wo): 5-bit fixed-point fractions, zero border.
 *                         Wo must be a multiple of 4.
 *   vo_erode3x3_u8        cv2.erode(mask, ones(3,3)) (:122-123) on a 0/1 mask; pixels outside
 *                         the image do not erode.
 * ------------------------------------------------------------------------- */
int vo_rectify_maps(const double* K_host, const double* dist_host, int n_dist, const double* R_host,
                    const double* P_host, float* map_x, float* map_y, int H, int W, void* stream);
int vo_remap_bilinear_u8(const uint8_t* src, const float* map_x, const float* map_y, uint8_t* dst,
                         int frames, int H, int W, int Ho, int Wo, void* stream);
int vo_erode3x3_u8(const uint8_t* in, uint8_t* out, int H, int W, void* stream);

/* Storage type of the conv nets' activation tensors in the fused passes below.  The
 * reference runs both networks in float32 (mixed_precision=False, flow/flow_map_RAFT.py:38,
 * stereo/stereo_disparity_RAFT.py:43); VO_F16 serves the published models'
 * mixed_precision switch (stereo/raft_args.json:7).  Arithmetic is float32 either way;
 * biases, scales and shifts are always float32. */
enum { VO_F32 = 0, VO_F16 = 1 };

/* ---------------------------------------------------------------------------
 * Fused, in-place normalisation / activation passes of the conv encoders (the
 * encoders themselves are part of the RAFT / RAFT-Stereo models the reference
 * instantiates at flow/flow_map_RAFT.py:42 and stereo/stereo_disparity_RAFT.py:46).
 * x: NCHW of `dtype`, planes = N*C slices of hw = H*W elements.
 *   vo_instance_norm_relu   InstanceNorm2d (no affine, biased variance, eps) [+ ReLU]
 *   vo_channel_affine_relu  x*scale[c] + shift[c] [+ ReLU]  (BatchNorm2d inference)
 *   vo_add_relu             a = max(a + b, 0)                (residual join)
 * skip (same shape and dtype as x, or NULL) folds the residual join into the same
 * pass: x = max(skip + act(norm(x)), 0).
 * ------------------------------------------------------------------------- */
int vo_instance_norm_relu(void* x, const void* skip, int64_t planes, int64_t hw, float eps, int relu,
                          int dtype, void* stream);
/* The same normalisation with the plane statistics already known (stats (planes,2) float64:
 * sum, sum of squares, as vo_conv3x3_winograd accumulates them): one read-modify-write pass. */
/* With skip_stats (planes,2) the residual branch `skip` is itself a raw convolution output: it is normalised
 * with its own sums and skip_eps (and rectified when skip_relu != 0) while it is read, i.e.
 * x = max(act_s(norm(skip)) + act(norm(x)), 0) -- the stems' and the projection shortcuts' norms cost no pass. */
int vo_instance_norm_apply(void* x, const void* skip, const double* stats, const double* skip_stats, int64_t planes,
                           int64_t hw, float eps, float skip_eps, int relu, int skip_relu, int dtype, void* stream);
int vo_channel_affine_relu(void* x, const float* scale, const float* shift, const void* skip,
                           int64_t batch, int C, int64_t hw, int relu, int dtype, void* stream);
int vo_add_relu(void* a, const void* b, int64_t n, int dtype, void* stream);
/* (batch,channels,h,w) -> (batch,channels,H,W) bilinear, corner-aligned (align_corners=True):
 * the inter-level state exchange of RAFT-Stereo's multi-level GRU.  out may be a channel
 * slice of a wider tensor: items out_batch_stride ELEMENTS apart (0 = dense). */
int vo_resize_bilinear(const void* in, void* out, int64_t out_batch_stride, int batch, int channels,
                       int h, int w, int H, int W, int dtype, void* stream);
/* (planes,h,w) -> (planes,(h-1)/2+1,(w-1)/2+1): 3x3 average pooling, stride 2, padding 1, the padding
 * counted in the divisor (torch's avg_pool2d(x, 3, stride=2, padding=1)): the downward state exchange
 * (pool2x) of the same multi-level GRU (core/update.py of the RAFT-Stereo model built at
 * stereo/stereo_disparity_RAFT.py:46).  Both tensors dense. */
int vo_avg_pool3x3s2(const void* in, void* out, int64_t planes, int h, int w, int dtype, void* stream);
/* The same pooling of (batch,channels,h,w) written into a channel slice of a wider tensor: items of out are
 * out_batch_stride ELEMENTS apart (0 = dense), each a dense (channels,H,W) block.  The multi-level GRU of
 * RAFT-Stereo (core/update.py of the model built at stereo/stereo_disparity_RAFT.py:46) concatenates the pooled
 * state of the finer level behind its own hidden state; this writes it where the concatenation would put it. */
int vo_avg_pool3x3s2_into(const void* in, void* out, int64_t out_batch_stride, int batch, int channels, int h, int w,
                          int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * 3x3, stride 1, padding 1 float32 convolution (NCHW, no bias) as Winograd F(2x2,3x3)
 * on the matrix cores: the dominant layer type of both networks' encoders and update
 * blocks (models built at flow/flow_map_RAFT.py:42, stereo/stereo_disparity_RAFT.py:46).
 *   vo_winograd_filter   weight (K,C,3,3) -> transformed filter, ceil(K/64)*64 * C * 16
 *                        floats (do once per weight).  C must be a multiple of 8.
 *   vo_conv3x3_winograd  x (N,C,H,W), transformed filter -> y (N,K,H,W), with an optional
 *                        epilogue at the store: v = v*scale[k] + shift[k] (shift alone = bias;
 *                        both = folded inference batch norm), ReLU when relu != 0, then the
 *                        residual join v = max(v + skip, 0) with skip (N,K,H,W).  Any of
 *                        scale / shift / skip may be NULL.  y may be a channel slice of a
 *                        wider NCHW tensor: consecutive items y_batch_stride ELEMENTS apart
 *                        (0 = dense, K*H*W), so a consumer's concatenated input is written
 *                        in place instead of being copied together.  stats (N,K,2) float64
 *                        or NULL: the kernel ADDS each (image, channel) plane's sum and sum of
 *                        squares of the raw output (caller zeroes it; no epilogue then) --
 *                        the statistics pass of a following instance norm, for free.
 *                        in_stats (N,C,2) float64 or NULL: the INPUT is normalised while it is
 *                        staged, x^ = relu((x - mean)*rstd) per (image, channel) with the plane
 *                        sums a producing launch left there (eps = in_eps; C <= 512): the
 *                        instance norm + ReLU between two convolutions without its own pass.
 * ------------------------------------------------------------------------- */
int vo_winograd_filter(const float* weight, float* transformed, int K, int C, void* stream);
int vo_conv3x3_winograd(const float* x, const float* transformed, float* y, int64_t y_batch_stride,
                        const float* scale, const float* shift, int relu, const float* skip,
                        double* stats, const double* in_stats, float in_eps,
                        int N, int C, int H, int W, int K, void* stream);
/* The same convolution as Winograd F(4x4,3x3): 36 instead of 64 transformed-domain
 * multiplies per 16 output pixels and channel pair, at a rounding error about twenty times
 * that of the direct sum (1e-5 of the largest output; still float32 arithmetic throughout).
 * Arguments, epilogue and statistics exactly as above; the transformed filter holds
 * ceil(K/64)*64 * C * 36 floats and is NOT interchangeable with vo_winograd_filter's. */
int vo_winograd4_filter(const float* weight, float* transformed, int K, int C, void* stream);
int vo_conv3x3_winograd4(const float* x, const float* transformed, float* y, int64_t y_batch_stride,
                         const float* scale, const float* shift, int relu, const float* skip,
                         double* stats, const double* in_stats, float in_eps,
                         int N, int C, int H, int W, int K, void* stream);

/* Which 16-bit format the split convolutions below cut a float32 operand into (their ``pieces`` argument), and the
 * power of two that goes with float16 pieces (``weight_exp``):
 *   VO_PIECES_BF16  x = xh + xl, both bf16: 16 significant bits per operand, float32's exponent range; largest error
 *                   4-6e-6 of a layer's largest output.  weight_exp must be 0.
 *   VO_PIECES_F16   both float16: 22 significant bits per operand; the error of a layer is that of the float32 matrix
 *                   instruction (the float32 accumulation dominates).  float16's exponent range is met with exact powers
 *                   of two: the *_filter call multiplies the weights by 2^weight_exp before cutting them (the caller
 *                   picks weight_exp so that max|w| * 2^weight_exp lies in [2^13, 2^14) and passes the same value to the
 *                   convolution, which multiplies its accumulators by 2^-weight_exp); an input's low part is kept 2^11
 *                   times too large and meets weights scaled by 2^-11.  Inputs keep 22 bits for 6.1e-5 <= |x| <= 65504
 *                   and an absolute error below 2^-36 under that; |x| > 65504 yields NaN, never a wrong number.
 * The reference runs float32 throughout (mixed_precision=False: flow/flow_map_RAFT.py:38,
 * stereo/stereo_disparity_RAFT.py:43); VO_PIECES_F16 is the arithmetic that stands for it, VO_PIECES_BF16 the faster,
 * narrower one.  Same instruction count, same kernels, same layouts either way. */
enum { VO_PIECES_BF16 = 0, VO_PIECES_F16 = 1 };

/* The same convolution (3x3, stride 1, padding 1, float32 in and out, NCHW) as a direct sum on
 * the bf16 / float16 matrix cores with every float32 product assembled from 16-bit pieces: x = xh + xl,
 * w = wh + wl (high part = the value rounded to bf16, low part = the rounded rest), and
 * x*w ~ xl*wh + xh*wl + xh*wh with float32 accumulation.  Each product keeps 16 significant bits;
 * the largest error on the networks' layers is 4-6e-6 of the largest output (the F(4x4,3x3) kernel
 * above: 1e-5).  Three bf16 matrix instructions per product still run at five times the rate of the
 * float32 matrix instruction, which is why this is the default 3x3 kernel where it applies.
 *   vo_conv3x3_bf16x3_filter  weight (K,C,3,3) float32 -> both bf16 parts in operand order,
 *                             ceil(K/64)*64 * C * 9 * 4 BYTES.  C must be a multiple of 16.
 *   vo_conv3x3_bf16x3         arguments, epilogue, statistics and normalise-on-load exactly as
 *                             vo_conv3x3_winograd.  W must be even, x 8-byte aligned. */
int vo_conv3x3_bf16x3_filter(const float* weight, void* arranged, int K, int C, int pieces, int weight_exp, void* stream);
int vo_conv3x3_bf16x3(const float* x, const void* arranged, float* y, int64_t y_batch_stride,
                      const float* scale, const float* shift, int relu, const float* skip,
                      double* stats, const double* in_stats, float in_eps,
                      int N, int C, int H, int W, int K, int pieces, int weight_exp, void* stream);

/* The same split-bf16 arithmetic for the two other heavy layer types (one kernel template):
 *   vo_conv5_bf16x3      1x5 (padding (0,2)) / 5x1 (padding (2,0)) stride-1 convolution, the gate
 *                        convolutions of RAFT's separable GRU (vertical = 0 / 1); weights from
 *                        vo_conv5_bf16x3_filter ((K,C,5) float32 -> ceil(K/64)*64 * C * 5 * 4 bytes);
 *                        optional epilogue v*scale[k] + shift[k], ReLU; y may be a channel slice.
 *   vo_conv3x3s2_bf16x3  3x3 stride-2 padding-1 convolution, the encoders' downsampling layers: x (N,C,H,W)
 *                        -> y (N,K,(H-1)/2+1,(W-1)/2+1); weights from vo_conv3x3_bf16x3_filter; optional
 *                        epilogue as above, or stats (N,K,2) float64 as vo_conv3x3_winograd's.
 * C a multiple of 16, W even, x 8-byte aligned. */
int vo_conv5_bf16x3_filter(const float* weight, void* arranged, int K, int C, int pieces, int weight_exp, void* stream);
int vo_conv5_bf16x3(const float* x, const void* arranged, float* y, int64_t y_batch_stride,
                    const float* scale, const float* shift, int relu,
                    int N, int C, int H, int W, int K, int vertical, int pieces, int weight_exp, void* stream);
int vo_conv3x3s2_bf16x3(const float* x, const void* arranged, float* y,
                        const float* scale, const float* shift, int relu, double* stats,
                        int N, int C, int H, int W, int K, int pieces, int weight_exp, void* stream);
/* A downsampling residual block's two convolutions of its input in ONE pass (flow/flow_map_RAFT.py:42 and
 * stereo/stereo_disparity_RAFT.py:46 construct the encoders whose stride-2 stages hold them): the 3x3 stride-2
 * convolution (weights from vo_conv3x3_bf16x3_filter) and the 1x1 stride-2 projection shortcut (weights from
 * vo_conv1x1_bf16x3_filter), K output channels each.  The shortcut's input samples x[2y][2x] are the 3x3 filter's centre
 * tap, so the patch staged for the one serves the other: one more tap of matrix instructions per chunk instead of a second
 * kernel that reads and stages x again.  Per output its own epilogue (scale / shift / ReLU, folded batch norm) or its own
 * plane sums (stats, stats2: float64 (N,K,2), zeroed by the caller; no epilogue with them).  Needs an even W, an output
 * width in multiples of 4 and 16-byte aligned outputs (dense (N,K,ceil(H/2),ceil(W/2))).  weight_exp / weight_exp_1x1:
 * the exponents of the two weight images (float16 pieces). */
int vo_conv3x3s2_shortcut_bf16x3(const float* x, const void* arranged, const void* arranged1x1, float* y, float* y2,
                                 const float* scale, const float* shift, int relu, double* stats,
                                 const float* scale2, const float* shift2, int relu2, double* stats2,
                                 int N, int C, int H, int W, int K, int pieces, int weight_exp, int weight_exp_1x1, void* stream);
/* The 1x1 stride-2 convolution of the residual stages' projection shortcuts on the same template (it reads
 * only the rows the stride lands on): weights (K,C) from vo_conv1x1_bf16x3_filter (ceil(K/64)*64 * C * 4 bytes),
 * epilogue or stats as vo_conv3x3s2_bf16x3. */
int vo_conv1x1_bf16x3_filter(const float* weight, void* arranged, int K, int C, int pieces, int weight_exp, void* stream);
/* ... and the stride-1 1x1 convolution with many input and output channels (the motion encoders' first layer on the
 * correlation window, the mask heads), where the float32 matrix instruction, not memory, bounds vo_conv1x1:
 * same weights, epilogue (scale / shift / ReLU) and channel-slice output as vo_conv3x3_bf16x3.  Layers with at least
 * 64 output channels in multiples of 8, planes of whole groups of four pixels and 16-byte aligned tensors run on a kernel that keeps
 * a block's weights in LDS (same bits in the result); use_template != 0 keeps them on the template's kernel
 * (a switch for comparing the two). */
int vo_conv1x1_bf16x3(const float* x, const void* arranged, float* y, int64_t y_batch_stride,
                      const float* scale, const float* shift, int relu,
                      int N, int C, int H, int W, int K, int pieces, int weight_exp, int use_template, void* stream);
/* vo_conv1x1_bf16x3 with y written pixel-major, (N, H*W, K) (K a multiple of 4): the feature encoders' last layer in the
 * layout the correlation kernels read (cf. vo_conv1x1_pixel_major). */
int vo_conv1x1_pixel_major_bf16x3(const float* x, const void* arranged, float* y,
                                  const float* scale, const float* shift, int relu,
                                  int N, int C, int H, int W, int K, int pieces, int weight_exp, void* stream);
/* vo_conv1x1_pixel_major_bf16x3 whose rows leave as bf16 pieces in blocks of 32 channels (K a multiple of 32): the
 * input form of vo_corr_volume_2d_presplit, see vo_split_feature_rows.  y holds N*H*W*K*4 bytes.  bf16 pieces only
 * (weights from vo_conv1x1_bf16x3_filter with VO_PIECES_BF16). */
int vo_conv1x1_split_rows_bf16x3(const float* x, const void* arranged, void* y,
                                 const float* scale, const float* shift, int relu,
                                 int N, int C, int H, int W, int K, void* stream);
int vo_conv1x1s2_bf16x3(const float* x, const void* arranged, float* y,
                        const float* scale, const float* shift, int relu, double* stats,
                        int N, int C, int H, int W, int K, int pieces, int weight_exp, void* stream);
/* One step of a convolutional GRU (RAFT's SepConvGRU halves, RAFT-Stereo's ConvGRU; the update operators of the
 * models built at flow/flow_map_RAFT.py:42 and stereo/stereo_disparity_RAFT.py:46) as TWO convolutions on the
 * split-bf16 template with the gate arithmetic in their stores -- no pointwise pass, no concatenation:
 *   vo_gru_conv_zr_bf16x3  x = [h | inputs] (N,Cin,H,W); weights = cat(W_z, W_r) (2C,Cin,taps) arranged by the filter
 *                          entry point of the shape; bias (2C) = cat(b_z, b_r) or NULL; cz, cr (N,C,H,W) context terms
 *                          with their batch strides, or NULL; h (N,C,H,W) with its batch stride.
 *                          z_out (N,C,H,W) = sigmoid(conv_z + b_z + cz);  rh_out (N,C,H,W) = sigmoid(conv_r + b_r + cr) * h.
 *   vo_gru_conv_q_bf16x3   the q convolution over [rh | x's channels C..] (x is the SAME buffer as above, its first C
 *                          channels are not read); h_out (N,C,H,W) = (1 - z)*h + z*tanh(conv_q + b_q + cq); h_copy
 *                          (or NULL; batch stride copy_batch_stride) receives the same values -- typically x's first C
 *                          channels, ready for the next step.
 * shape: 0 = 3x3 padding 1, 1 = 1x5 padding (0,2), 2 = 5x1 padding (2,0).  C a multiple of 64, Cin of 16, W of 4;
 * every tensor 16-byte aligned, batch strides multiples of 4.  Replaces vo_conv* + vo_gru_gates / vo_gru_update. */
int vo_gru_conv_zr_bf16x3(const float* x, const void* arranged, const float* bias, const float* cz, int64_t cz_batch_stride,
                          const float* cr, int64_t cr_batch_stride, const float* h, int64_t h_batch_stride, float* z_out,
                          float* rh_out, int N, int Cin, int H, int W, int C, int shape, int pieces, int weight_exp, void* stream);
int vo_gru_conv_q_bf16x3(const float* x, const float* rh, const void* arranged, const float* bias, const float* cq,
                         int64_t cq_batch_stride, const float* z, const float* h, int64_t h_batch_stride, float* h_out,
                         float* h_copy, int64_t copy_batch_stride, int N, int Cin, int H, int W, int C, int shape,
                         int pieces, int weight_exp, void* stream);

/* 7x7 padding-3 convolution (stride 1 or 2) of a ONE- or TWO-channel input in the same arithmetic: the encoders'
 * stems on a gray frame (the wrappers' three equal channels folded into one, flow/flow_map_RAFT.py:96,101-102)
 * and the motion encoders' flow branch.  Bound by the store of its output: a workgroup walks four output rows
 * end to end, so stats (N,K,2) float64 (plane sums of the raw output, for the instance norm that follows the
 * stem) cost one pair of atomics per channel and four rows.
 *   vo_conv7x7_bf16x3_filter  weight (K,C,7,7) -> ceil(K/64)*64 * C * 64 * 4 bytes in operand order.
 *   vo_conv7x7_bf16x3         x (N,C,H,W) -> y (N,K,Ho,Wo), Ho = (H-1)/stride + 1; epilogue v*scale[k] + shift[k]
 *                             (either may be NULL), ReLU -- or stats, then no epilogue. */
int vo_conv7x7_bf16x3_filter(const float* weight, void* arranged, int K, int C, int pieces, int weight_exp, void* stream);
int vo_conv7x7_bf16x3(const float* x, const void* arranged, float* y, const float* scale, const float* shift, int relu,
                      double* stats, int N, int C, int H, int W, int K, int stride, int pieces, int weight_exp, void* stream);

/* 1x5 (padding (0,2)) and 5x1 (padding (2,0)) stride-1 float32 convolutions, NCHW, no bias, as
 * one-dimensional Winograd F(4,5) on the matrix cores: the gate convolutions of the separable
 * GRU in RAFT's update block (model built at flow/flow_map_RAFT.py:42).
 *   vo_winograd1d_filter  weight (K,C,5) (the (K,C,1,5) or (K,C,5,1) tensor, contiguous) ->
 *                         transformed filter, ceil(K/64)*64 * C * 8 floats.  C multiple of 8.
 *   vo_conv5_winograd     x (N,C,H,W), transformed filter -> y (N,K,H,W); vertical = 0: the
 *                         filter runs along W (1x5), 1: along H (5x1).  y must not alias x. */
int vo_winograd1d_filter(const float* weight, float* transformed, int K, int C, void* stream);
int vo_conv5_winograd(const float* x, const float* transformed, float* y, int N, int C, int H, int W, int K,
                      int vertical, void* stream);

/* 1x1 stride-1 float32 convolution, NCHW, as a streaming matrix-core GEMM (no LDS): the motion
 * encoders' first layers, projection shortcuts and mask heads of the models built at
 * flow/flow_map_RAFT.py:42 and stereo/stereo_disparity_RAFT.py:46, and the per-tap planes of the
 * two-channel flow heads.
 *   vo_conv1x1_filter  weight (K,C) -> weights in matrix-operand order, ceil(K/64)*64 * ceil(C/8)*8 floats.
 *   vo_conv1x1         x (N,C,P), P = H*W pixels -> y (N,K,P): v = sum_c W[k,c] x[c,p]; then
 *                      v*scale[k] + shift[k] (either may be NULL) and ReLU when relu != 0.  y may be a
 *                      channel slice of a wider tensor (items y_batch_stride ELEMENTS apart, 0 = dense).
 *   vo_conv1x1_pixel_major  the same convolution with the output written as y (N,P,K): each pixel's K
 *                      values contiguous, the layout vo_corr_volume_2d / _1d read.  It is the last layer
 *                      of the feature encoders (core/extractor.py's conv2 in both published models) and
 *                      replaces their NCHW output plus the permute(0,2,3,1).contiguous() copy in front
 *                      of the correlation (core/corr.py). */
int vo_conv1x1_filter(const float* weight, float* arranged, int K, int C, void* stream);
int vo_conv1x1(const float* x, const float* arranged, float* y, int64_t y_batch_stride, const float* scale,
               const float* shift, int relu, int N, int C, int64_t P, int K, void* stream);
int vo_conv1x1_pixel_major(const float* x, const float* arranged, float* y, const float* scale, const float* shift,
                           int relu, int N, int C, int64_t P, int K, void* stream);

/* 7x7 padding-3 float32 convolution (stride 1 or 2) with at most 8 input channels, NCHW, as a
 * streaming matrix-core GEMM (no LDS): the stems of the encoders (3 -> 64 at full resolution,
 * stride 1 in RAFT-Stereo, 2 in RAFT) and the motion encoders' flow branch (2 -> 64 / 128) of the models built at
 * stereo/stereo_disparity_RAFT.py:46 and flow/flow_map_RAFT.py:42.
 *   vo_conv7x7_filter  weight (K,C,7,7) -> weights in matrix-operand order, ceil(K/64)*64 * C * 56 floats.
 *   vo_conv7x7         x (N,C,H,W) -> y (N,K,Ho,Wo), Ho = (H-1)/stride + 1, then v*scale[k] + shift[k]
 *                      (either may be NULL) and ReLU when relu != 0. */
int vo_conv7x7_filter(const float* weight, float* arranged, int K, int C, void* stream);
int vo_conv7x7(const float* x, const float* arranged, float* y, const float* scale, const float* shift, int relu,
               int N, int C, int H, int W, int K, int stride, void* stream);

/* 3x3 stride-2 padding-1 float32 convolution, NCHW, as a streaming matrix-core GEMM (no LDS): the
 * downsampling layers of the encoders' residual stages (models built at flow/flow_map_RAFT.py:42,
 * stereo/stereo_disparity_RAFT.py:46).  C must be a multiple of 8.
 *   vo_conv3x3s2_filter  weight (K,C,3,3) -> weights in matrix-operand order, ceil(K/64)*64 * C * 9 floats.
 *   vo_conv3x3s2         x (N,C,H,W) -> y (N,K,Ho,Wo), Ho = (H-1)/2 + 1, then v*scale[k] + shift[k]
 *                        (either may be NULL) and ReLU when relu != 0. */
int vo_conv3x3s2_filter(const float* weight, float* arranged, int K, int C, void* stream);
int vo_conv3x3s2(const float* x, const float* arranged, float* y, const float* scale, const float* shift, int relu,
                 int N, int C, int H, int W, int K, void* stream);
/* The 1x1 stride-2 convolution of the same stages' projection shortcuts, on the same kernel with a
 * single tap: weights arranged by vo_conv1x1_filter (C a multiple of 8), x (N,C,H,W) -> y (N,K,Ho,Wo). */
int vo_conv1x1s2(const float* x, const float* arranged, float* y, const float* scale, const float* shift, int relu,
                 int N, int C, int H, int W, int K, void* stream);

/* 3x3 / padding 1 convolutions with very few output channels (the flow heads' last
 * layer, 256 -> 2): the framework runs a 1x1 convolution with 9K output channels, one
 * plane per filter tap, planes (N, K*9, H, W) in (k, r, s) order; this pass adds the nine
 * shifted planes and the bias: out[n,k,y,x] = bias[k] + sum_rs planes[n,(k*3+r)*3+s,y+r-1,x+s-1]. */
int vo_shift_add_3x3(const float* planes, const float* bias, float* out, int N, int K, int H, int W,
                     void* stream);

/* ---------------------------------------------------------------------------
 * Gate arithmetic of the convolutional GRUs (SepConvGRU of RAFT, ConvGRU of
 * RAFT-Stereo; same models, same call sites as above).  The z|r and q convolutions
 * run without bias in the framework; these passes add bias + context term, apply the
 * non-linearities and blend.  Contiguous operands are (B,C,hw) / (B,2C,hw) of `dtype`;
 * a "batch stride" operand is a channel slice of a larger NCHW tensor (plane-
 * contiguous, consecutive batch items that many ELEMENTS apart).  cz/cr/cq, the
 * biases (float32) and h_copy may be NULL.
 *   gates : z_out = sigmoid(zr[:, :C] + bz + cz);  rh_out = sigmoid(zr[:, C:] + br + cr) * h
 *   update: h_out = (1 - z)*h + z*tanh(q_conv + bq + cq)   (h_copy receives the same values)
 * ------------------------------------------------------------------------- */
int vo_gru_gates(const void* zr_conv, const float* bias_z, const float* bias_r,
                 const void* cz, int64_t cz_batch_stride, const void* cr, int64_t cr_batch_stride,
                 const void* h, int64_t h_batch_stride, void* z_out,
                 void* rh_out, int64_t rh_batch_stride, int batch, int C, int64_t hw, int dtype, void* stream);
int vo_gru_update(const void* q_conv, const float* bias_q, const void* cq, int64_t cq_batch_stride,
                  const void* z, const void* h, int64_t h_batch_stride,
                  void* h_out, int64_t out_batch_stride, void* h_copy, int64_t copy_batch_stride,
                  int batch, int C, int64_t hw, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * All-pairs correlation volume (RAFT CorrBlock / RAFT-Stereo CorrBlock1D "reg").
 * Call sites in the reference: flow/flow_map_RAFT.py:15,42,81 and
 * stereo/stereo_disparity_RAFT.py:15,39,46,91 (the model code itself lives in
 * un-vendored submodules).
 *
 * Feature maps are PIXEL-MAJOR: fmap[b][y][x][d], d contiguous (channels-last).
 * f32 MFMA (v_mfma_f32_32x32x2_f32), i.e. exact float32 multiply-accumulate.
 *
 * 2-D: corr[b][y1*W+x1][y2*W+x2] = <f1[b,y1,x1,:], f2[b,y2,x2,:]> / sqrt(D)
 * 1-D: corr[b][y][x1][x2]        = <f1[b,y,x1,:],  f2[b,y,x2,:]>  / sqrt(D)
 * D must be a multiple of 32.
 * ------------------------------------------------------------------------- */
int vo_corr_volume_2d(const float* fmap1, const float* fmap2, float* corr,
                      int batch, int H, int W, int D, void* stream);
int vo_corr_volume_1d(const float* fmap1, const float* fmap2, float* corr,
                      int batch, int H, int W1, int W2, int D, void* stream);
/* The same two volumes with every float32 product assembled from three 16-bit matrix-core terms
 * (al*bh + ah*bl + ah*bh, float32 accumulation); the 16-bit instructions run at 16 times the float32 one's rate, so
 * the store of the volume becomes the bound.
 *   pieces = VO_PIECES_BF16: 16 significant bits per operand, largest error about 5e-6 of the largest entry, exact
 *     on small integers; amax1 / amax2 are ignored (NULL).
 *   pieces = VO_PIECES_F16: 22 significant bits per operand (the float32 instruction's error).  A feature map has no
 *     known scale, so each IMAGE's map is multiplied by the power of two that puts its largest |x| into [2^13, 2^14)
 *     before it is cut and the volume is multiplied by the inverse powers in the store (all exact).  amax1 / amax2
 *     (device memory, `batch` entries each, float bits of the largest |x| of fmap1[b] / fmap2[b]) come from
 *     vo_feature_rows_amax; the values never pass through the host.  Per image and not per launch, so a pair's volume
 *     does not depend on the other pairs of its batch.  Multiplying a map by a power of two multiplies the volume by
 *     it, bit for bit.
 * These kernels (and the _presplit ones below) scale by the float32 value 1/sqrt(D) where vo_corr_volume_* divide by
 * sqrt(D): the same bits when D is a power of 4 (D = 256, the networks' width), one more rounding of 2^-24
 * otherwise. */
int vo_corr_volume_2d_bf16x3(const float* fmap1, const float* fmap2, float* corr,
                             int batch, int H, int W, int D, int pieces, const unsigned* amax1, const unsigned* amax2, void* stream);
int vo_corr_volume_1d_bf16x3(const float* fmap1, const float* fmap2, float* corr,
                             int batch, int H, int W1, int W2, int D, int pieces, const unsigned* amax1, const unsigned* amax2,
                             void* stream);
/* amax[i] = float bits of the largest |x| among the elems_per_image values of image i (a multiple of 4; x 16-byte
 * aligned; at most 65535 images); a NaN anywhere in an image makes its entry a NaN's bits.  Part of the same call
 * sites' arithmetic, no counterpart of its own in the reference. */
int vo_feature_rows_amax(const float* x, unsigned* amax, int images, int64_t elems_per_image, void* stream);
/* ... and on feature maps whose rows were split into their pieces beforehand, once per map instead of once per
 * tile that reads the row (the call sites are the same: flow/flow_map_RAFT.py:81, stereo/stereo_disparity_RAFT.py:91).
 * vo_split_feature_rows turns `rows` rows of D float32 into the same 4*D bytes per row: per block of `block`
 * features (32 for the 2-D volume, 16 for the 1-D one; D a multiple of it), `block` high parts followed by
 * `block` low parts; out must not alias x.  VO_PIECES_BF16: high = bf16(x), low = bf16(x - high), amax and
 * rows_per_image ignored; vo_conv1x1_split_rows_bf16x3 (above) writes this block-32 form from a feature encoder's
 * last layer directly.  VO_PIECES_F16: x' = x * 2^e(image), high = f16(x'), low = f16(x' - high), with amax from
 * vo_feature_rows_amax over images of rows_per_image rows (rows a multiple of it).  Results equal
 * vo_corr_volume_*_bf16x3 with the same pieces bit for bit.
 * The 2-D kernel: 256 x 256 tiles, one workgroup per CU walking a tile list, the next tile's first slices requested
 * before the finished tile is stored. */
int vo_split_feature_rows(const float* x, void* out, int64_t rows, int D, int block, int pieces, const unsigned* amax,
                          int64_t rows_per_image, void* stream);
int vo_corr_volume_2d_presplit(const void* fmap1, const void* fmap2, float* corr,
                               int batch, int H, int W, int D, int pieces, const unsigned* amax1, const unsigned* amax2, void* stream);
int vo_corr_volume_1d_presplit(const void* fmap1, const void* fmap2, float* corr,
                               int batch, int H, int W1, int W2, int D, int pieces, const unsigned* amax1, const unsigned* amax2,
                               void* stream);

/* ---------------------------------------------------------------------------
 * Correlation pyramid: levels 1..3 from level 0 in one pass (avg_pool 2x2 / 1x2,
 * stride 2, floor sizes).  `images` = batch*H*W source pixels, each owning an
 * (H2 x W2) slice (2-D) or a W2 row (1-D).  Unused upper levels may be NULL.
 * ------------------------------------------------------------------------- */
int vo_corr_pyramid_2d(const float* l0, float* l1, float* l2, float* l3,
                       int64_t images, int H2, int W2, void* stream);
int vo_corr_pyramid_1d(const float* l0, float* l1, float* l2, float* l3,
                       int64_t rows, int W2, void* stream);

/* ---------------------------------------------------------------------------
 * Windowed correlation lookup (CorrBlock.__call__ / CorrBlock1D.__call__).
 * coords: (B,2,H,W) (x,y) in level-0 pixels of the target frame.
 * 2-D out: (B, levels*(2r+1)^2, H, W); channel = l*(2r+1)^2 + a*(2r+1) + b samples
 *          level l at (x/2^l + a - r, y/2^l + b - r), bilinear, zeros outside,
 *          pixel-centre coordinates (grid_sample align_corners=True).
 * 1-D out: (B, levels*(2r+1), H, W); channel = l*(2r+1) + a samples the row at
 *          x/2^l + a - r (only coords channel 0 is read).
 * pyr[l] is level l as produced above; level sizes are floor-halved from (H2,W2).
 * out_batch_stride: elements between the batch items of `out` (0 = dense); larger when `out` is the leading channels of
 * a wider tensor, e.g. the window padded to a multiple of 16 channels for the convolution that consumes it.
 * Limits (32-bit buffer addressing): 256 * H2 * W2 * 4 bytes (the slices of one workgroup's pixels) and one pair's output
 * below 2 GiB, i.e. target maps up to about two million pixels; larger sizes are refused with VO_ERR_INVALID_ARG.
 * ------------------------------------------------------------------------- */
int vo_corr_lookup_2d(const float* l0, const float* l1, const float* l2, const float* l3,
                      const float* coords, float* out, int64_t out_batch_stride, int batch, int H, int W,
                      int H2, int W2, int levels, int radius, void* stream);
int vo_corr_lookup_1d(const float* l0, const float* l1, const float* l2, const float* l3,
                      const float* coords, float* out, int64_t out_batch_stride, int batch, int H, int W,
                      int W2, int levels, int radius, void* stream);

/* ---------------------------------------------------------------------------
 * Convex upsampling (RAFT.upsample_flow / RAFTStereo.upsample_flow).
 * flow: (B,Cf,H,W); mask: (B, 9*f*f, H, W) raw logits times mask_scale;
 * out:  (B,Cout,f*H,f*W) with Cout <= Cf leading channels (stereo keeps 1).
 * out[b,c,f*y+i,f*x+j] = sum_k softmax_k(mask[b,k*f*f+i*f+j,y,x]) * f*flow[b,c,y+k/3-1,x+k%3-1]
 * ------------------------------------------------------------------------- */
int vo_convex_upsample(const float* flow, const float* mask, float* out,
                       int batch, int Cf, int Cout, int H, int W, int factor,
                       float mask_scale, void* stream);

/* ---------------------------------------------------------------------------
 * StereoDepth.compute_depth -- stereo/stereo_depth.py:25-49.
 * depth = fxB/|d| where |d| > 0 else 0.  divide_in_f64 != 0 reproduces NumPy's
 * float64 quotient rounded once to float32 (np.float64 fx, baseline); 0 divides in
 * float32 with fxB rounded to float32 first (plain Python floats).
 * ------------------------------------------------------------------------- */
int vo_disparity_to_depth(const float* disparity, float* depth, int64_t n,
                          double fxB, int divide_in_f64, void* stream);

/* ---------------------------------------------------------------------------
 * Keypoint stages.  kp: (N,2) float64 (x,y), shared by all B items when
 * kp_batch_stride == 0, else item b starts at kp + b*kp_batch_stride doubles.
 * depth maps (B,H,W), flow (B,2,H,W); Kinv_host = 9 doubles, row-major, on the host.
 * Masks are uint8 0/1 (NumPy bool layout).
 *
 * vo_keypoints_to_3d      Keypoints3DXform.to_3d        keypoints/keypoints_3d.py:20-49
 * vo_keypoints_flow_2d    Keypoints3DFlow.compute_2d_flow keypoints/keypoints_3d_flow.py:23-47
 * vo_keypoints_flow_3d    Keypoints3DFlow.compute_3d_flow keypoints/keypoints_3d_flow.py:49-88
 *                         (p1_out, the frame-1 lift the reference computes and drops, may be NULL)
 * vo_pair_correspondences test_camera_tracking.py:176-200 for B pairs at once:
 *                         P1 = lift(kp, depth1); P2 = 3-D flow; weight = in depth
 *                         range & flow-valid & |Z2-Z1| <= max_dz; n_in_range[b] = points
 *                         left after the range filter (the "< 4 -> skip" test, :188).
 * ------------------------------------------------------------------------- */
int vo_keypoints_to_3d(const double* kp, int64_t kp_batch_stride, const float* depth,
                       const double* Kinv_host, double* points_out,
                       int batch, int N, int H, int W, void* stream);
int vo_keypoints_flow_2d(const double* kp, int64_t kp_batch_stride, const float* flow,
                         double* next_out, uint8_t* valid_out,
                         int batch, int N, int H, int W, void* stream);
int vo_keypoints_flow_3d(const double* kp, int64_t kp_batch_stride,
                         const float* depth1, const float* depth2, const float* flow,
                         const double* Kinv_host, double* p2_out, uint8_t* valid_out,
                         double* p1_out, int batch, int N, int H, int W, void* stream);
int vo_pair_correspondences(const double* kp, int64_t kp_batch_stride,
                            const float* depth1, const float* depth2, const float* flow,
                            const double* Kinv_host, double min_depth, double max_depth,
                            double max_dz, double* p1_out, double* p2_out,
                            uint8_t* weight_out, int32_t* n_in_range_out,
                            int batch, int N, int H, int W, void* stream);

/* Keypoints3DXform.to_2d -- keypoints/keypoints_3d.py:51-70.  points (n,3) float64,
 * K_host = 9 doubles on the host; uv_out (n,2) float64, keep_out (n) uint8: 0 where
 * Z <= 0 (the reference DROPS those rows; the caller compacts with keep_out). */
int vo_keypoints_to_2d(const double* points, const double* K_host, double* uv_out, uint8_t* keep_out,
                       int64_t n, void* stream);

/* Multi-frame keypoint tracks -- the loop of test_keypoints_flow.py:235-310 over
 * `frames` consecutive frames whose first has sequence index first_index.
 * depth (frames,H,W), flow (frames-1,2,H,W) with flow[i]: frame i -> i+1;
 * grid_kp (N,2) float64 = the seeding grid.  Iteration i seeds (grid points with
 * depth[i] > 0) when (first_index+i) % k == 0, at i == 0, or when fewer than
 * min_points rows survived iteration i-1 (:280); otherwise it advances the surviving
 * rows with compute_3d_flow(depth[i], depth[i+1], flow[i]) and re-projects them with
 * to_2d.  Outputs, slot j = grid point j of the latest seeding: tracks_out
 * (frames-1,N,3) float64 (x, y, Z), zeros for dead slots; alive_out (frames-1,N)
 * uint8; reseeded_out (frames-1) uint8.  One workgroup per k-aligned segment. */
int vo_track_keypoints(const double* grid_kp, const float* depth, const float* flow,
                       const double* K_host, const double* Kinv_host, int first_index, int k,
                       int min_points, double* tracks_out, uint8_t* alive_out, uint8_t* reseeded_out,
                       int frames, int N, int H, int W, void* stream);

/* ---------------------------------------------------------------------------
 * CameraSvdXform.compute_camera_xform -- camera/camera_svd_xform.py:18-48,
 * batched: one 64-lane wavefront per pair.
 * P, Q: (B,N,3) float64; weight: (B,N) uint8 or NULL (all ones);
 * offset_host: 3 doubles on the host or NULL.
 * T_out: (B,4,4) float64 row-major [R t; 0 0 0 1] with Q ~ R P + t (+ R offset);
 * n_used_out: (B) int32 number of weighted points, or NULL.
 * ------------------------------------------------------------------------- */
int vo_kabsch_batch(const double* P, const double* Q, const uint8_t* weight,
                    const double* offset_host, double* T_out, int32_t* n_used_out,
                    int batch, int N, void* stream);

/* ---------------------------------------------------------------------------
 * The geometry half of the tracking loop for a batch of frame pairs in ONE call --
 * test_camera_tracking.py:176-210: vo_pair_correspondences -> vo_kabsch_batch ->
 * skip flag, without the correspondences ever leaving the device.
 * Inputs as vo_pair_correspondences plus offset_host (CameraSvdXform's offset, 3 doubles
 * on the host or NULL) and min_points (:188, 4 in the driver).
 * T_out: (B,4,4) float64; skipped_out: (B) uint8, 1 where fewer than min_points keypoints
 * pass the depth-range filter (the reference's `continue`, :188-190) or no point survives
 * all filters (T_out is then NaN, as CameraSvdXform returns for an empty set).
 * workspace: DEVICE memory, 256-byte aligned, at least vo_pair_poses_workspace_bytes(B, N)
 * bytes (P1, P2, weights and two counters per pair live there); contents on entry do not
 * matter, contents on return are unspecified.  Calls sharing a workspace must be ordered
 * on one stream.
 * ------------------------------------------------------------------------- */
int64_t vo_pair_poses_workspace_bytes(int batch, int N);
int vo_pair_poses(const double* kp, int64_t kp_batch_stride,
                  const float* depth1, const float* depth2, const float* flow,
                  const double* Kinv_host, double min_depth, double max_depth, double max_dz,
                  const double* offset_host, int min_points,
                  double* T_out, uint8_t* skipped_out,
                  void* workspace, int64_t workspace_bytes,
                  int batch, int N, int H, int W, void* stream);

/* vo_pair_poses without the workspace argument: the correspondences live in a fixed array
 * inside the library, so the call allocates nothing and needs nothing allocated for it.
 * It has NO counterpart in the reference; it exists for callers that cannot allocate
 * device memory around the call -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  Arguments and results exactly as vo_pair_poses.
 * Limits: batch <= VO_PAIR_POSES_STATIC_MAX_BATCH and batch*N <=
 * VO_PAIR_POSES_STATIC_MAX_POINTS (VO_ERR_UNSUPPORTED beyond; use vo_pair_poses then).
 * The array (about 12.4 MiB) exists once per GPU the library runs on, for the life of the
 * process, and is shared by every call on that GPU: calls that overlap in time on one GPU
 * (different streams or threads) race on it -- order them on one stream.  The FIRST call on
 * each GPU looks the array's device address up (hipGetSymbolAddress, under a mutex): make that
 * call outside any stream capture; every later call on that GPU only enqueues kernels and a
 * memset on `stream` and may be captured into a HIP graph. */
#define VO_PAIR_POSES_STATIC_MAX_BATCH  8192
#define VO_PAIR_POSES_STATIC_MAX_POINTS 262144
int vo_pair_poses_static(const double* kp, int64_t kp_batch_stride,
                         const float* depth1, const float* depth2, const float* flow,
                         const double* Kinv_host, double min_depth, double max_depth, double max_dz,
                         const double* offset_host, int min_points,
                         double* T_out, uint8_t* skipped_out,
                         int batch, int N, int H, int W, void* stream);

/* ---------------------------------------------------------------------------
 * CameraRansacXform.compute_camera_xform -- camera/camera_svd_xform.py:52-149,
 * batched: one 256-lane workgroup per pair, one 3-point hypothesis per lane per round.
 * P, Q: (B,N,3) float64, 3 <= N <= 65535; weight: (B,N) uint8 or NULL.
 * A hypothesis' inliers are the weighted points with |R p + t - q| < threshold; the
 * FIRST hypothesis with the strictly largest count wins and the pose is refitted on
 * its inliers (identity when fewer than 3).  The reference samples from NumPy's
 * unseeded global generator; here triple h of the batch's item i is a pure function of
 * (seed, b, h) with b = first_pair + i, the pair's index in the whole sequence (so that the
 * result does not depend on how a sequence is cut into batches or shards; b is taken
 * modulo 2^32): s = splitmix64(seed ^ (b << 32 | h)); i0 = s % n;
 * s = splitmix64(s); i1 = s % (n-1), skipping i0; s = splitmix64(s); i2 = s % (n-2),
 * skipping both (n = number of weighted points, indices into their compacted order).
 * T_out: (B,4,4) float64; n_inliers_out: (B) int32 inliers of the winner (-1 when a
 * pair has fewer than 3 weighted points); inlier_mask_out: (B,N) uint8 or NULL.
 * ------------------------------------------------------------------------- */
int vo_ransac_kabsch_batch(const double* P, const double* Q, const uint8_t* weight,
                           double threshold, int iterations, uint64_t seed,
                           double* T_out, int32_t* n_inliers_out, uint8_t* inlier_mask_out,
                           int batch, int N, int64_t first_pair, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VO_HIP_H */
