// 7x7 padding-3 convolution (stride 1 or 2) with very few input channels (the networks' stems read the
// 3-channel image, the motion encoders the 2-channel flow) in float32 as a streaming matrix-core
// GEMM without LDS:   y[n,k,py,px] = act(scale[k] * sum_{c,r,s} w[k,c,r,s] x[n,c,py+r-3,px+s-3] + shift[k]).
//
// The reduction index is ordered (c, r, s) with s padded from 7 to 8, so one 16-byte load of the
// pre-arranged weights is the A operand of the four k-steps of one filter row (k-step ks holds the
// taps s = 2ks and 2ks+1).  A wave owns 64 output channels x 64 consecutive pixels of one image row
// (pixel tile t = pixels 2n+t, 64 accumulator registers); its B operand for tap s is the input row
// shifted by s-3, read straight from memory (the 49-fold reuse lives in L1/L2).  Columns outside the
// image get a lane offset past the buffer's extent (the range check returns 0.0f); rows outside
// the image are skipped altogether.  The four waves of a workgroup take four consecutive rows.
#include "vo_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KB = 64;
constexpr int OOB = 0x40000000;            // lane offset no image reaches (images are below 1 GiB)

// Wg[kb][cr][kt][lane][ks]:  k = kb*64 + kt*32 + (lane & 31), cr = c*7 + r, tap s = 2*ks + (lane >> 5) (s = 7: zero)
__global__ __launch_bounds__(256) void filter7_kernel(const float* __restrict__ w, float* __restrict__ Wg, int K, int C, int Kpad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Kpad * C * 7 * 8) return;
    const int s = (int)(idx & 7), cr = (int)((idx >> 3) % (C * 7)), k = (int)((idx >> 3) / (C * 7));
    const float v = (k < K && s < 7) ? w[((int64_t)k * C * 7 + cr) * 7 + s] : 0.0f;
    const int kb = k >> 6, kt = (k >> 5) & 1, kl = k & 31, ks = s >> 1, kk = s & 1;
    Wg[(((((int64_t)kb * (C * 7) + cr) * 2 + kt) * 64) + kk * 32 + kl) * 4 + ks] = v;
}

struct Epi { const float* scale; const float* shift; int relu; };

// S: stride.  (H, W): input size, (Ho, Wo): output size.
template <int S>
__global__ __launch_bounds__(256) void conv7_kernel(const float* __restrict__ x, const float* __restrict__ Wg,
                                                    float* __restrict__ y, Epi ep, int C, int H, int W, int Ho, int Wo, int K,
                                                    int blocks_x, int blocks_y, int kblocks) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bid = blockIdx.x;
    const int kb = bid % kblocks; bid /= kblocks;
    const int bx = bid % blocks_x; bid /= blocks_x;
    const int by = bid % blocks_y;
    const int n = bid / blocks_y;
    const int py = by * 4 + wave;                                // this wave's output row
    if (py >= Ho) return;                                        // waves are independent: no barrier anywhere
    const int kk = lane >> 5, nn = lane & 31;
    const int px0 = bx * 64 + 2 * nn;                            // this lane's first output pixel of the row
    const int ncr = C * 7;

    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (int64_t)n * C * H * W), 0, C * H * W * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wg + (int64_t)kb * ncr * 512), 0, ncr * 512 * 4, 0x00020000);
    // byte offset inside an input row of the element (k-step ks, pixel tile t) needs, or OOB
    int coff[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = S * (px0 + t) + 2 * ks + kk - 3;
            coff[ks][t] = (col >= 0 && col < W) ? col * 4 : OOB;
        }
    const int w_off = lane * 16;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][t][e] = 0.0f;

    float bcur[4][2], bnxt[4][2];
    u32x4 acur[2], anxt[2];
    // the (channel, filter row) pairs whose input row lies inside the image, in order
    const int iy = S * py - 3;                                   // input row of filter row 0
    const int r_lo = iy < 0 ? -iy : 0, r_hi = iy + 6 >= H ? H - 1 - iy : 6;         // valid filter rows r_lo..r_hi
    const int nr = r_hi - r_lo + 1, total = C * nr;
    auto load = [&](int i) {
        i = i < total ? i : total - 1;                           // past the end: reload the last one, never used
        const int c = i / nr, r = r_lo + i - c * nr;
        const int cr = c * 7 + r;
        const int row = ((c * H + iy + r) * W) * 4;              // wave-uniform
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) anxt[kt] = __builtin_amdgcn_raw_buffer_load_b128(wrs, w_off + kt * 1024, cr * 2048, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                bnxt[ks][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, coff[ks][t] + row, 0, 0));
    };
    load(0);
    for (int i = 0; i < total; ++i) {
        acur[0] = anxt[0]; acur[1] = anxt[1];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { bcur[ks][0] = bnxt[ks][0]; bcur[ks][1] = bnxt[ks][1]; }
        load(i + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    acc[kt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(acur[kt][ks]), bcur[ks][t], acc[kt][t], 0, 0, 0);
    }

    float* yn = y + (int64_t)n * K * Ho * Wo;
    const bool vec = px0 + 1 < Wo && (((uintptr_t)yn | (uintptr_t)(Wo * 4)) & 7) == 0;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = kb * KB + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (k >= K || px0 >= Wo) continue;
            const float sc = ep.scale ? ep.scale[k] : 1.0f, sh = ep.shift ? ep.shift[k] : 0.0f;
            float o0 = fmaf(acc[kt][0][e], sc, sh), o1 = fmaf(acc[kt][1][e], sc, sh);
            if (ep.relu) { o0 = fmaxf(o0, 0.0f); o1 = fmaxf(o1, 0.0f); }
            float* q = yn + ((int64_t)k * Ho + py) * Wo + px0;
            if (vec) {
                *reinterpret_cast<float2*>(q) = make_float2(o0, o1);
            } else {
                q[0] = o0;
                if (px0 + 1 < Wo) q[1] = o1;
            }
        }
}

}  // namespace

extern "C" int vo_conv7x7_filter(const float* weight, float* arranged, int K, int C, void* stream) {
    VO_REQUIRE(weight && arranged, "vo_conv7x7_filter: null pointer");
    VO_REQUIRE(K > 0 && C > 0 && C <= 8, "vo_conv7x7_filter: bad sizes K=%d C=%d (at most 8 input channels)", K, C);
    const int Kpad = (K + KB - 1) / KB * KB;
    const int64_t n = (int64_t)Kpad * C * 56;
    filter7_kernel<<<dim3((unsigned)vo_ceil_div(n, 256)), dim3(256), 0, vo_stream(stream)>>>(weight, arranged, K, C, Kpad);
    VO_CHECK_LAUNCH("vo_conv7x7_filter");
    return VO_OK;
}

extern "C" int vo_conv7x7(const float* x, const float* arranged, float* y, const float* scale, const float* shift, int relu,
                          int N, int C, int H, int W, int K, int stride, void* stream) {
    VO_REQUIRE(x && arranged && y, "vo_conv7x7: null pointer");
    VO_REQUIRE(N > 0 && C > 0 && C <= 8 && H > 0 && W > 0 && K > 0, "vo_conv7x7: bad sizes");
    VO_REQUIRE(stride == 1 || stride == 2, "vo_conv7x7: stride must be 1 or 2");
    VO_REQUIRE(x != y, "vo_conv7x7: the output must not alias the input");
    if ((int64_t)C * H * W * 4 >= (int64_t)OOB)
        return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv7x7: one image of the input exceeds 1 GiB");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int bx = (Wo + 63) / 64, by = (Ho + 3) / 4, kblocks = (K + KB - 1) / KB;
    const int64_t blocks = (int64_t)bx * by * N * kblocks;
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv7x7: grid too large");
    const dim3 grid((unsigned)blocks);
    const Epi ep{scale, shift, relu};
    if (stride == 1) conv7_kernel<1><<<grid, dim3(256), 0, vo_stream(stream)>>>(x, arranged, y, ep, C, H, W, Ho, Wo, K, bx, by, kblocks);
    else conv7_kernel<2><<<grid, dim3(256), 0, vo_stream(stream)>>>(x, arranged, y, ep, C, H, W, Ho, Wo, K, bx, by, kblocks);
    VO_CHECK_LAUNCH("vo_conv7x7");
    return VO_OK;
}
