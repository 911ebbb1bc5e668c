// Convex upsampling of the coarse flow / disparity field.
//
// Every fine pixel (f*y+i, f*x+j) is a softmax-weighted combination of the 3x3
// coarse neighbourhood of (y,x), the 9 logits living in mask channels
// k*f*f + i*f + j.  Stock PyTorch materialises view/softmax/unfold/mul/sum/permute
// intermediates (several passes over the 9*f*f-channel mask); here the mask is
// read exactly once and the fine field written exactly once.
//
// One lane per (pair, sub-row i, coarse pixel), lanes along the coarse pixel
// index: every mask load is a fully coalesced 256-byte wave access, and each lane
// stores its f consecutive fine pixels as 16-byte vectors, which are contiguous
// across the wave.
#include "vo_common.h"

namespace {

template <int F, int COUT>
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ flow, const float* __restrict__ mask,
                                                       float* __restrict__ out, int64_t total, int Cf, int H, int W,
                                                       float mask_scale) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int HW = H * W;
    const int p = (int)(t % HW);
    const int i = (int)((t / HW) % F);
    const int64_t b = t / ((int64_t)HW * F);
    const int y = p / W, x = p - y * W;

    // 3x3 neighbourhood of f*flow, zero outside the image (unfold with padding 1)
    float nb[COUT][9];
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        const float* fl = flow + (b * Cf + c) * HW;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            nb[c][k] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (float)F * fl[yy * W + xx] : 0.0f;
        }
    }

    const float* m = mask + (b * 9 * F * F + (int64_t)i * F) * HW + p;
    float res[COUT][F];
#pragma unroll
    for (int j = 0; j < F; ++j) {
        float e[9];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            e[k] = m[((int64_t)k * F * F + j) * HW] * mask_scale;
            mx = fmaxf(mx, e[k]);
        }
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { e[k] = expf(e[k] - mx); s += e[k]; }
        const float inv = 1.0f / s;
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc += (e[k] * inv) * nb[c][k];
            res[c][j] = acc;
        }
    }

    const int FW = F * W;
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        float* o = out + ((b * COUT + c) * (int64_t)(F * H) + (F * y + i)) * FW + F * x;
        if constexpr (F % 4 == 0) {
#pragma unroll
            for (int j = 0; j < F; j += 4)
                *reinterpret_cast<float4*>(o + j) = make_float4(res[c][j], res[c][j + 1], res[c][j + 2], res[c][j + 3]);
        } else {
#pragma unroll
            for (int j = 0; j < F; ++j) o[j] = res[c][j];
        }
    }
}

template <int F>
int launch(const float* flow, const float* mask, float* out, int batch, int Cf, int Cout, int H, int W, float scale,
           hipStream_t stream) {
    const int64_t total = (int64_t)batch * F * H * W;
    const int64_t blocks = vo_ceil_div(total, 256);
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_convex_upsample: grid too large");
    if (Cout == 1)
        upsample_kernel<F, 1><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(flow, mask, out, total, Cf, H, W, scale);
    else
        upsample_kernel<F, 2><<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(flow, mask, out, total, Cf, H, W, scale);
    VO_CHECK_LAUNCH("vo_convex_upsample");
    return VO_OK;
}

}  // namespace

extern "C" int vo_convex_upsample(const float* flow, const float* mask, float* out, int batch, int Cf, int Cout, int H,
                                  int W, int factor, float mask_scale, void* stream) {
    VO_REQUIRE(flow && mask && out, "vo_convex_upsample: null pointer");
    VO_REQUIRE(batch > 0 && H > 0 && W > 0, "vo_convex_upsample: bad sizes batch=%d H=%d W=%d", batch, H, W);
    VO_REQUIRE(Cout >= 1 && Cout <= 2 && Cout <= Cf, "vo_convex_upsample: Cout=%d must be 1 or 2 and <= Cf=%d", Cout, Cf);
    VO_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "vo_convex_upsample: out must be 16-byte aligned");
    VO_REQUIRE((int64_t)H * W * factor * factor <= 0x7fffffffLL, "vo_convex_upsample: image too large");
    switch (factor) {
        case 2: return launch<2>(flow, mask, out, batch, Cf, Cout, H, W, mask_scale, vo_stream(stream));
        case 4: return launch<4>(flow, mask, out, batch, Cf, Cout, H, W, mask_scale, vo_stream(stream));
        case 8: return launch<8>(flow, mask, out, batch, Cf, Cout, H, W, mask_scale, vo_stream(stream));
        default: return vo_fail(VO_ERR_UNSUPPORTED, "vo_convex_upsample: factor %d (built for 2, 4, 8)", factor);
    }
}
