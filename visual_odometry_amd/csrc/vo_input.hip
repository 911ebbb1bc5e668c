// Input stage of both networks in one pass: 8-bit gray frame -> three equal float32
// channels (or the one plane they all equal), replicate-padded to the network's stride multiple and (optionally)
// normalised to [-1, 1].  Replaces the host-side PIL / NumPy conversion of the
// reference wrappers (flow/flow_map_RAFT.py:91-106, stereo/stereo_disparity_RAFT.py:97-110),
// the InputPadder call (:71-72 / :81-82) and the 2*(x/255)-1 on entry of the models.
// HBM-bound: 1 byte read, 12 bytes written per padded pixel; one lane per 4 output
// pixels of a row (16-byte stores), the three channel planes written from registers.
#include "vo_common.h"

namespace {

__global__ __launch_bounds__(256) void prepare_kernel(const uint8_t* __restrict__ gray, float* __restrict__ out,
                                                      int64_t total, int H, int W, int Hp, int Wp, int pad_l,
                                                      int pad_t, int normalize, int channels) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int wq = Wp >> 2;                       // Wp is a multiple of 4 (checked on the host)
    const int xq = (int)(t % wq);
    const int y = (int)((t / wq) % Hp);
    const int64_t f = t / ((int64_t)wq * Hp);
    const int sy = min(max(y - pad_t, 0), H - 1);
    const uint8_t* row = gray + (f * H + sy) * (int64_t)W;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sx = min(max(4 * xq + i - pad_l, 0), W - 1);
        const float g = (float)row[sx];
        v[i] = normalize ? 2.0f * (g / 255.0f) - 1.0f : g;     // same operation order as the models' entry
    }
    const float4 o = make_float4(v[0], v[1], v[2], v[3]);
    float* dst = out + ((f * channels) * Hp + y) * (int64_t)Wp + 4 * xq;
    const int64_t plane = (int64_t)Hp * Wp;
    *reinterpret_cast<float4*>(dst) = o;
    if (channels == 3) {
        *reinterpret_cast<float4*>(dst + plane) = o;
        *reinterpret_cast<float4*>(dst + 2 * plane) = o;
    }
}

int prepare(const char* name, const uint8_t* gray, float* out, int frames, int H, int W, int pad_left, int pad_right,
            int pad_top, int pad_bottom, int normalize, int channels, hipStream_t st) {
    if (!gray || !out) return vo_fail(VO_ERR_INVALID_ARG, "%s: null pointer", name);
    if (frames <= 0 || H <= 0 || W <= 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: bad sizes frames=%d H=%d W=%d", name, frames, H, W);
    if (pad_left < 0 || pad_right < 0 || pad_top < 0 || pad_bottom < 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: negative padding", name);
    const int Hp = H + pad_top + pad_bottom, Wp = W + pad_left + pad_right;
    if (Wp % 4 != 0) return vo_fail(VO_ERR_UNSUPPORTED, "%s: padded width %d is not a multiple of 4", name, Wp);
    if ((reinterpret_cast<uintptr_t>(out) & 15) != 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: out must be 16-byte aligned", name);
    const int64_t total = (int64_t)frames * Hp * (Wp / 4);
    const int64_t blocks = vo_ceil_div(total, 256);
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "%s: grid too large", name);
    prepare_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(gray, out, total, H, W, Hp, Wp, pad_left, pad_top, normalize,
                                                                 channels);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e));
    return VO_OK;
}

}  // namespace

extern "C" int vo_prepare_images(const uint8_t* gray, float* out, int frames, int H, int W, int pad_left,
                                 int pad_right, int pad_top, int pad_bottom, int normalize, void* stream) {
    return prepare("vo_prepare_images", gray, out, frames, H, W, pad_left, pad_right, pad_top, pad_bottom, normalize, 3,
                   vo_stream(stream));
}

extern "C" int vo_prepare_images_gray(const uint8_t* gray, float* out, int frames, int H, int W, int pad_left,
                                      int pad_right, int pad_top, int pad_bottom, int normalize, void* stream) {
    return prepare("vo_prepare_images_gray", gray, out, frames, H, W, pad_left, pad_right, pad_top, pad_bottom, normalize, 1,
                   vo_stream(stream));
}
