// 3x3 stride-2 padding-1 convolution in float32, NCHW, as a streaming matrix-core GEMM without LDS:
// the downsampling layers of the encoders' residual stages (64 -> 96 at full resolution, 96 -> 128
// at half resolution in both networks).
//     y[n,k,py,px] = act(scale[k] * sum_{r,s,c} w[k,c,r,s] x[n,c,2py+r-1,2px+s-1] + shift[k])
//
// The reduction runs tap by tap (r, s) and, inside a tap, over chunks of 8 input channels: one
// 16-byte load of the pre-arranged weights is the A operand of a chunk's four k-steps, and the B
// operand of k-step ks is the input plane of channel 2ks + (lane >> 5), sampled at every second
// pixel, read straight from memory.  A wave owns 64 (or 32) output channels x 64 consecutive pixels
// of one output row (pixel tile t = pixels 2n+t).  Columns outside the image get a lane offset past
// the buffer's extent (the range check returns 0.0f), rows outside the image are skipped; all
// address arithmetic of the loop is scalar.  No wave needs another: no LDS, no barrier.
#include "vo_common.h"

#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KB = 64;
constexpr int CB = 8;
constexpr int OOB = 0x40000000;            // lane offset no image reaches (images are below 1 GiB)

// Wg[kb][tap][chunk][kt][lane][ks]:  k = kb*64 + kt*32 + (lane & 31), tap = 3r + s, c = chunk*8 + 2*ks + (lane >> 5)
__global__ __launch_bounds__(256) void filter3s2_kernel(const float* __restrict__ w, float* __restrict__ Wg, int K, int C, int Kpad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Kpad * C * 9) return;
    const int tap = (int)(idx % 9), c = (int)((idx / 9) % C), k = (int)(idx / (9 * C));
    const float v = k < K ? w[((int64_t)k * C + c) * 9 + tap] : 0.0f;
    const int kb = k >> 6, kt = (k >> 5) & 1, kl = k & 31, chunk = c >> 3, ks = (c & 7) >> 1, kk = c & 1;
    const int nchunks = C / CB;
    Wg[((((((int64_t)kb * 9 + tap) * nchunks + chunk) * 2 + kt) * 64) + kk * 32 + kl) * 4 + ks] = v;
}

struct Epi { const float* scale; const float* shift; int relu; };

// KT: k-tiles of 32 output channels per wave (1 for a last block with at most 32 channels).
// ONE_TAP: the 1x1 stride-2 convolution of the projection shortcuts (only the centre tap exists; the
// weights are in vo_conv1x1_filter's order, which is this kernel's order with a single tap).
template <int KT, bool ONE_TAP>
__global__ __launch_bounds__(256) void conv3s2_kernel(const float* __restrict__ x, const float* __restrict__ Wg,
                                                      float* __restrict__ y, Epi ep, int C, int H, int W, int Ho, int Wo, int K,
                                                      int blocks_x, int blocks_y, int kblocks, int kb_base) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bid = blockIdx.x;
    const int kb = kb_base + bid % kblocks; bid /= kblocks;
    const int bx = bid % blocks_x; bid /= blocks_x;
    const int by = bid % blocks_y;
    const int n = bid / blocks_y;
    const int py = by * 4 + wave;                                // this wave's output row
    if (py >= Ho) return;
    const int kk = lane >> 5, nn = lane & 31;
    const int px0 = bx * 64 + 2 * nn;                            // this lane's first output pixel of the row
    const int nchunks = C / CB;
    const int plane = H * W * 4;                                 // bytes of one input plane

    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (int64_t)n * C * H * W), 0, C * plane, 0x00020000);
    constexpr int TAPS = ONE_TAP ? 1 : 9;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wg + (int64_t)kb * TAPS * nchunks * 512), 0, TAPS * nchunks * 512 * 4, 0x00020000);
    // lane part of the byte offset of the element (filter column s, pixel tile t): the lane's channel
    // parity plane plus the column, or OOB for the padding columns
    int coff[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = 2 * (px0 + t) + s - 1;
            coff[s][t] = (col >= 0 && col < W) ? kk * plane + col * 4 : OOB;
        }
    const int w_off = lane * 16;

    f32x16 acc[KT][2];
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][t][e] = 0.0f;

    const int iy = 2 * py - 1;                                   // input row of filter row 0
    const int r_lo = ONE_TAP ? 1 : (iy < 0 ? 1 : 0), r_hi = ONE_TAP ? 1 : (iy + 2 >= H ? H - 1 - iy : 2);   // valid filter rows
    const int total = (r_hi - r_lo + 1) * nchunks;               // (filter row, chunk) pairs; three filter columns each

    float bcur[4][2], bnxt[4][2];
    u32x4 acur[KT], anxt[KT];
    int lr = r_lo, lc = 0;                                       // (filter row, chunk) of the loads being issued
    auto load = [&](auto stag) {                                 // the filter column is a compile-time constant
        constexpr int ls = decltype(stag)::value;
        const int tap = ONE_TAP ? 0 : 3 * lr + ls;
        const int soff = (lc * CB) * plane + (iy + lr) * W * 4;  // wave-uniform: chunk's first plane, input row
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
            anxt[kt] = __builtin_amdgcn_raw_buffer_load_b128(wrs, w_off + kt * 1024, (tap * nchunks + lc) * 2048, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                bnxt[ks][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, coff[ls][t], soff + 2 * ks * plane, 0));
    };
    auto step = [&]() {                                          // take the loaded operands, multiply
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    acc[kt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(acur[kt][ks]), bcur[ks][t], acc[kt][t], 0, 0, 0);
    };
    auto take = [&]() {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acur[kt] = anxt[kt];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { bcur[ks][0] = bnxt[ks][0]; bcur[ks][1] = bnxt[ks][1]; }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    if (ONE_TAP) {
        load(S1{});
        for (int i = 0; i < total; ++i) {
            take();
            if (lc + 1 < nchunks) ++lc;                          // past the end: the last chunk again, never used
            load(S1{});
            step();
        }
    } else {
        load(S0{});
        for (int i = 0; i < total; ++i) {
            take(); load(S1{}); step();
            take(); load(S2{}); step();
            take();
            if (++lc == nchunks) { lc = 0; if (lr < r_hi) ++lr; }    // past the end: a valid pair again, never used
            load(S0{});
            step();
        }
    }

    float* yn = y + (int64_t)n * K * Ho * Wo;
    const bool vec = px0 + 1 < Wo && (((uintptr_t)yn | (uintptr_t)(Wo * 4)) & 7) == 0;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
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

extern "C" int vo_conv3x3s2_filter(const float* weight, float* arranged, int K, int C, void* stream) {
    VO_REQUIRE(weight && arranged, "vo_conv3x3s2_filter: null pointer");
    VO_REQUIRE(K > 0 && C > 0, "vo_conv3x3s2_filter: bad sizes K=%d C=%d", K, C);
    if (C % CB != 0) return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv3x3s2_filter: C=%d is not a multiple of %d", C, CB);
    const int Kpad = (K + KB - 1) / KB * KB;
    const int64_t n = (int64_t)Kpad * C * 9;
    filter3s2_kernel<<<dim3((unsigned)vo_ceil_div(n, 256)), dim3(256), 0, vo_stream(stream)>>>(weight, arranged, K, C, Kpad);
    VO_CHECK_LAUNCH("vo_conv3x3s2_filter");
    return VO_OK;
}

template <bool ONE_TAP>
static int launch_s2(const char* name, const float* x, const float* arranged, float* y, const float* scale, const float* shift,
                     int relu, int N, int C, int H, int W, int K, void* stream) {
    if (!(x && arranged && y)) return vo_fail(VO_ERR_INVALID_ARG, "%s: null pointer", name);
    if (!(N > 0 && C > 0 && H > 0 && W > 0 && K > 0)) return vo_fail(VO_ERR_INVALID_ARG, "%s: bad sizes", name);
    if (x == y) return vo_fail(VO_ERR_INVALID_ARG, "%s: the output must not alias the input", name);
    if (C % CB != 0) return vo_fail(VO_ERR_UNSUPPORTED, "%s: C=%d is not a multiple of %d", name, C, CB);
    if ((int64_t)C * H * W * 4 >= (int64_t)OOB) return vo_fail(VO_ERR_UNSUPPORTED, "%s: one image of the input exceeds 1 GiB", name);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int bx = (Wo + 63) / 64, by = (Ho + 3) / 4, kblocks = (K + KB - 1) / KB;
    const int64_t blocks = (int64_t)bx * by * N;
    if (blocks * kblocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "%s: grid too large", name);
    const Epi ep{scale, shift, relu};
    hipStream_t st = vo_stream(stream);
    const int rem = K % KB;
    const int full = (rem >= 1 && rem <= 32) ? kblocks - 1 : kblocks;         // a last block of <= 32 channels: one k-tile
    if (full > 0)
        conv3s2_kernel<2, ONE_TAP><<<dim3((unsigned)(blocks * full)), dim3(256), 0, st>>>(x, arranged, y, ep, C, H, W, Ho, Wo, K, bx,
                                                                                          by, full, 0);
    if (full < kblocks)
        conv3s2_kernel<1, ONE_TAP><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(x, arranged, y, ep, C, H, W, Ho, Wo, K, bx, by, 1,
                                                                                 full);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e));
    return VO_OK;
}

extern "C" int vo_conv3x3s2(const float* x, const float* arranged, float* y, const float* scale, const float* shift, int relu,
                            int N, int C, int H, int W, int K, void* stream) {
    return launch_s2<false>("vo_conv3x3s2", x, arranged, y, scale, shift, relu, N, C, H, W, K, stream);
}

extern "C" int vo_conv1x1s2(const float* x, const float* arranged, float* y, const float* scale, const float* shift, int relu,
                            int N, int C, int H, int W, int K, void* stream) {
    return launch_s2<true>("vo_conv1x1s2", x, arranged, y, scale, shift, relu, N, C, H, W, K, stream);
}
