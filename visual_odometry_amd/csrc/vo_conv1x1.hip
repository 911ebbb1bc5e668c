// 1x1 stride-1 convolution in float32, NCHW, as a streaming matrix-core GEMM without LDS:
//     y[n, k, p] = act( scale[k] * sum_c W[k, c] * x[n, c, p] + shift[k] ),   p = pixel index in the plane.
// The networks use it for the motion encoders' first layer (324 -> 256, 36 -> 64), the encoders'
// projection shortcuts, the mask heads and the per-tap planes of the two-channel flow heads
// (256 -> 18); the library GEMMs behind the framework's convolution reach 16-80 TFLOP/s on these
// skinny shapes and several of them are bounded by reading x once.
//
// x is already the B operand of v_mfma_f32_32x32x2_f32: B[k = channel][n = pixel] with pixels
// contiguous.  A lane loads NTL consecutive pixels of one channel (8 or 16 bytes) and uses component t
// in pixel tile t, i.e. tile t covers the pixels NTL*n + t: a permutation that lets the same lane
// store NTL consecutive output pixels of a channel with one instruction.  Weights are pre-arranged
// in the A operand's register order (one 16-byte load = four k-steps).  A wave owns 64 (or 32) output
// channels x 64 (or 128) pixels = 64 accumulator registers, needs no other wave, and keeps the loads
// of the next chunk of 8 channels in flight while it multiplies.
#include "vo_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int CB = 8;
constexpr int KB = 64;

// Wg[kb][chunk][kt][lane][ks]:  k = kb*64 + kt*32 + (lane & 31),  c = chunk*8 + 2*ks + (lane >> 5); zero outside (K, C)
__global__ __launch_bounds__(256) void filter1x1_kernel(const float* __restrict__ w, float* __restrict__ Wg, int K, int C,
                                                        int Kpad, int Cpad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Kpad * Cpad) return;
    const int k = (int)(idx / Cpad), c = (int)(idx % Cpad);
    const float v = (k < K && c < C) ? w[(int64_t)k * C + c] : 0.0f;
    const int kb = k >> 6, kt = (k >> 5) & 1, kl = k & 31, chunk = c >> 3, ks = (c & 7) >> 1, kk = c & 1;
    const int nchunks = Cpad / CB;
    Wg[(((((int64_t)kb * nchunks + chunk) * 2 + kt) * 64) + kk * 32 + kl) * 4 + ks] = v;
}

struct Epi { const float* scale; const float* shift; int relu; int64_t y_batch_stride; };

// KT k-tiles of 32 channels and NTL pixel tiles of 32 pixels per wave (64 or 128 accumulator registers).
// PM: the output is written pixel-major, y[n][p][k] (each pixel's K values contiguous: the layout the
// correlation kernels read), four consecutive channels per 16-byte store.
template <int KT, int NTL, bool PM>
__global__ __launch_bounds__(256, 3) void conv1x1_kernel(const float* __restrict__ x, const float* __restrict__ Wg,
                                                      float* __restrict__ y, Epi ep, int C, int P, int K, int pblocks,
                                                      int kblocks) {
    constexpr int PW = 32 * NTL;                                 // pixels per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int bid = blockIdx.x;
    const int kb = bid % kblocks; bid /= kblocks;
    const int pb = bid % pblocks;
    const int n = bid / pblocks;
    const int nchunks = (C + CB - 1) / CB;
    const int kk = lane >> 5, nn = lane & 31;
    const int p0 = (pb * 4 + wave) * PW + NTL * nn;              // this lane's first pixel
    if ((pb * 4 + wave) * PW >= P) return;                       // the wave has no pixels (waves are independent)

    // one image of x: channels past C and bytes past the image read as 0.0f
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (int64_t)n * C * P), 0, C * P * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wg + (int64_t)kb * nchunks * (2 * 64 * 4)), 0, nchunks * (2 * 64 * 4) * 4, 0x00020000);
    const int x_off = (kk * P + p0) * 4;                         // + (chunk*8 + 2*ks) * P * 4
    const int w_off = lane * 16;                                 // + kt * 1024 + chunk * 2048

    f32x16 acc[KT][NTL];
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][t][e] = 0.0f;

    float bcur[4][NTL], bnxt[4][NTL];
    u32x4 acur[KT], anxt[KT];
    auto load = [&](int chunk) {
        chunk = chunk < nchunks ? chunk : nchunks - 1;           // past the end: reload the last chunk, never used
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) anxt[kt] = __builtin_amdgcn_raw_buffer_load_b128(wrs, w_off + kt * 1024, chunk * 2048, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int vo = x_off + (chunk * CB + 2 * ks) * P * 4;       // in the lane offset, which the range check covers
            if (NTL == 4) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) bnxt[ks][t] = __uint_as_float(v[t]);
            } else {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, vo, 0, 0);
#pragma unroll
                for (int t = 0; t < 2; ++t) bnxt[ks][t] = __uint_as_float(v[t]);
            }
        }
    };
    load(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acur[kt] = anxt[kt];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < NTL; ++t) bcur[ks][t] = bnxt[ks][t];
        load(chunk + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < NTL; ++t)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    acc[kt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(acur[kt][ks]), bcur[ks][t], acc[kt][t], 0, 0, 0);
    }

    // All scale and shift values are loaded before the first store: the wait counter of vector memory
    // operations counts loads and stores in one order, so a load issued between stores could only be
    // waited for together with the stores before it.  Channels past K read as 0 and are not stored.
    float* yn = y + (int64_t)n * ep.y_batch_stride;
    float sc[KT][16], sh[KT][16];
    const __amdgpu_buffer_rsrc_t scr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ep.scale), 0, ep.scale ? K * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t shr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ep.shift), 0, ep.shift ? K * 4 : 0, 0x00020000);
    auto load_affine = [&](int kt, int e0, int count) {
#pragma unroll
        for (int e = e0; e < e0 + count; ++e) {
            const int off = (kb * KB + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk) * 4;
            sc[kt][e] = ep.scale ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(scr, off, 0, 0)) : 1.0f;
            sh[kt][e] = ep.shift ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(shr, off, 0, 0)) : 0.0f;
        }
    };
    constexpr bool ALL_FIRST = KT * NTL <= 4;                    // with 128 accumulators there is room for 8 channels' values only
    if (ALL_FIRST) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) load_affine(kt, 0, 16);
    }
    if (PM) {
        // a lane's elements e..e+3 are the channels k0..k0+3 of its pixels
        float* yp = y + (int64_t)n * P * K;
        const bool quad = (K & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (!ALL_FIRST && (g & 1) == 0) load_affine(kt, 4 * g, 8);
                const int k0 = kb * KB + kt * 32 + 8 * g + 4 * kk;
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    const int p = p0 + t;
                    if (p >= P || k0 >= K) continue;
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = fmaf(acc[kt][t][4 * g + j], sc[kt][4 * g + j], sh[kt][4 * g + j]);
                        if (ep.relu) o[j] = fmaxf(o[j], 0.0f);
                    }
                    float* q = yp + (int64_t)p * K + k0;
                    if (quad) {
                        *reinterpret_cast<float4*>(q) = make_float4(o[0], o[1], o[2], o[3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (k0 + j < K) q[j] = o[j];
                    }
                }
            }
        }
        return;
    }
    const bool vec = p0 + NTL <= P && ((reinterpret_cast<uintptr_t>(yn) | (uintptr_t)((int64_t)P * 4) | (uintptr_t)(p0 * 4)) & (NTL * 4 - 1)) == 0;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (!ALL_FIRST && (e & 7) == 0) load_affine(kt, e, 8);
            const int k = kb * KB + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (k >= K || p0 >= P) continue;
            float o[NTL];
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                o[t] = fmaf(acc[kt][t][e], sc[kt][e], sh[kt][e]);
                if (ep.relu) o[t] = fmaxf(o[t], 0.0f);
            }
            float* q = yn + (int64_t)k * P + p0;
            if (vec) {
                if (NTL == 4) *reinterpret_cast<float4*>(q) = make_float4(o[0], o[1], o[2 % NTL], o[3 % NTL]);
                else *reinterpret_cast<float2*>(q) = make_float2(o[0], o[1]);
            } else {
#pragma unroll
                for (int t = 0; t < NTL; ++t)
                    if (p0 + t < P) q[t] = o[t];
            }
        }
    }
}

}  // namespace

extern "C" int vo_conv1x1_filter(const float* weight, float* arranged, int K, int C, void* stream) {
    VO_REQUIRE(weight && arranged, "vo_conv1x1_filter: null pointer");
    VO_REQUIRE(K > 0 && C > 0, "vo_conv1x1_filter: bad sizes K=%d C=%d", K, C);
    const int Kpad = (K + KB - 1) / KB * KB, Cpad = (C + CB - 1) / CB * CB;
    const int64_t n = (int64_t)Kpad * Cpad;
    filter1x1_kernel<<<dim3((unsigned)vo_ceil_div(n, 256)), dim3(256), 0, vo_stream(stream)>>>(weight, arranged, K, C, Kpad, Cpad);
    VO_CHECK_LAUNCH("vo_conv1x1_filter");
    return VO_OK;
}

template <bool PM>
static int launch_1x1(const float* x, const float* arranged, float* y, int64_t y_batch_stride, const float* scale,
                      const float* shift, int relu, int N, int C, int64_t P, int K, void* stream) {
    VO_REQUIRE(x && arranged && y, "vo_conv1x1: null pointer");
    VO_REQUIRE(N > 0 && C > 0 && P > 0 && K > 0, "vo_conv1x1: bad sizes");
    VO_REQUIRE(x != y, "vo_conv1x1: the output must not alias the input");
    if (y_batch_stride == 0) y_batch_stride = (int64_t)K * P;
    VO_REQUIRE(y_batch_stride >= (int64_t)K * P, "vo_conv1x1: y_batch_stride smaller than one item");
    if ((int64_t)(C + CB) * P * 4 >= 0x7fffffffLL)
        return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv1x1: one image of the input exceeds 2 GiB");
    const int kblocks = (K + KB - 1) / KB;
    const Epi ep{scale, shift, relu, y_batch_stride};
    hipStream_t st = vo_stream(stream);
    if (K <= 32) {                                               // one k-tile, 128 pixels per wave
        const int64_t pblocks = vo_ceil_div(P, 4 * 128);
        if (pblocks * N > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv1x1: grid too large");
        conv1x1_kernel<1, 4, PM><<<dim3((unsigned)(pblocks * N)), dim3(256), 0, st>>>(x, arranged, y, ep, C, (int)P, K, (int)pblocks, 1);
    } else {
        const int64_t pblocks = vo_ceil_div(P, 4 * 64);
        if (pblocks * N * kblocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv1x1: grid too large");
        if (C <= 128) {                                          // short reductions: 128 pixels per wave (measured faster)
            const int64_t pb4 = vo_ceil_div(P, 4 * 128);
            conv1x1_kernel<2, 4, PM><<<dim3((unsigned)(pb4 * N * kblocks)), dim3(256), 0, st>>>(x, arranged, y, ep, C, (int)P, K,
                                                                                           (int)pb4, kblocks);
        } else {
            conv1x1_kernel<2, 2, PM><<<dim3((unsigned)(pblocks * N * kblocks)), dim3(256), 0, st>>>(x, arranged, y, ep, C, (int)P, K,
                                                                                               (int)pblocks, kblocks);
        }
    }
    VO_CHECK_LAUNCH("vo_conv1x1");
    return VO_OK;
}

extern "C" int vo_conv1x1(const float* x, const float* arranged, float* y, int64_t y_batch_stride, const float* scale,
                          const float* shift, int relu, int N, int C, int64_t P, int K, void* stream) {
    return launch_1x1<false>(x, arranged, y, y_batch_stride, scale, shift, relu, N, C, P, K, stream);
}

extern "C" int vo_conv1x1_pixel_major(const float* x, const float* arranged, float* y, const float* scale, const float* shift,
                                      int relu, int N, int C, int64_t P, int K, void* stream) {
    return launch_1x1<true>(x, arranged, y, 0, scale, shift, relu, N, C, P, K, stream);
}
