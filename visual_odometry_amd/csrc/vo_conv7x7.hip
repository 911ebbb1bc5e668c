// 7x7 padding-3 convolution (stride 1 or 2) with very few input channels (the networks' stems read the
// 3-channel image, the motion encoders the 2-channel flow) in float32 as a streaming matrix-core
// GEMM without LDS:   y[n,k,py,px] = act(scale[k] * sum_{c,r,s} w[k,c,r,s] x[n,c,py+r-3,px+s-3] + shift[k]).
//
// The reduction index is ordered (c, r, s) with s padded from 7 to 8, so one 16-byte load of the
// pre-arranged weights is the A operand of the four k-steps of one filter row (k-step ks holds the
// taps s = 2ks and 2ks+1).  A wave owns 64 output channels x 64 consecutive pixels of one image row
// (pixel tile t = pixels 32t..32t+31, 64 accumulator registers: neighbouring lanes read neighbouring
// input pixels, so a load touches as few cache lines as the stride allows); its B operand for tap s is
// the input row shifted by s-3, read straight from memory (the 49-fold reuse lives in L1/L2).  Columns outside the
// image get a lane offset past the buffer's extent (the range check returns 0.0f); rows outside
// the image are skipped altogether.  The four waves of a workgroup take four consecutive rows.
#include "vo_common.h"

#include <type_traits>

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
__global__ __launch_bounds__(256, 3) void conv7_kernel(const float* __restrict__ x, const float* __restrict__ Wg,
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
    const int px0 = bx * 64 + nn;                                // this lane's output pixels of the row: px0 and px0 + 32
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
            const int col = S * (px0 + 32 * t) + 2 * ks + kk - 3;
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

    // the (channel, filter row) pairs whose input row lies inside the image, in order
    const int iy = S * py - 3;                                   // input row of filter row 0
    const int r_lo = iy < 0 ? -iy : 0, r_hi = iy + 6 >= H ? H - 1 - iy : 6;         // valid filter rows r_lo..r_hi
    const int total = C * (r_hi - r_lo + 1);                     // at most 8 * 7
    // Three operand buffers in rotation: the loads of pair i+3 are issued right after the
    // multiplications of pair i, so every load has two pairs' worth of matrix instructions (of this wave
    // alone) to arrive.  The fences keep that issue order, so each wait covers one buffer's ten loads.
    // The loop body has no branch (the waits stay exact and the accumulators stay in place): the
    // number of pairs is rounded up to a multiple of three, and a pair past the end reads its weights
    // from past the buffer's extent, i.e. multiplies by zero.
    float b[3][4][2];
    u32x4 a[3][2];
    const int nr = r_hi - r_lo + 1;
    const int inv_nr = (65536 + nr - 1) / nr;                    // i / nr == (i * inv_nr) >> 16 for the i < 56 that occur
    int issued = 0;                                              // pairs whose loads have been issued
    auto load = [&](auto btag) {
        constexpr int lb = decltype(btag)::value;
        // wave-uniform, said explicitly so that no load is wrapped in a loop over lanes
        const int ii = min(issued, total - 1);                   // past the end: the last pair's input again
        const int lc = (ii * inv_nr) >> 16, lr = r_lo + ii - lc * nr;      // branch-free on purpose (see above)
        const int woff = __builtin_amdgcn_readfirstlane((lc * 7 + lr) * 2048);
        const int row = __builtin_amdgcn_readfirstlane(((lc * H + iy + lr) * W) * 4);
        const int wv = issued < total ? w_off : OOB;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) a[lb][kt] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wv + kt * 1024, woff, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                b[lb][ks][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, coff[ks][t], row, 0));
        ++issued;
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](auto btag) {
        constexpr int lb = decltype(btag)::value;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    acc[kt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[lb][kt][ks]), b[lb][ks][t], acc[kt][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using B2 = std::integral_constant<int, 2>;
    load(B0{}); load(B1{}); load(B2{});
    // The compiler interleaves these first thirty loads and would then make every pass of the loop wait
    // as if they were still in that order.  Naming all their registers here makes it wait for them
    // once, before the loop; with nothing outstanding at the loop's entry the waits inside follow the
    // loop's own order (the oldest ten loads before each step).
#pragma unroll
    for (int q = 0; q < 3; ++q)
        asm volatile("" : "+v"(a[q][0]), "+v"(a[q][1]), "+v"(b[q][0][0]), "+v"(b[q][0][1]), "+v"(b[q][1][0]), "+v"(b[q][1][1]),
                          "+v"(b[q][2][0]), "+v"(b[q][2][1]), "+v"(b[q][3][0]), "+v"(b[q][3][1]));
    for (int i = 0; i < total; i += 3) {
        step(B0{}); load(B0{});
        step(B1{}); load(B1{});
        step(B2{}); load(B2{});
    }

    // All scale and shift values are loaded before the first store: the wait counter of vector memory
    // operations counts loads and stores in one order, so a load issued between stores could only be
    // waited for together with the stores before it.  Channels past K read as 0 and are not stored.
    float* yn = y + (int64_t)n * K * Ho * Wo;
    float sc[2][16], sh[2][16];
    const __amdgpu_buffer_rsrc_t scr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ep.scale), 0, ep.scale ? K * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t shr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ep.shift), 0, ep.shift ? K * 4 : 0, 0x00020000);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int off = (kb * KB + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk) * 4;
            sc[kt][e] = ep.scale ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(scr, off, 0, 0)) : 1.0f;
            sh[kt][e] = ep.shift ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(shr, off, 0, 0)) : 0.0f;
        }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = kb * KB + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (k >= K) continue;
            float* q = yn + ((int64_t)k * Ho + py) * Wo + px0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float o = fmaf(acc[kt][t][e], sc[kt][e], sh[kt][e]);
                if (ep.relu) o = fmaxf(o, 0.0f);
                if (px0 + 32 * t < Wo) q[32 * t] = o;
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
