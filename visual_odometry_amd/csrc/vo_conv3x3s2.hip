// 3x3 stride-2 padding-1 convolution in float32, NCHW, as a streaming matrix-core GEMM without LDS:
// the downsampling layers of the encoders' residual stages (64 -> 96 at full resolution, 96 -> 128
// at half resolution in both networks), and the 1x1 stride-2 convolution of their projection shortcuts.
//     y[n,k,py,px] = act(scale[k] * sum_{r,s,c} w[k,c,r,s] x[n,c,2py+r-1,2px+s-1] + shift[k])
//
// A wave owns 64 (or 32) output channels x 64 consecutive pixels of one output row (pixel tile t =
// pixels 32t..32t+31, so neighbouring lanes read neighbouring input pixels) and needs no other wave: no
// LDS, no barrier.  The reduction runs over units of (filter row, 4 input channels, all three filter
// columns): the B operand of k-step j is the input plane of channel 2j + (lane >> 5) sampled at every
// second pixel, read straight from memory.  On maps of even width one 8-byte load fetches the
// columns 2px and 2px+1 (filter columns 1 and 2) and a 4-byte load column 2px-1, so every cache line
// that is fetched is used in full; the kernel is bound by the rate of these loads, not by their
// latency.  Columns outside the image get a lane offset past the buffer's extent (the range check
// returns 0.0f), rows outside the image are skipped; all address arithmetic of the loop is scalar.
#include "vo_common.h"

#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int KB = 64;
constexpr int CB = 8;
constexpr int OOB = 0x40000000;            // lane offset no image reaches (images are below 1 GiB)
constexpr int UNIT = 3 * 2 * 64 * 2;       // floats of one unit's weights: [s][kt][lane][j]

// Wg[kb][r][chunk][h][s][kt][lane][j]:  k = kb*64 + kt*32 + (lane & 31),  c = chunk*8 + 2*(2h + j) + (lane >> 5)
__global__ __launch_bounds__(256) void filter3s2_kernel(const float* __restrict__ w, float* __restrict__ Wg, int K, int C, int Kpad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Kpad * C * 9) return;
    const int tap = (int)(idx % 9), c = (int)((idx / 9) % C), k = (int)(idx / (9 * C));
    const float v = k < K ? w[((int64_t)k * C + c) * 9 + tap] : 0.0f;
    const int r = tap / 3, sx = tap % 3;
    const int kb = k >> 6, kt = (k >> 5) & 1, kl = k & 31, chunk = c >> 3, ks = (c & 7) >> 1, kk = c & 1, h = ks >> 1, j = ks & 1;
    const int nchunks = C / CB;
    Wg[((((((((int64_t)kb * 3 + r) * nchunks + chunk) * 2 + h) * 3 + sx) * 2 + kt) * 64) + kk * 32 + kl) * 2 + j] = v;
}

struct Epi { const float* scale; const float* shift; int relu; };

// scale / shift / ReLU and store of a wave's accumulators: lane holds the pixels px0 and px0 + 32 of row py.
// All scale and shift values are loaded before the first store: the wait counter of vector memory
// operations counts loads and stores in one order, so a load issued between stores could only be
// waited for together with the stores before it (a write round trip per channel).  Channels past K
// read as 0 through the range check and are not stored.
// ROWS2: the wave's two pixel tiles are the same 32 pixels of the rows py and py + 1 instead of 64 pixels of row py.
template <int KT, bool ROWS2 = false>
__device__ __forceinline__ void store_tiles(const f32x16 (&acc)[KT][2], float* __restrict__ yn, const Epi& ep, int kb, int kk,
                                            int K, int Ho, int Wo, int py, int px0) {
    float sc[KT][16], sh[KT][16];
    const __amdgpu_buffer_rsrc_t scr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ep.scale), 0, ep.scale ? K * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t shr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ep.shift), 0, ep.shift ? K * 4 : 0, 0x00020000);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int off = (kb * KB + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk) * 4;
            sc[kt][e] = ep.scale ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(scr, off, 0, 0)) : 1.0f;
            sh[kt][e] = ep.shift ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(shr, off, 0, 0)) : 0.0f;
        }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = kb * KB + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (k >= K) continue;
            float* q = yn + ((int64_t)k * Ho + py) * Wo + px0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float o = fmaf(acc[kt][t][e], sc[kt][e], sh[kt][e]);
                if (ep.relu) o = fmaxf(o, 0.0f);
                if (ROWS2) {
                    if (py + t < Ho && px0 < Wo) q[t * Wo] = o;
                } else {
                    if (px0 + 32 * t < Wo) q[32 * t] = o;
                }
            }
        }
}

// KT: k-tiles of 32 output channels per wave (1 for a last block with at most 32 channels).
// PAIR: the width is even and x is 8-byte aligned, so the columns 2px and 2px+1 come with one load.
// ROWS2: a wave owns 32 pixels of two consecutive output rows instead of 64 pixels of one (output widths
// that fill the last 64-pixel segment less than half); filter rows outside the image then read zeros
// instead of being skipped.
template <int KT, bool PAIR, bool ROWS2>
__global__ __launch_bounds__(256, 3) void conv3s2_kernel(const float* __restrict__ x, const float* __restrict__ Wg,
                                                      float* __restrict__ y, Epi ep, int C, int H, int W, int Ho, int Wo, int K,
                                                      int blocks_x, int blocks_y, int kblocks, int kb_base) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bid = blockIdx.x;
    const int kb = kb_base + bid % kblocks; bid /= kblocks;
    const int bx = bid % blocks_x; bid /= blocks_x;
    const int by = bid % blocks_y;
    const int n = bid / blocks_y;
    const int py = ROWS2 ? 2 * (by * 4 + wave) : by * 4 + wave;  // this wave's (first) output row
    if (py >= Ho) return;
    const int kk = lane >> 5, nn = lane & 31;
    const int px0 = bx * (ROWS2 ? 32 : 64) + nn;                 // this lane's output pixels: px0 and px0 + 32 of row py, or px0 of rows py and py + 1
    const int nchunks = C / CB;
    const int plane = H * W * 4;                                 // bytes of one input plane

    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (int64_t)n * C * H * W), 0, C * plane, 0x00020000);
    const int kb_floats = 3 * nchunks * 2 * UNIT;
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wg + (int64_t)kb * kb_floats), 0, kb_floats * 4, 0x00020000);
    // lane part of the byte offsets (the lane's channel parity plane plus the column), or OOB for padding:
    // PAIR: [0] = column 2px-1, [1] = columns 2px, 2px+1 together; otherwise one per filter column
    int coff[3][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
            const int col = 2 * (px0 + (ROWS2 ? 0 : 32 * t)) + sx - 1;
            coff[sx][t] = (col >= 0 && col < W) ? kk * plane + col * 4 : OOB;
        }
    const int w_off = lane * 8;

    f32x16 acc[KT][2];
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][t][e] = 0.0f;

    const int iy = 2 * py - 1;                                   // input row of filter row 0
    const int r_lo = ROWS2 ? 0 : (iy < 0 ? 1 : 0), r_hi = ROWS2 ? 2 : (iy + 2 >= H ? H - 1 - iy : 2);      // filter rows to visit
    const int per_row = 2 * nchunks;                             // units of one filter row
    const int total = (r_hi - r_lo + 1) * per_row;

    // Three operand buffers in rotation: the loads of unit i+3 are issued right after the multiplications
    // of unit i.  The fences keep that issue order, so each wait covers one buffer's loads.  The loop
    // body has no branch (the waits stay exact and the accumulators stay in place): the number of units
    // is rounded up to a multiple of three, and a unit past the end reads its weights from past the
    // buffer's extent, i.e. multiplies by zero.
    float b[3][3][2][2];                                         // [buffer][filter column][k-step j][pixel tile]
    u32x2 a[3][3][KT];                                           // [buffer][filter column][k-tile], the two k-steps each
    int issued = 0;                                              // units whose loads have been issued
    auto load = [&](auto btag) {
        constexpr int lb = decltype(btag)::value;
        const int u = min(issued, total - 1);                    // past the end: the last unit's input again
        const int ri = (int)(u >= per_row) + (int)(u >= 2 * per_row);      // branch-free on purpose (see above)
        const int rest = u - ri * per_row, lr = r_lo + ri;       // rest = 2 * chunk + h
        // wave-uniform, said explicitly so that no load is wrapped in a loop over lanes
        const int soff = __builtin_amdgcn_readfirstlane(rest * 4 * plane + (iy + lr) * W * 4);
        const int woff = __builtin_amdgcn_readfirstlane((lr * per_row + rest) * (UNIT * 4));
        const int wv = issued < total ? w_off : OOB;
#pragma unroll
        for (int sx = 0; sx < 3; ++sx)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
                a[lb][sx][kt] = __builtin_amdgcn_raw_buffer_load_b64(wrs, wv + (sx * 2 + kt) * 512, woff, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // ROWS2: the second tile is the next output row (two input rows further down); a filter row
            // outside the image, or an output row past the last one, reads zeros
            const int in_row = iy + lr + 2 * t;
            const bool row_ok = !ROWS2 || (in_row >= 0 && in_row < H && py + t < Ho);
            const int st = ROWS2 ? soff + t * 2 * W * 4 : soff;
            const int c0 = row_ok ? coff[0][t] : OOB, c1 = row_ok ? coff[1][t] : OOB, c2 = row_ok ? coff[2][t] : OOB;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b[lb][0][j][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, c0, st + 2 * j * plane, 0));
                if (PAIR) {
                    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, c1, st + 2 * j * plane, 0);
                    b[lb][1][j][t] = __uint_as_float(v[0]);
                    b[lb][2][j][t] = __uint_as_float(v[1]);
                } else {
                    b[lb][1][j][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, c1, st + 2 * j * plane, 0));
                    b[lb][2][j][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, c2, st + 2 * j * plane, 0));
                }
            }
        }
        ++issued;
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](auto btag) {
        constexpr int lb = decltype(btag)::value;
#pragma unroll
        for (int sx = 0; sx < 3; ++sx)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
                        acc[kt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[lb][sx][kt][j]), b[lb][sx][j][t],
                                                                          acc[kt][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using B2 = std::integral_constant<int, 2>;
    load(B0{}); load(B1{}); load(B2{});
    // The compiler interleaves these first loads and would then make every pass of the loop wait as if
    // they were still in that order.  Naming their registers here makes it wait for them once, before
    // the loop; with nothing outstanding at the loop's entry the waits inside follow the loop's own order.
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) asm volatile("" : "+v"(a[q][sx][kt]));
            asm volatile("" : "+v"(b[q][sx][0][0]), "+v"(b[q][sx][0][1]), "+v"(b[q][sx][1][0]), "+v"(b[q][sx][1][1]));
        }
    for (int i = 0; i < total; i += 3) {
        step(B0{}); load(B0{});
        step(B1{}); load(B1{});
        step(B2{}); load(B2{});
    }

    store_tiles<KT, ROWS2>(acc, y + (int64_t)n * K * Ho * Wo, ep, kb, kk, K, Ho, Wo, py, px0);
}

// The 1x1 stride-2 convolution of the projection shortcuts: the same wave tile with a single tap (the
// centre one) and the reduction over chunks of 8 input channels; the weights are in
// vo_conv1x1_filter's order, Wg[kb][chunk][kt][lane][ks] with c = chunk*8 + 2*ks + (lane >> 5).
template <int KT>
__global__ __launch_bounds__(256, 3) void conv1s2_kernel(const float* __restrict__ x, const float* __restrict__ Wg,
                                                      float* __restrict__ y, Epi ep, int C, int H, int W, int Ho, int Wo, int K,
                                                      int blocks_x, int blocks_y, int kblocks, int kb_base) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bid = blockIdx.x;
    const int kb = kb_base + bid % kblocks; bid /= kblocks;
    const int bx = bid % blocks_x; bid /= blocks_x;
    const int by = bid % blocks_y;
    const int n = bid / blocks_y;
    const int py = by * 4 + wave;
    if (py >= Ho) return;
    const int kk = lane >> 5, nn = lane & 31;
    const int px0 = bx * 64 + nn;
    const int nchunks = C / CB;
    const int plane = H * W * 4;

    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (int64_t)n * C * H * W), 0, C * plane, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wg + (int64_t)kb * nchunks * 512), 0, nchunks * 512 * 4, 0x00020000);
    int coff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = 2 * (px0 + 32 * t);
        coff[t] = col < W ? kk * plane + col * 4 : OOB;
    }
    const int w_off = lane * 16;
    const int row = 2 * py * W * 4;                              // the sampled input row

    f32x16 acc[KT][2];
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][t][e] = 0.0f;

    float bcur[4][2], bnxt[4][2];
    u32x4 acur[KT], anxt[KT];
    auto load = [&](int chunk) {
        chunk = chunk < nchunks ? chunk : nchunks - 1;           // past the end: the last chunk again, never used
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) anxt[kt] = __builtin_amdgcn_raw_buffer_load_b128(wrs, w_off + kt * 1024, chunk * 2048, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                bnxt[ks][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, coff[t], (chunk * CB + 2 * ks) * plane + row, 0));
    };
    load(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acur[kt] = anxt[kt];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { bcur[ks][0] = bnxt[ks][0]; bcur[ks][1] = bnxt[ks][1]; }
        load(chunk + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    acc[kt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(acur[kt][ks]), bcur[ks][t], acc[kt][t], 0, 0, 0);
    }

    store_tiles<KT>(acc, y + (int64_t)n * K * Ho * Wo, ep, kb, kk, K, Ho, Wo, py, px0);
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
    // two rows of 32 pixels per wave where 64-pixel segments would leave the last one at most half full
    const bool rows2 = !ONE_TAP && (Wo + 31) / 32 * 32 < (Wo + 63) / 64 * 64;
    const int bx = rows2 ? (Wo + 31) / 32 : (Wo + 63) / 64, by = rows2 ? (Ho + 7) / 8 : (Ho + 3) / 4, kblocks = (K + KB - 1) / KB;
    const int64_t blocks = (int64_t)bx * by * N;
    if (blocks * kblocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "%s: grid too large", name);
    const Epi ep{scale, shift, relu};
    hipStream_t st = vo_stream(stream);
    const int rem = K % KB;
    const int full = (rem >= 1 && rem <= 32) ? kblocks - 1 : kblocks;         // a last block of <= 32 channels: one k-tile
    const dim3 gfull((unsigned)(blocks * (full > 0 ? full : 1))), ghalf((unsigned)blocks), nt(256);
    const bool pair = W % 2 == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0;
#define VO_S2_LAUNCH(KERNEL)                                                                                          \
    do {                                                                                                              \
        if (full > 0) KERNEL(2)<<<gfull, nt, 0, st>>>(x, arranged, y, ep, C, H, W, Ho, Wo, K, bx, by, full, 0);       \
        if (full < kblocks) KERNEL(1)<<<ghalf, nt, 0, st>>>(x, arranged, y, ep, C, H, W, Ho, Wo, K, bx, by, 1, full); \
    } while (0)
#define VO_S2_ONE(KT) conv1s2_kernel<KT>
#define VO_S2_PAIR(KT) conv3s2_kernel<KT, true, false>
#define VO_S2_SINGLE(KT) conv3s2_kernel<KT, false, false>
#define VO_S2_PAIR2(KT) conv3s2_kernel<KT, true, true>
#define VO_S2_SINGLE2(KT) conv3s2_kernel<KT, false, true>
    if (ONE_TAP) VO_S2_LAUNCH(VO_S2_ONE);
    else if (rows2 && pair) VO_S2_LAUNCH(VO_S2_PAIR2);
    else if (rows2) VO_S2_LAUNCH(VO_S2_SINGLE2);
    else if (pair) VO_S2_LAUNCH(VO_S2_PAIR);
    else VO_S2_LAUNCH(VO_S2_SINGLE);
#undef VO_S2_LAUNCH
#undef VO_S2_ONE
#undef VO_S2_PAIR
#undef VO_S2_SINGLE
#undef VO_S2_PAIR2
#undef VO_S2_SINGLE2
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
