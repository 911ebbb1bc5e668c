// Float32 convolutions (NCHW in and out) with every float32 product formed from bf16 pieces on the
// bf16 matrix cores:
//
//     x = xh + xl (+ 2^-16 x),  w = wh + wl (+ 2^-16 w),   xh = bf16(x), xl = bf16(x - xh)
//     x * w  ~  xl*wh + xh*wl + xh*wh        (three v_mfma_f32_32x32x16_bf16, float32 accumulation)
//
// The term left out (xl*wl) and the rounding of the low parts are 2^-16 of a product, i.e. each
// product keeps 16 significant bits (TF32, the tensor-core default of the reference's framework
// for convolutions, keeps 10); on the networks' layers the largest error is 4-6e-6 of the largest
// output, below the float32 Winograd F(4x4,3x3) kernel's 1e-5.  The float32 matrix instruction
// runs at the vector rate (1/16 of the bf16 rate), so three bf16 instructions per product are
// still five times its throughput, and the direct form needs no transforms.
//
// One kernel template, several layer types: 3x3 stride 1 (padding 1), 1x5 / 5x1 stride 1 (padding
// (0,2) / (2,0), the separable GRU's gates), 3x3 stride 2 (padding 1, the encoders' downsampling),
// 1x1 stride 2 (their projection shortcuts) and 1x1 stride 1 with many channels; and the two
// convolutions of a GRU step with the gate arithmetic in their stores (GruEp).
//
// Implicit GEMM: out[k, pixel] = sum over (tap, c) of W[k, c, tap] * x[c, stride*pixel + tap].
//   workgroup = 4 waves: 64 (or 32) output channels x (16 rows x 32 columns | 32 rows x 16 columns) of one image
//   wave w    = a quarter of the rows: MT x NTW accumulator tiles of 32 channels x 32 pixels (one row x 32
//               columns, or two rows x 16 columns)
//   input channels stream in chunks of 16 (one k-step): the patch of a chunk is split into high and
//   low parts while it is staged and lies in LDS as [part][k-half][row][column][8 channels] (a lane's
//   B fragment of any tap is one 16-byte read, lanes on consecutive 16-byte slots; with stride 2 the
//   even and the odd columns of a row are stored apart so that this still holds);
//   weights are pre-split and pre-arranged in operand order; a chunk's taps x 4 KiB are copied to LDS
//   as they are.  The multiply phase reads only LDS; two workgroups share a CU so that one stages
//   while the other multiplies.
#pragma once
#include "vo_common.h"
#include "vo_pieces.h"

#include <type_traits>

namespace {


constexpr int CK = 16;                        // input channels per chunk (the k of one matrix instruction)
constexpr int KB = 64;                        // output channels per workgroup
constexpr int NT = 256;                       // threads per workgroup
constexpr int A_TAP = 2 * 2 * 64 * 16;        // bytes of weights per (chunk, tap): [part][m-tile][lane][16]

// Filter R x S, stride ST; NTW = pixel tiles per wave; a pixel tile (the 32 columns of one matrix tile) is
// one output row x 32 columns (TW = 32) or two rows x 16 columns (TW = 16, for maps whose width fills
// 16-column blocks better: 80 columns are 5 x 16 but 3 x 32).  The workgroup's tile is TH rows x TW columns.
template <int R_, int S_, int ST_, int NTW_, int TW_ = 32>
struct Geo {
    static constexpr int R = R_, S = S_, ST = ST_, NTW = NTW_, TW = TW_;
    static constexpr int TAPS = R * S;
    static constexpr int PH = 32 / TW;                                        // rows of a pixel tile
    static constexpr int TH = 4 * NTW * PH;
    static_assert(TW == 32 || (TW == 16 && ST == 1), "two-row pixel tiles are written for stride 1");
    static constexpr int MAX_NORM_C = TW == 32 ? 256 : 128;                   // normalise-on-load table (what LDS leaves room for)
    static constexpr int PAD_H = R / 2, PAD_W = S / 2;
    static constexpr int SLACK = PAD_W & 1;                                   // the patch starts at an even input column
    static constexpr int RS = R == 1 ? ST : 1;                                // input rows between patch rows (a one-row filter skips the rows a stride passes over)
    static constexpr int PR = ((TH - 1) * ST + R - 1) / RS + 1;               // patch rows
    static constexpr int PC = ((TW - 1) * ST + S + SLACK + 1) / 2 * 2;        // patch columns (even)
    static constexpr int PLANE = PR * PC * 16;                                // bytes of one (part, k-half) plane
    static constexpr int UNITS = 2 * PR * (PC / 2);                           // staging units: (k-half, row, column pair)
    static constexpr int ROUNDS = (UNITS + NT - 1) / NT;
    static constexpr int A_CHUNK = TAPS * A_TAP;                              // bytes of weights per chunk
    static constexpr int A_PIECES = (A_CHUNK / 16 + NT - 1) / NT;             // 16-byte pieces per thread and chunk
    static constexpr int STEPS = TAPS * NTW;                                  // (tap, pixel tile) steps per chunk
    // position of patch column c inside a row of the LDS image
    static constexpr int col_pos(int c) { return ST == 1 ? c : (c & 1) * (PC / 2) + (c >> 1); }
};

// Weights (K,C,taps) float32 -> high and low parts in the register order of the A operand:
//   Ag[kb][chunk][tap][part][mt][lane][j]:  k = kb*64 + mt*32 + (lane & 31),  c = chunk*16 + 8*(lane >> 5) + j
template <typename P>
__global__ __launch_bounds__(256) void filter_split_kernel(const float* __restrict__ g, unsigned short* __restrict__ Ag,
                                                           int K, int C, int Kpad, int taps, float wscale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Kpad * C) return;
    const int k = (int)(idx / C), c = (int)(idx % C);
    const int kb = k >> 6, mt = (k >> 5) & 1, kl = k & 31, chunk = c >> 4, kg = (c >> 3) & 1, j = c & 7;
    const int nchunks = C / CK;
    for (int tap = 0; tap < taps; ++tap) {
        const float w = k < K ? g[((int64_t)k * C + c) * taps + tap] : 0.0f;
        unsigned hi, lo;
        P::cut_weight(w, wscale, hi, lo);
        const int64_t base = ((((int64_t)kb * nchunks + chunk) * taps + tap) * 2) * 2;       // [part][mt]
        const int lane = kg * 32 + kl;
        Ag[((base + 0 * 2 + mt) * 64 + lane) * 8 + j] = (unsigned short)hi;
        Ag[((base + 1 * 2 + mt) * 64 + lane) * 8 + j] = (unsigned short)lo;
    }
}

// nt: the 16-byte store path marks its stores nontemporal (set by the host for outputs far larger than the caches:
// the tensor streams out and leaves the patch rows and weights in L2; +1.4 % on the full-resolution 64 -> 64 layers)
// unscale: the factor that takes the accumulators back to the weights' own scale (PiecesF16; 1 otherwise)
struct Epilogue { const float* scale; const float* shift; const float* skip; int relu; int64_t y_batch_stride; double* stats; int pixel_major; int nt = 0; float unscale = 1.0f; };
struct InputNorm { const double* stats; float eps; };

// Gate arithmetic of a convolutional GRU in the convolution's store (hidden width C, a multiple of 64):
//   mode 1, the merged z|r convolution (2C output channels, shift = the two biases):
//     channels < C :  z = sigmoid(conv + bias + a)          -> y   (N,C,Ho,Wo)
//     channels >= C:  r*h = sigmoid(conv + bias + b) * h    -> out2 (N,C,Ho,Wo)
//   mode 2, the q convolution (C output channels), whose first C INPUT channels come from `head` (= r*h)
//   instead of x:   h' = (1 - z)*h + z*tanh(conv + bias + a) -> y (batch stride y_batch_stride) and out2 (copy)
// a, b: the precomputed context terms (or null); h, z as named; *_bs = batch strides in elements.
struct GruEp {
    int mode, C;
    const float* head;
    const float* a; int64_t a_bs;
    const float* b; int64_t b_bs;
    const float* h; int64_t h_bs;
    const float* z;
    float* out2; int64_t out2_bs;
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// MT: 32-channel tiles per workgroup (2: 64 output channels; 1: a last block of at most 32).
// H, W: input size; Ho, Wo: output size.
// SHORT (3x3 stride 2 only): the residual block's projection shortcut, a 1x1 stride-2 convolution of the same input
// with as many output channels, is computed beside the 3x3 one from the same staged patch: its sample x[2y][2x]
// is the 3x3 filter's centre tap, so it costs one more tap of matrix instructions per chunk instead of a kernel
// that stages the input a second time.  `sc` holds its weights (vo_conv1x1_bf16x3_filter), output and epilogue.
struct Shortcut { const unsigned short* Ag; float* y; Epilogue ep; };


// VO_CONV_TIMELINE (tools/conv_timeline.hip only): every workgroup leaves five stamps of the chip-wide 100 MHz counter
// -- entry, first multiplication, end of the main loop, last store issued, stores complete -- and where it ran.
#ifdef VO_CONV_TIMELINE                      // the including file defines __device__ unsigned long long vo_conv_timeline[] first
struct ConvTimeline {
    unsigned long long t[5], c[4] = {0, 0, 0, 0};                          // c: core-clock stamps inside the second chunk
    __device__ __forceinline__ void stamp(int i) { t[i] = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void cycles(int i) { c[i] = __builtin_readcyclecounter(); }
    __device__ ~ConvTimeline() {
        stamp(3);
        __builtin_amdgcn_s_waitcnt(0);
        stamp(4);
        if (threadIdx.x == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* o = vo_conv_timeline + (size_t)blockIdx.x * 10;
            for (int i = 0; i < 5; ++i) o[i] = t[i];
            o[5] = ((unsigned long long)xcc << 32) | hw;
            for (int i = 0; i < 4; ++i) o[6 + i] = c[i];
        }
    }
};
#define VO_TL_DECL ConvTimeline tl_; tl_.stamp(0)
#define VO_TL(i) tl_.stamp(i)
#define VO_TLC(i) tl_.cycles(i)
#else
#define VO_TL_DECL
#define VO_TL(i)
#define VO_TLC(i)
#endif

template <typename P, typename G, bool NORM_IN, int MT, bool GRU = false, bool SHORT = false>
__global__ __launch_bounds__(NT, 2) void conv_bf16x3_kernel(const float* __restrict__ x, const unsigned short* __restrict__ Ag,
                                                            float* __restrict__ y, Epilogue ep, InputNorm in, int C, int H,
                                                            int W, int Ho, int Wo, int K, int blocks_x, int blocks_y,
                                                            int kblocks, int kb_base, GruEp gru, Shortcut sc) {
    static_assert(!SHORT || (G::ST == 2 && G::TAPS == 9 && !NORM_IN && !GRU), "the fused shortcut belongs to the 3x3 stride-2 layer");
    constexpr int PLANE = G::PLANE, PC = G::PC, PR = G::PR, UNITS = G::UNITS, ROUNDS = G::ROUNDS, NTW = G::NTW;
    constexpr int A_CHUNK = G::A_CHUNK, A_PIECES = G::A_PIECES, TAPS = G::TAPS, STEPS = G::STEPS;
    // patch | weights of the chunk | (NORM_IN) per-channel scale and shift.  The reduction buffer of the
    // statistics epilogue re-uses the patch area after the last chunk.
    __shared__ __align__(16) unsigned char lds[4 * PLANE + A_CHUNK + (NORM_IN ? G::MAX_NORM_C * 8 : 0) + (SHORT ? A_TAP : 0)];
    unsigned char* a2_lds = lds + 4 * PLANE + A_CHUNK;                       // SHORT: the shortcut's weights of the chunk
    float2* nrm = reinterpret_cast<float2*>(lds + 4 * PLANE + A_CHUNK);
    double* red = reinterpret_cast<double*>(lds);                            // [2][4 waves][64]
    static_assert(4 * PLANE >= 4 * 4096, "the epilogue's transposition slots and the reduction buffer do not fit the patch area");

    VO_TL_DECL;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid;
    {   // one contiguous range of the block list per XCD (workgroups are dealt round robin over 8)
        const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, q = nwg >> 3, r = nwg & 7u;
        bid = (int)(((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3));
    }
    const int kb = kb_base + bid % kblocks; bid /= kblocks;
    const int bx = bid % blocks_x; bid /= blocks_x;
    const int by = bid % blocks_y;
    const int n = bid / blocks_y;
    const int nchunks = C / CK;
    const int y0 = by * G::TH, x0 = bx * G::TW;                              // output pixel origin
    const int HW4 = H * W * 4, HWo4 = Ho * Wo * 4;

    const float* xn = x + (int64_t)n * C * H * W;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, C * HW4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(Ag) + (int64_t)kb * nchunks * (A_CHUNK / 2), 0, nchunks * A_CHUNK, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(SHORT ? sc.Ag + (int64_t)kb * nchunks * (A_TAP / 2) : Ag), 0, SHORT ? nchunks * A_TAP : 0, 0x00020000);
    // GRU mode 2: the first gru.C input channels are read from `head` (dense, gru.C planes per image)
    const int head_chunks = (GRU && gru.head) ? gru.C / CK : 0;
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(head_chunks ? gru.head + (int64_t)n * gru.C * H * W : xn), 0, head_chunks ? gru.C * HW4 : 0, 0x00020000);

    // staging units of this thread: (k-half, patch row, column pair); a unit outside the image carries an
    // offset past the extent, which the range check of a buffer load turns into zeros (the padding)
    int u_off[ROUNDS], u_lds[ROUNDS];
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        const int u = tid + i * NT;
        const int kg = u / (PR * (PC / 2)), rem = u - kg * (PR * (PC / 2)), row = rem / (PC / 2), pr = rem - row * (PC / 2);
        const int gy = y0 * G::ST - G::PAD_H + row * G::RS, gx = x0 * G::ST - G::PAD_W - G::SLACK + 2 * pr;
        const bool ok = u < UNITS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        u_off[i] = ok ? ((kg * 8 * H + gy) * W + gx) * 4 : (int)0x80000000;
        // stride 1: the pair's pixels lie side by side; stride 2: the even one in the row's first half, the odd one in its second
        u_lds[i] = kg * PLANE + (row * PC + (G::ST == 1 ? 2 * pr : pr)) * 16;                // + part * 2 * PLANE
    }
    constexpr int PX_STEP = G::ST == 1 ? 16 : (PC / 2) * 16;                 // bytes from a pair's first pixel to its second
    if (NORM_IN) {
        const double hw = (double)H * (double)W;
        for (int c = tid; c < C; c += NT) {
            const double* st = in.stats + ((int64_t)n * C + c) * 2;
            const double mean = st[0] / hw;
            double var = st[1] / hw - mean * mean;
            if (var < 0.0) var = 0.0;
            const float r = (float)(1.0 / sqrt(var + (double)in.eps));
            nrm[c] = make_float2(r, -(float)mean * r);
        }
    }

    u32x2 rv[ROUNDS][8];
    auto load_raw = [&](int chunk) {
        chunk = chunk < nchunks ? chunk : nchunks - 1;                       // past the end: reloaded, never used
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            if (i * NT + NT > UNITS && tid + i * NT >= UNITS) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                rv[i][j] = __builtin_amdgcn_raw_buffer_load_b64((GRU && chunk < head_chunks) ? hrs : xrs, u_off[i], (chunk * CK + j) * HW4, 0);
        }
    };
    auto store_raw = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            if (i * NT + NT > UNITS && tid + i * NT >= UNITS) continue;
            float v[8][2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j][0] = __uint_as_float(rv[i][j][0]);
                v[j][1] = __uint_as_float(rv[i][j][1]);
            }
            if (NORM_IN) {                                                   // the padding pads the NORMALISED tensor: it stays zero
                const int kg = (tid + i * NT) / (PR * (PC / 2));
                const bool ok = u_off[i] >= 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float2 ss = nrm[chunk * CK + kg * 8 + j];
                    v[j][0] = ok ? fmaxf(fmaf(v[j][0], ss.x, ss.y), 0.0f) : 0.0f;
                    v[j][1] = ok ? fmaxf(fmaf(v[j][1], ss.x, ss.y), 0.0f) : 0.0f;
                }
            }
            // the remainders are taken on the register pairs the loads delivered (a channel's two pixels): two-wide
            // subtractions with no register moves to pair their operands up
            u32x4 hi[2], lo[2];
            if constexpr (P::F16) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        unsigned h, l;
                        P::cut(v[2 * jj][px], v[2 * jj + 1][px], h, l);
                        hi[px][jj] = h;
                        lo[px][jj] = l;
                    }
            } else
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x2 a = {v[2 * jj][0], v[2 * jj][1]}, b = {v[2 * jj + 1][0], v[2 * jj + 1][1]};
                const unsigned h0 = pack_bf16(a[0], b[0]), h1 = pack_bf16(a[1], b[1]);
                const f32x2 ha = {__uint_as_float(h0 << 16), __uint_as_float(h1 << 16)};
                const f32x2 hb = {__uint_as_float(h0 & 0xffff0000u), __uint_as_float(h1 & 0xffff0000u)};
                const f32x2 la = a - ha, lb = b - hb;
                hi[0][jj] = h0;
                hi[1][jj] = h1;
                lo[0][jj] = pack_bf16(la[0], lb[0]);
                lo[1][jj] = pack_bf16(la[1], lb[1]);
            }
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                *reinterpret_cast<u32x4*>(lds + u_lds[i] + px * PX_STEP) = hi[px];
                *reinterpret_cast<u32x4*>(lds + 2 * PLANE + u_lds[i] + px * PX_STEP) = lo[px];
            }
        }
    };

    f32x16 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.0f;

    f32x16 acc2[SHORT ? MT : 1][SHORT ? NTW : 1];
    if (SHORT) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[mt][nt][e] = 0.0f;
    }

    // A fragment of (tap, part, mt): weights area + ((tap*2 + part)*2 + mt) KiB + 16*lane
    // B fragment of (tap r, s), pixel tile nt: patch row (NTW*wave + nt)*stride + r, patch column (lane & 31)*stride + s + slack
    const unsigned char* a_lds = lds + 4 * PLANE + lane * 16;
    const int prow = (lane & 31) / G::TW, pcol = (lane & 31) % G::TW;        // this lane's pixel inside a pixel tile
    const unsigned char* b_lds = lds + (lane >> 5) * PLANE + (((NTW * wave * G::PH + prow) * G::ST / G::RS) * PC + pcol) * 16;
    auto read_a = [&](int tap, int part, int mt) {
        return *reinterpret_cast<const u32x4*>(a_lds + ((tap * 2 + part) * 2 + mt) * 1024);
    };
    auto read_b = [&](int tap, int nt, int part) {
        const int r = tap / G::S, c = tap % G::S + G::SLACK;                 // lane n reads patch column n*stride + c
        const int pos = G::ST == 1 ? c : (c & 1) * (PC / 2) + (c >> 1);
        return *reinterpret_cast<const u32x4*>(b_lds + part * 2 * PLANE + (((nt * G::PH * G::ST + r) / G::RS) * PC + pos) * 16);
    };

    // a chunk's weights are a linear copy (they are stored in operand order); they are requested at the
    // end of the previous chunk's multiply phase, so their latency passes under the barrier and the
    // conversion of the patch
    u32x4 aw[A_PIECES], aw2;
    static_assert(A_TAP == NT * 16, "the shortcut's weights of a chunk are one 16-byte piece per thread");
    auto load_aw = [&](int chunk) {
        chunk = chunk < nchunks ? chunk : nchunks - 1;
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i)
            if ((i + 1) * NT * 16 <= A_CHUNK || (tid + i * NT) * 16 < A_CHUNK)
                aw[i] = __builtin_amdgcn_raw_buffer_load_b128(ars, (tid + i * NT) * 16, chunk * A_CHUNK, 0);
        if (SHORT) aw2 = __builtin_amdgcn_raw_buffer_load_b128(ars2, tid * 16, chunk * A_TAP, 0);
    };
    load_raw(0);
    load_aw(0);
    if (NORM_IN) __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk == 1) { VO_TLC(0); }
        store_raw(chunk);
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i)
            if ((i + 1) * NT * 16 <= A_CHUNK || (tid + i * NT) * 16 < A_CHUNK)
                *reinterpret_cast<u32x4*>(lds + 4 * PLANE + (tid + i * NT) * 16) = aw[i];
        if (SHORT) *reinterpret_cast<u32x4*>(a2_lds + tid * 16) = aw2;
        __syncthreads();
        if (chunk == 0) { VO_TL(1); }
        if (chunk == 1) { VO_TLC(1); }
        load_raw(chunk + 1);                                                 // in flight during the whole multiply phase

        // STEPS steps of (tap, pixel tile): 3*MT matrix instructions each, operands double-buffered in
        // registers: a step issues the reads of the next one (and its share of the next tap's weights)
        u32x4 fa[2][2][MT], fb[2][2];
#pragma unroll
        for (int part = 0; part < 2; ++part) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[0][part][mt] = read_a(0, part, mt);
            fb[0][part] = read_b(0, 0, part);
        }
        constexpr int A_PER_STEP = (2 * MT + NTW - 1) / NTW;                 // weight fragments fetched per step
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int tap = st / NTW, nt = st % NTW, ca = tap & 1, cb = st & 1;
            if (st + 1 < STEPS) {
                fb[cb ^ 1][0] = read_b((st + 1) / NTW, (st + 1) % NTW, 0);
                fb[cb ^ 1][1] = read_b((st + 1) / NTW, (st + 1) % NTW, 1);
            }
            if (tap + 1 < TAPS) {
#pragma unroll
                for (int f = nt * A_PER_STEP; f < (nt + 1) * A_PER_STEP && f < 2 * MT; ++f)
                    fa[ca ^ 1][f / MT][f % MT] = read_a(tap + 1, f / MT, f % MT);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                // the small terms first, so that they are not rounded away against the large one
                acc[mt][nt] = P::mfma(fa[ca][1][mt], fb[cb][0], acc[mt][nt]);
                acc[mt][nt] = P::mfma(P::for_low_input(fa[ca][0][mt]), fb[cb][1], acc[mt][nt]);
                acc[mt][nt] = P::mfma(fa[ca][0][mt], fb[cb][0], acc[mt][nt]);
            }
            if (SHORT && tap == 4) {
                // the centre tap's B fragments are the shortcut's input samples
                u32x4 f2[2][MT];
#pragma unroll
                for (int part = 0; part < 2; ++part)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) f2[part][mt] = *reinterpret_cast<const u32x4*>(a2_lds + lane * 16 + (part * 2 + mt) * 1024);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc2[mt][nt] = P::mfma(f2[1][mt], fb[cb][0], acc2[mt][nt]);
                    acc2[mt][nt] = P::mfma(P::for_low_input(f2[0][mt]), fb[cb][1], acc2[mt][nt]);
                    acc2[mt][nt] = P::mfma(f2[0][mt], fb[cb][0], acc2[mt][nt]);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2 + A_PER_STEP + 2 * MT, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6 * MT, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                // the step's reads first (they feed the NEXT step), then its matrix instructions
                __builtin_amdgcn_sched_group_barrier(0x100, 2 + A_PER_STEP, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * MT, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (chunk == 1) { VO_TLC(2); }
        load_aw(chunk + 1);
        __syncthreads();                                                     // everyone has read the patch and the weights
        if (chunk == 1) { VO_TLC(3); }
    }
    VO_TL(2);

    if constexpr (P::F16) {
        // back to the weights' own scale (a power of two: exact)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                acc[mt][nt] *= ep.unscale;
                if constexpr (SHORT) acc2[mt][nt] *= sc.ep.unscale;
            }
    }

    // Epilogue.  C/D layout of the 32x32 tile: pixel = lane & 31, channel row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5).
    float* yn = y + (int64_t)n * ep.y_batch_stride;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yn, 0, K * HWo4, 0x00020000);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(ep.skip ? ep.skip + (int64_t)n * K * Ho * Wo : x), 0, ep.skip ? K * HWo4 : 0, 0x00020000);
    const int col = x0 + pcol, lh = lane >> 5;
    const int row0 = y0 + NTW * G::PH * wave + prow;                         // this lane's row of pixel tile 0

    if (ep.pixel_major) {
        // y as (N, Ho*Wo, K): a lane's four consecutive channel rows of one pixel are one 16-byte store (K % 4 == 0);
        // the layout the correlation kernels read their feature maps in
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(y + (int64_t)n * Ho * Wo * K, 0, Ho * Wo * K * 4, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int k = kb * KB + mt * 32 + 8 * g + 4 * lh;
                float sc[4], sh[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sc[i] = 1.0f; sh[i] = 0.0f;
                    if (ep.shift && k < K) { sc[i] = ep.scale ? ep.scale[k + i] : 1.0f; sh[i] = ep.shift[k + i]; }
                }
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int row = row0 + nt * G::PH;
                    const bool ok = row < Ho && col < Wo && k < K;
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[mt][nt][4 * g + i];
                        if (ep.shift) v = fmaf(v, sc[i], sh[i]);
                        if (ep.relu) v = fmaxf(v, 0.0f);
                        o[i] = v;
                    }
                    const int base = ok ? (row * Wo + col) * K * 4 : (int)0x80000000;
                    if (ep.pixel_major == 2) {
                        // the row as bf16 pieces in blocks of 32 channels, [32 high parts | 32 low parts] each:
                        // what the correlation kernel for rows split beforehand reads
                        const unsigned h0 = pack_bf16(o[0], o[1]), h1 = pack_bf16(o[2], o[3]);
                        const unsigned l0 = pack_bf16(o[0] - __uint_as_float(h0 << 16), o[1] - __uint_as_float(h0 & 0xffff0000u));
                        const unsigned l1 = pack_bf16(o[2] - __uint_as_float(h1 << 16), o[3] - __uint_as_float(h1 & 0xffff0000u));
                        const int off = base + (k >> 5) * 128 + (k & 31) * 2;
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{h0, h1}, prs, off, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{l0, l1}, prs, off + 64, 0, 0);
                    } else {
                        const u32x4 ov = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
                        __builtin_amdgcn_raw_buffer_store_b128(ov, prs, base + k * 4, 0, 0);
                    }
                }
            }
        return;
    }

    // 16-byte store path (below): needs rows that keep 16-byte alignment; otherwise the accumulator layout is stored as it is
    const bool wide = (Wo & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (ep.y_batch_stride & 3) == 0 &&
                      (reinterpret_cast<uintptr_t>(ep.skip) & 15) == 0;
    if (ep.stats && !wide) {
        // plane sums of the raw output: per lane over its four rows, then over the 32 pixel lanes by a
        // halving exchange (each step hands half of the lane's channel rows to the partner lane and adds
        // the partner's half to its own), then over the four waves through LDS: one pair of float64
        // atomics per channel and workgroup
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            // float32 only over the lane's own few pixels; everything across lanes in float64, so that the
            // variance E[x^2] - mean^2 formed from these sums survives planes whose mean dwarfs their spread
            double s1[16], s2[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float a1 = 0.0f, a2 = 0.0f;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const bool ok = row0 + nt * G::PH < Ho && col < Wo;
                    const float v = ok ? acc[mt][nt][e] : 0.0f;
                    a1 += v; a2 = fmaf(v, v, a2);
                }
                s1[e] = (double)a1; s2[e] = (double)a2;
            }
            // 16 values on 32 lanes -> lane l (of 32) ends with the total of value (l & 15)
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                const int m = 16 >> step;                                    // lane mask 16, 8, 4, 2; values kept: 8, 4, 2, 1
                const int keep = 8 >> step;
                const bool upper = (lane & m) != 0;
#pragma unroll
                for (int i = 0; i < keep; ++i) {
                    const double mine1 = upper ? s1[i + keep] : s1[i], send1 = upper ? s1[i] : s1[i + keep];
                    const double mine2 = upper ? s2[i + keep] : s2[i], send2 = upper ? s2[i] : s2[i + keep];
                    s1[i] = mine1 + __shfl_xor(send1, m, VO_WAVE);
                    s2[i] = mine2 + __shfl_xor(send2, m, VO_WAVE);
                }
            }
            s1[0] += __shfl_xor(s1[0], 1, VO_WAVE);
            s2[0] += __shfl_xor(s2[0], 1, VO_WAVE);
            // lane l now holds value index v(l) = bits (l>>4 &1)*8 + (l>>3 &1)*4 + (l>>2 &1)*2 + (l>>1 &1), both lanes of a pair
            if ((lane & 1) == 0) {
                const int l5 = lane & 31;
                const int e = ((l5 >> 4) & 1) * 8 + ((l5 >> 3) & 1) * 4 + ((l5 >> 2) & 1) * 2 + ((l5 >> 1) & 1);
                const int krow = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                red[(0 * 4 + wave) * 64 + krow] = s1[0];
                red[(1 * 4 + wave) * 64 + krow] = s2[0];
            }
        }
        __syncthreads();
        if (tid < 2 * 32 * MT) {
            const int which = tid / (32 * MT), krow = tid - which * (32 * MT);
            const double* rp = red + which * 4 * 64 + krow;
            const double tot = (rp[0] + rp[64]) + (rp[128] + rp[192]);
            const int k = kb * KB + krow;
            if (k < K) atomicAdd(ep.stats + ((int64_t)n * K + k) * 2 + which, tot);
        }
    }


    if constexpr (GRU) {
        // Gate arithmetic in the store (16-byte path only: the host checks Wo % 4 == 0 and the alignments).
        float* slot = reinterpret_cast<float*>(lds) + wave * 1024;
        const int q = lane >> 3, px4 = (lane & 7) * 4;
        const int colw = x0 + px4 % G::TW, roww = y0 + NTW * G::PH * wave + px4 / G::TW;
        const int CH = gru.C;
        const bool second = gru.mode == 1 && kb * KB >= CH;                  // this workgroup's channels are r channels
        const int cbase = kb * KB - (second ? CH : 0);
        const int plane4 = CH * HWo4;                                        // bytes of one image of a C-channel tensor
        const float* ctx = gru.mode == 1 ? (second ? gru.b : gru.a) : gru.a;
        const int64_t ctx_bs = gru.mode == 1 ? (second ? gru.b_bs : gru.a_bs) : gru.a_bs;
        const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ctx ? ctx + (int64_t)n * ctx_bs : x), 0, ctx ? plane4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t h2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gru.h + (int64_t)n * gru.h_bs), 0, plane4, 0x00020000);
        const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gru.z ? gru.z + (int64_t)n * CH * Ho * Wo : x), 0, gru.z ? plane4 : 0, 0x00020000);
        // mode 1: z -> y, r*h -> out2 (both dense);  mode 2: h' -> y (its batch stride) and -> out2 (copy, its batch stride)
        float* o1 = gru.mode == 1 ? (second ? gru.out2 + (int64_t)n * CH * Ho * Wo : y + (int64_t)n * CH * Ho * Wo) : y + (int64_t)n * ep.y_batch_stride;
        float* o2 = (gru.mode == 2 && gru.out2) ? gru.out2 + (int64_t)n * gru.out2_bs : nullptr;
        const __amdgpu_buffer_rsrc_t o1rs = __builtin_amdgcn_make_buffer_rsrc(o1, 0, plane4, 0x00020000);
        const __amdgpu_buffer_rsrc_t o2rs = __builtin_amdgcn_make_buffer_rsrc(o2 ? o2 : o1, 0, o2 ? plane4 : 0, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float sh[4];
            int koff[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kb * KB + mt * 32 + q + 8 * j;
                sh[j] = ep.shift ? ep.shift[k] : 0.0f;
                koff[j] = (cbase + mt * 32 + q + 8 * j) * HWo4;
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int row = roww + nt * G::PH;
                const bool pok = row < Ho && colw < Wo;
                const int poff = (row * Wo + colw) * 4;
                u32x4 cx[4], hh[4], zz[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int off = pok ? koff[j] + poff : (int)0x80000000;
                    cx[j] = __builtin_amdgcn_raw_buffer_load_b128(crs, off, 0, 0);              // zeros without context terms
                    if (gru.mode == 2 || second) hh[j] = __builtin_amdgcn_raw_buffer_load_b128(h2rs, off, 0, 0);
                    if (gru.mode == 2) zz[j] = __builtin_amdgcn_raw_buffer_load_b128(zrs, off, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) slot[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + (lane & 31)] = acc[mt][nt][e];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 t = *reinterpret_cast<const float4*>(slot + (q + 8 * j) * 32 + px4);
                    float v[4] = {t.x, t.y, t.z, t.w};
                    u32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float pre = (v[i] + __uint_as_float(cx[j][i])) + sh[j];
                        float r;
                        if (gru.mode == 1) {
                            r = sigmoid_f(pre);
                            if (second) r *= __uint_as_float(hh[j][i]);
                        } else {
                            const float zv = __uint_as_float(zz[j][i]);
                            r = (1.0f - zv) * __uint_as_float(hh[j][i]) + zv * tanhf(pre);
                        }
                        o[i] = __float_as_uint(r);
                    }
                    const int off = pok ? koff[j] + poff : (int)0x80000000;
                    __builtin_amdgcn_raw_buffer_store_b128(o, o1rs, off, 0, 0);
                    if (gru.mode == 2) __builtin_amdgcn_raw_buffer_store_b128(o, o2rs, off, 0, 0);  // dropped when there is no copy
                }
            }
        }
        return;
    }

    // Wide path: every tile passes through a 4 KiB LDS slot of its wave (32 channels x 32 pixels), so that a
    // lane ends up with four consecutive pixels of one channel: 16-byte stores (and residual loads), a
    // quarter of the store instructions, each wave-instruction covering eight full 128-byte segments.
    // Plane sums ride along: a lane adds up the pixels it stores (float32 over its 16 pixels of a channel),
    // the eight lanes that share a channel row combine in float64, the four waves through LDS.
    if (wide) {
      auto wide_store = [&](auto& acc, const Epilogue& ep, const __amdgpu_buffer_rsrc_t& yrs, const __amdgpu_buffer_rsrc_t& srs) {
        float* slot = reinterpret_cast<float*>(lds) + wave * 1024;
        double* redw = reinterpret_cast<double*>(lds + 4 * 4096);            // [2][4 waves][64], behind the four slots
        static_assert(4 * PLANE + A_CHUNK >= 4 * 4096 + 2 * 4 * 64 * 8, "no room for the reduction buffer behind the slots");
        const int q = lane >> 3, px4 = (lane & 7) * 4;                       // read-back role: channel rows q + 8j, pixels px4..px4+3
        const int colw = x0 + px4 % G::TW, roww = y0 + NTW * G::PH * wave + px4 / G::TW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float sc[4], sh[4];
            int koff[4];
            float w1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, w2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kb * KB + mt * 32 + q + 8 * j;
                const bool kok = k < K;
                sc[j] = 1.0f; sh[j] = 0.0f;
                if (ep.shift && kok) { sc[j] = ep.scale ? ep.scale[k] : 1.0f; sh[j] = ep.shift[k]; }
                koff[j] = kok ? k * HWo4 : (int)0x80000000;
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int row = roww + nt * G::PH;
                const bool pok = row < Ho && colw < Wo;
                const int poff = (row * Wo + colw) * 4;
                u32x4 sk[4];
                if (ep.skip) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        sk[j] = __builtin_amdgcn_raw_buffer_load_b128(srs, (pok && koff[j] >= 0) ? koff[j] + poff : (int)0x80000000, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) slot[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + (lane & 31)] = acc[mt][nt][e];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 t = *reinterpret_cast<const float4*>(slot + (q + 8 * j) * 32 + px4);
                    float v[4] = {t.x, t.y, t.z, t.w};
                    if (ep.stats && pok) {                                   // rows are whole groups of four pixels: all in or all out
                        w1[j] += (v[0] + v[1]) + (v[2] + v[3]);
                        w2[j] = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], w2[j]))));
                    }
                    u32x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (ep.shift) v[i] = fmaf(v[i], sc[j], sh[j]);
                        if (ep.relu) v[i] = fmaxf(v[i], 0.0f);
                        if (ep.skip) v[i] = fmaxf(v[i] + __uint_as_float(sk[j][i]), 0.0f);
                        o[i] = __float_as_uint(v[i]);
                    }
                    if (ep.nt) __builtin_amdgcn_raw_buffer_store_b128(o, yrs, (pok && koff[j] >= 0) ? koff[j] + poff : (int)0x80000000, 0, 2);
                    else __builtin_amdgcn_raw_buffer_store_b128(o, yrs, (pok && koff[j] >= 0) ? koff[j] + poff : (int)0x80000000, 0, 0);
                }
            }
            if (ep.stats) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double d1 = (double)w1[j], d2 = (double)w2[j];
#pragma unroll
                    for (int m = 1; m < 8; m <<= 1) { d1 += __shfl_xor(d1, m, VO_WAVE); d2 += __shfl_xor(d2, m, VO_WAVE); }
                    if ((lane & 7) == 0) {
                        redw[(0 * 4 + wave) * 64 + mt * 32 + q + 8 * j] = d1;
                        redw[(1 * 4 + wave) * 64 + mt * 32 + q + 8 * j] = d2;
                    }
                }
            }
        }
        if (ep.stats) {
            __syncthreads();
            if (tid < 2 * 32 * MT) {
                const int which = tid / (32 * MT), krow = tid - which * (32 * MT);
                const double* rp = redw + which * 4 * 64 + krow;
                const double tot = (rp[0] + rp[64]) + (rp[128] + rp[192]);
                const int k = kb * KB + krow;
                if (k < K) atomicAdd(ep.stats + ((int64_t)n * K + k) * 2 + which, tot);
            }
        }
      };
        wide_store(acc, ep, yrs, srs);
        if constexpr (SHORT) {
            // the shortcut's tile: same path, its own output, epilogue and sums (the slots and the reduction
            // buffer are free again once every wave is through the first store)
            __syncthreads();
            float* y2n = sc.y + (int64_t)n * sc.ep.y_batch_stride;
            const __amdgpu_buffer_rsrc_t yrs2 = __builtin_amdgcn_make_buffer_rsrc(y2n, 0, K * HWo4, 0x00020000);
            const __amdgpu_buffer_rsrc_t srs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0, 0x00020000);
            wide_store(acc2, sc.ep, yrs2, srs2);
        }
        return;
    }

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float sc[16], sh[16];
        int koff[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int k = kb * KB + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const bool kok = k < K;
            sc[e] = 1.0f; sh[e] = 0.0f;
            if (ep.shift && kok) { sc[e] = ep.scale ? ep.scale[k] : 1.0f; sh[e] = ep.shift[k]; }
            koff[e] = kok ? k * HWo4 : (int)0x80000000;
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int row = row0 + nt * G::PH;
            const bool pok = row < Ho && col < Wo;
            const int poff = (row * Wo + col) * 4;
            float sk[16];
            if (ep.skip) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    sk[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(srs, pok ? koff[e] + poff : (int)0x80000000, 0, 0));
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[mt][nt][e];
                if (ep.shift) v = fmaf(v, sc[e], sh[e]);
                if (ep.relu) v = fmaxf(v, 0.0f);
                if (ep.skip) v = fmaxf(v + sk[e], 0.0f);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, (pok && koff[e] >= 0) ? koff[e] + poff : (int)0x80000000, 0, 0);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// host side

// weight_exp: PiecesF16 only -- the weights are cut after a multiplication by 2^weight_exp (the caller picks it so
// that the largest |w| * 2^weight_exp lies in [2^13, 2^14) and hands the same value to the convolution).
template <typename P>
int split_filter(const char* name, const float* weight, void* arranged, int K, int C, int taps, int weight_exp, hipStream_t st) {
    if (!weight || !arranged) return vo_fail(VO_ERR_INVALID_ARG, "%s: null pointer", name);
    if (K <= 0 || C <= 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: bad sizes K=%d C=%d", name, K, C);
    if (C % CK != 0) return vo_fail(VO_ERR_UNSUPPORTED, "%s: C=%d is not a multiple of %d", name, C, CK);
    const int Kpad = (K + KB - 1) / KB * KB;
    const int64_t n = (int64_t)Kpad * C;
    if (weight_exp < -100 || weight_exp > 100) return vo_fail(VO_ERR_INVALID_ARG, "%s: weight_exp %d out of range", name, weight_exp);
    filter_split_kernel<P><<<dim3((unsigned)vo_ceil_div(n, 256)), dim3(256), 0, st>>>(weight, static_cast<unsigned short*>(arranged), K,
                                                                                    C, Kpad, taps, ldexpf(1.0f, P::F16 ? weight_exp : 0));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e));
    return VO_OK;
}

// an output of half a gigabyte or more passes through the caches only on its way out
inline int streams_out(int N, int K, int Ho, int Wo) { return (int64_t)N * K * Ho * Wo * 4 >= (512LL << 20) ? 1 : 0; }

template <typename P, typename G>
int launch_conv(const char* name, const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale,
                const float* shift, int relu, const float* skip, double* stats, const double* in_stats, float in_eps, int N,
                int C, int H, int W, int K, int weight_exp, hipStream_t st, int pixel_major = 0) {
    if (P::F16 && pixel_major == 2) return vo_fail(VO_ERR_UNSUPPORTED, "%s: rows split beforehand are bf16 pieces", name);
    if (!x || !arranged || !y) return vo_fail(VO_ERR_INVALID_ARG, "%s: null pointer", name);
    if (pixel_major && (K % 4 != 0 || skip || stats || (reinterpret_cast<uintptr_t>(y) & 15) != 0))
        return vo_fail(VO_ERR_UNSUPPORTED, "%s: the pixel-major store needs K %% 4 == 0, a 16-byte aligned output, no skip, no statistics", name);
    if (pixel_major == 2 && K % 32 != 0) return vo_fail(VO_ERR_UNSUPPORTED, "%s: rows split in blocks of 32 need K %% 32 == 0 (K=%d)", name, K);
    if (scale && !shift) return vo_fail(VO_ERR_INVALID_ARG, "%s: scale without shift", name);
    if (stats && (shift || relu || skip))
        return vo_fail(VO_ERR_INVALID_ARG, "%s: statistics are of the raw output, no epilogue with them", name);
    if (skip == y) return vo_fail(VO_ERR_INVALID_ARG, "%s: skip must not alias the output", name);
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: bad sizes", name);
    const int Ho = (H - 1) / G::ST + 1, Wo = (W - 1) / G::ST + 1;
    if (y_batch_stride == 0) y_batch_stride = (int64_t)K * Ho * Wo;
    if (y_batch_stride < (int64_t)K * Ho * Wo) return vo_fail(VO_ERR_INVALID_ARG, "%s: y_batch_stride smaller than one item", name);
    if (C % CK != 0) return vo_fail(VO_ERR_UNSUPPORTED, "%s: C=%d is not a multiple of %d", name, C, CK);
    if (W % 2 != 0 || (reinterpret_cast<uintptr_t>(x) & 7) != 0)
        return vo_fail(VO_ERR_UNSUPPORTED, "%s: needs an even width and an 8-byte aligned input (W=%d)", name, W);
    if ((int64_t)C * H * W * 4 >= 0x7fffffffLL || (int64_t)K * Ho * Wo * 4 >= 0x7fffffffLL)
        return vo_fail(VO_ERR_UNSUPPORTED, "%s: one image of the input or output exceeds 2 GiB", name);
    if (pixel_major) y_batch_stride = (int64_t)K * Ho * Wo;
    if (in_stats && C > G::MAX_NORM_C)
        return vo_fail(VO_ERR_UNSUPPORTED, "%s: normalise-on-load supports at most %d input channels", name, G::MAX_NORM_C);
    const int bx = (Wo + G::TW - 1) / G::TW, by = (Ho + G::TH - 1) / G::TH;
    const int64_t blocks = (int64_t)bx * by * N;
    const int kblocks = (K + KB - 1) / KB;
    if (blocks * kblocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "%s: grid too large", name);
    const Epilogue ep{scale, shift, skip, relu, y_batch_stride, stats, pixel_major, streams_out(N, K, Ho, Wo),
                      P::F16 ? ldexpf(1.0f, -weight_exp) : 1.0f};
    const InputNorm nrm{in_stats, in_eps};
    const unsigned short* ag = static_cast<const unsigned short*>(arranged);
    // a last block with at most 32 channels goes out as a launch of its own with one channel tile --
    // except for the stride-2 layers: their few input channels make them bound by reading the input, and
    // a second launch would read it from memory a second time (blocks of one launch that differ only in
    // their channel block run side by side and share the input through L2)
    const int rem = K % KB;
    const int full = (rem >= 1 && rem <= 32 && G::ST == 1) ? kblocks - 1 : kblocks;
    if (full > 0) {
        const dim3 grid((unsigned)(blocks * full));
        if (in_stats) conv_bf16x3_kernel<P, G, true, 2><<<grid, dim3(NT), 0, st>>>(x, ag, y, ep, nrm, C, H, W, Ho, Wo, K, bx, by, full, 0, GruEp{}, Shortcut{});
        else conv_bf16x3_kernel<P, G, false, 2><<<grid, dim3(NT), 0, st>>>(x, ag, y, ep, nrm, C, H, W, Ho, Wo, K, bx, by, full, 0, GruEp{}, Shortcut{});
    }
    if (full < kblocks) {
        const dim3 grid((unsigned)blocks);
        if (in_stats) conv_bf16x3_kernel<P, G, true, 1><<<grid, dim3(NT), 0, st>>>(x, ag, y, ep, nrm, C, H, W, Ho, Wo, K, bx, by, 1, full, GruEp{}, Shortcut{});
        else conv_bf16x3_kernel<P, G, false, 1><<<grid, dim3(NT), 0, st>>>(x, ag, y, ep, nrm, C, H, W, Ho, Wo, K, bx, by, 1, full, GruEp{}, Shortcut{});
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e));
    return VO_OK;
}

using Geo3x3 = Geo<3, 3, 1, 4>;
using Geo1x5 = Geo<1, 5, 1, 4>;
using Geo5x1 = Geo<5, 1, 1, 4>;
using Geo1x1 = Geo<1, 1, 1, 4>;
using Geo1x1n = Geo<1, 1, 1, 4, 16>;
using Geo3x3n = Geo<3, 3, 1, 4, 16>;          // 32 rows x 16 columns per workgroup
using Geo1x5n = Geo<1, 5, 1, 4, 16>;
using Geo5x1n = Geo<5, 1, 1, 4, 16>;

// true when 32 x 16 blocks cover an H x W map with less waste than 16 x 32 blocks
inline bool narrow_blocks_fit_better(int H, int W) {
    auto up = [](int v, int m) { return (int64_t)((v + m - 1) / m * m); };
    return up(H, 32) * up(W, 16) < up(H, 16) * up(W, 32);
}
using Geo3x3s2 = Geo<3, 3, 2, 1>;
using Geo1x1s2 = Geo<1, 1, 2, 1>;

// The 3x3 stride-2 convolution of a downsampling residual block together with the block's 1x1 stride-2 projection
// shortcut (same input, same number of output channels), each with its own epilogue or plane sums.
template <typename P>
int launch_conv_shortcut(const char* name, const float* x, const void* arranged, const void* arranged1x1, float* y, float* y2,
                         const float* scale, const float* shift, int relu, double* stats, const float* scale2, const float* shift2,
                         int relu2, double* stats2, int N, int C, int H, int W, int K, int weight_exp, int weight_exp2, hipStream_t st) {
    using G = Geo3x3s2;
    if (!x || !arranged || !arranged1x1 || !y || !y2) return vo_fail(VO_ERR_INVALID_ARG, "%s: null pointer", name);
    if ((scale && !shift) || (scale2 && !shift2)) return vo_fail(VO_ERR_INVALID_ARG, "%s: scale without shift", name);
    if ((stats && (shift || relu)) || (stats2 && (shift2 || relu2)))
        return vo_fail(VO_ERR_INVALID_ARG, "%s: statistics are of the raw output, no epilogue with them", name);
    if (y == y2) return vo_fail(VO_ERR_INVALID_ARG, "%s: the two outputs must differ", name);
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: bad sizes", name);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (C % CK != 0) return vo_fail(VO_ERR_UNSUPPORTED, "%s: C=%d is not a multiple of %d", name, C, CK);
    if (W % 2 != 0 || (reinterpret_cast<uintptr_t>(x) & 7) != 0)
        return vo_fail(VO_ERR_UNSUPPORTED, "%s: needs an even width and an 8-byte aligned input (W=%d)", name, W);
    if (Wo % 4 != 0 || ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(y2)) & 15) != 0)
        return vo_fail(VO_ERR_UNSUPPORTED, "%s: needs an output width in multiples of 4 and 16-byte aligned outputs (Wo=%d)", name, Wo);
    if ((int64_t)C * H * W * 4 >= 0x7fffffffLL || (int64_t)K * Ho * Wo * 4 >= 0x7fffffffLL)
        return vo_fail(VO_ERR_UNSUPPORTED, "%s: one image of the input or output exceeds 2 GiB", name);
    const int bx = (Wo + G::TW - 1) / G::TW, by = (Ho + G::TH - 1) / G::TH;
    const int kblocks = (K + KB - 1) / KB;
    const int64_t blocks = (int64_t)bx * by * N * kblocks;
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "%s: grid too large", name);
    const int64_t ybs = (int64_t)K * Ho * Wo;
    const Epilogue ep{scale, shift, nullptr, relu, ybs, stats, 0, streams_out(N, K, Ho, Wo), P::F16 ? ldexpf(1.0f, -weight_exp) : 1.0f};
    const Shortcut sc{static_cast<const unsigned short*>(arranged1x1), y2,
                      Epilogue{scale2, shift2, nullptr, relu2, ybs, stats2, 0, streams_out(N, K, Ho, Wo), P::F16 ? ldexpf(1.0f, -weight_exp2) : 1.0f}};
    conv_bf16x3_kernel<P, G, false, 2, false, true><<<dim3((unsigned)blocks), dim3(NT), 0, st>>>(
        x, static_cast<const unsigned short*>(arranged), y, ep, InputNorm{nullptr, 0.0f}, C, H, W, Ho, Wo, K, bx, by, kblocks, 0, GruEp{}, sc);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e));
    return VO_OK;
}

// GRU convolutions: shape 0 = 3x3 (padding 1), 1 = 1x5 (padding (0,2)), 2 = 5x1 (padding (2,0)); K output channels
// in blocks of 64, gate arithmetic in the store (16-byte path).
template <typename P, typename G>
int launch_gru(const char* name, const float* x, const unsigned short* ag, float* y, int64_t y_batch_stride, const float* shift,
               const GruEp& gru, int N, int C, int H, int W, int K, int weight_exp, hipStream_t st) {
    const int bx = (W + G::TW - 1) / G::TW, by = (H + G::TH - 1) / G::TH, kblocks = K / KB;
    const int64_t blocks = (int64_t)bx * by * N * kblocks;
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "%s: grid too large", name);
    const Epilogue ep{nullptr, shift, nullptr, 0, y_batch_stride, nullptr, 0, 0, P::F16 ? ldexpf(1.0f, -weight_exp) : 1.0f};
    conv_bf16x3_kernel<P, G, false, 2, true><<<dim3((unsigned)blocks), dim3(NT), 0, st>>>(x, ag, y, ep, InputNorm{nullptr, 0.0f}, C, H, W, H, W, K,
                                                                                     bx, by, kblocks, 0, gru, Shortcut{});
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e));
    return VO_OK;
}

template <typename P>
int gru_conv(const char* name, const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* shift,
             const GruEp& gru, int N, int C, int H, int W, int K, int shape, int weight_exp, hipStream_t st) {
    if (!x || !arranged || !y || !gru.h) return vo_fail(VO_ERR_INVALID_ARG, "%s: null pointer", name);
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: bad sizes", name);
    if (shape < 0 || shape > 2) return vo_fail(VO_ERR_INVALID_ARG, "%s: shape must be 0 (3x3), 1 (1x5) or 2 (5x1)", name);
    if (gru.C <= 0 || gru.C % KB != 0) return vo_fail(VO_ERR_UNSUPPORTED, "%s: hidden width %d is not a multiple of %d", name, gru.C, KB);
    if (C % CK != 0 || W % 4 != 0) return vo_fail(VO_ERR_UNSUPPORTED, "%s: needs input channels in multiples of %d and a width in multiples of 4", name, CK);
    if ((int64_t)C * H * W * 4 >= 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "%s: one image of the input exceeds 2 GiB", name);
    const uintptr_t ptrs = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gru.head) |
                           reinterpret_cast<uintptr_t>(gru.a) | reinterpret_cast<uintptr_t>(gru.b) | reinterpret_cast<uintptr_t>(gru.h) |
                           reinterpret_cast<uintptr_t>(gru.z) | reinterpret_cast<uintptr_t>(gru.out2);
    if ((ptrs & 15) != 0 || ((y_batch_stride | gru.a_bs | gru.b_bs | gru.h_bs | gru.out2_bs) & 3) != 0)
        return vo_fail(VO_ERR_UNSUPPORTED, "%s: tensors must be 16-byte aligned with batch strides in multiples of 4", name);
    const unsigned short* ag = static_cast<const unsigned short*>(arranged);
    const bool narrow = narrow_blocks_fit_better(H, W);
    switch (shape) {
        case 0: return narrow ? launch_gru<P, Geo3x3n>(name, x, ag, y, y_batch_stride, shift, gru, N, C, H, W, K, weight_exp, st)
                              : launch_gru<P, Geo3x3>(name, x, ag, y, y_batch_stride, shift, gru, N, C, H, W, K, weight_exp, st);
        case 1: return narrow ? launch_gru<P, Geo1x5n>(name, x, ag, y, y_batch_stride, shift, gru, N, C, H, W, K, weight_exp, st)
                              : launch_gru<P, Geo1x5>(name, x, ag, y, y_batch_stride, shift, gru, N, C, H, W, K, weight_exp, st);
        default: return narrow ? launch_gru<P, Geo5x1n>(name, x, ag, y, y_batch_stride, shift, gru, N, C, H, W, K, weight_exp, st)
                               : launch_gru<P, Geo5x1>(name, x, ag, y, y_batch_stride, shift, gru, N, C, H, W, K, weight_exp, st);
    }
}


// The entry points' bodies for one piece format.  weight_exp (wx): see split_filter.
template <typename P>
struct SplitApi {
    static int filter(const char* name, const float* weight, void* arranged, int K, int C, int taps, int wx, void* stream) {
        return split_filter<P>(name, weight, arranged, K, C, taps, wx, vo_stream(stream));
    }

    static int conv3x3(const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale, const float* shift,
                       int relu, const float* skip, double* stats, const double* in_stats, float in_eps, int N, int C, int H, int W,
                       int K, int wx, void* stream) {
        if (narrow_blocks_fit_better(H, W) && !(in_stats && C > Geo3x3n::MAX_NORM_C))
            return launch_conv<P, Geo3x3n>("vo_conv3x3_bf16x3", x, arranged, y, y_batch_stride, scale, shift, relu, skip, stats, in_stats,
                                           in_eps, N, C, H, W, K, wx, vo_stream(stream));
        return launch_conv<P, Geo3x3>("vo_conv3x3_bf16x3", x, arranged, y, y_batch_stride, scale, shift, relu, skip, stats, in_stats,
                                      in_eps, N, C, H, W, K, wx, vo_stream(stream));
    }

    static int conv5(const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale, const float* shift,
                     int relu, int N, int C, int H, int W, int K, int vertical, int wx, void* stream) {
        const char* name = "vo_conv5_bf16x3";
        hipStream_t st = vo_stream(stream);
        if (narrow_blocks_fit_better(H, W)) {
            if (vertical)
                return launch_conv<P, Geo5x1n>(name, x, arranged, y, y_batch_stride, scale, shift, relu, nullptr, nullptr, nullptr, 0.0f, N, C, H, W, K, wx, st);
            return launch_conv<P, Geo1x5n>(name, x, arranged, y, y_batch_stride, scale, shift, relu, nullptr, nullptr, nullptr, 0.0f, N, C, H, W, K, wx, st);
        }
        if (vertical)
            return launch_conv<P, Geo5x1>(name, x, arranged, y, y_batch_stride, scale, shift, relu, nullptr, nullptr, nullptr, 0.0f, N, C, H, W, K, wx, st);
        return launch_conv<P, Geo1x5>(name, x, arranged, y, y_batch_stride, scale, shift, relu, nullptr, nullptr, nullptr, 0.0f, N, C, H, W, K, wx, st);
    }

    static int conv3x3s2(const float* x, const void* arranged, float* y, const float* scale, const float* shift, int relu, double* stats,
                         int N, int C, int H, int W, int K, int wx, void* stream) {
        return launch_conv<P, Geo3x3s2>("vo_conv3x3s2_bf16x3", x, arranged, y, 0, scale, shift, relu, nullptr, stats, nullptr, 0.0f, N, C,
                                        H, W, K, wx, vo_stream(stream));
    }

    static int conv3x3s2_shortcut(const float* x, const void* arranged, const void* arranged1x1, float* y, float* y2, const float* scale,
                                  const float* shift, int relu, double* stats, const float* scale2, const float* shift2, int relu2,
                                  double* stats2, int N, int C, int H, int W, int K, int wx, int wx2, void* stream) {
        return launch_conv_shortcut<P>("vo_conv3x3s2_shortcut_bf16x3", x, arranged, arranged1x1, y, y2, scale, shift, relu, stats, scale2,
                                       shift2, relu2, stats2, N, C, H, W, K, wx, wx2, vo_stream(stream));
    }

    static int conv1x1s2(const float* x, const void* arranged, float* y, const float* scale, const float* shift, int relu, double* stats,
                         int N, int C, int H, int W, int K, int wx, void* stream) {
        return launch_conv<P, Geo1x1s2>("vo_conv1x1s2_bf16x3", x, arranged, y, 0, scale, shift, relu, nullptr, stats, nullptr, 0.0f, N, C,
                                        H, W, K, wx, vo_stream(stream));
    }

    // pixel_major: 0 = NCHW (y_batch_stride applies), 1 = (N, pixels, K) float32, 2 = that with rows cut into bf16 pieces
    static int conv1x1(const char* name, const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale,
                       const float* shift, int relu, int N, int C, int H, int W, int K, int wx, int pixel_major, void* stream) {
        if (narrow_blocks_fit_better(H, W))
            return launch_conv<P, Geo1x1n>(name, x, arranged, y, y_batch_stride, scale, shift, relu, nullptr, nullptr, nullptr, 0.0f, N, C,
                                           H, W, K, wx, vo_stream(stream), pixel_major);
        return launch_conv<P, Geo1x1>(name, x, arranged, y, y_batch_stride, scale, shift, relu, nullptr, nullptr, nullptr, 0.0f, N, C, H, W,
                                      K, wx, vo_stream(stream), pixel_major);
    }

    static int gru_zr(const float* x, const void* arranged, const float* bias, const float* cz, int64_t cz_batch_stride, const float* cr,
                      int64_t cr_batch_stride, const float* h, int64_t h_batch_stride, float* z_out, float* rh_out, int N, int Cin, int H,
                      int W, int C, int shape, int wx, void* stream) {
        if (!rh_out) return vo_fail(VO_ERR_INVALID_ARG, "vo_gru_conv_zr_bf16x3: null pointer");
        const GruEp gru{1, C, nullptr, cz, cz_batch_stride, cr, cr_batch_stride, h, h_batch_stride, nullptr, rh_out, (int64_t)C * H * W};
        return gru_conv<P>("vo_gru_conv_zr_bf16x3", x, arranged, z_out, (int64_t)C * H * W, bias, gru, N, Cin, H, W, 2 * C, shape, wx,
                           vo_stream(stream));
    }

    static int gru_q(const float* x, const float* rh, const void* arranged, const float* bias, const float* cq, int64_t cq_batch_stride,
                     const float* z, const float* h, int64_t h_batch_stride, float* h_out, float* h_copy, int64_t copy_batch_stride, int N,
                     int Cin, int H, int W, int C, int shape, int wx, void* stream) {
        if (!rh || !z) return vo_fail(VO_ERR_INVALID_ARG, "vo_gru_conv_q_bf16x3: null pointer");
        if (h_out == h || h_copy == h)
            return vo_fail(VO_ERR_INVALID_ARG, "vo_gru_conv_q_bf16x3: the new state must not overwrite the tensor h is read from");
        const GruEp gru{2, C, rh, cq, cq_batch_stride, nullptr, 0, h, h_batch_stride, z, h_copy, copy_batch_stride};
        return gru_conv<P>("vo_gru_conv_q_bf16x3", x, arranged, h_out, (int64_t)C * H * W, bias, gru, N, Cin, H, W, C, shape, wx,
                           vo_stream(stream));
    }
};

}  // namespace

// The float16-piece instantiation lives in a translation unit of its own (vo_conv_f16x3.hip); the entry points in
// vo_conv_bf16x3.hip reach it through these functions.
namespace vo_f16_pieces {
int filter(const char* name, const float* weight, void* arranged, int K, int C, int taps, int wx, void* stream);
int conv3x3(const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale, const float* shift, int relu,
            const float* skip, double* stats, const double* in_stats, float in_eps, int N, int C, int H, int W, int K, int wx, void* stream);
int conv5(const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale, const float* shift, int relu, int N,
          int C, int H, int W, int K, int vertical, int wx, void* stream);
int conv3x3s2(const float* x, const void* arranged, float* y, const float* scale, const float* shift, int relu, double* stats, int N, int C,
              int H, int W, int K, int wx, void* stream);
int conv3x3s2_shortcut(const float* x, const void* arranged, const void* arranged1x1, float* y, float* y2, const float* scale,
                       const float* shift, int relu, double* stats, const float* scale2, const float* shift2, int relu2, double* stats2,
                       int N, int C, int H, int W, int K, int wx, int wx2, void* stream);
int conv1x1s2(const float* x, const void* arranged, float* y, const float* scale, const float* shift, int relu, double* stats, int N, int C,
              int H, int W, int K, int wx, void* stream);
int conv1x1(const char* name, const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale, const float* shift,
            int relu, int N, int C, int H, int W, int K, int wx, int pixel_major, void* stream);
int gru_zr(const float* x, const void* arranged, const float* bias, const float* cz, int64_t cz_batch_stride, const float* cr,
           int64_t cr_batch_stride, const float* h, int64_t h_batch_stride, float* z_out, float* rh_out, int N, int Cin, int H, int W, int C,
           int shape, int wx, void* stream);
int gru_q(const float* x, const float* rh, const void* arranged, const float* bias, const float* cq, int64_t cq_batch_stride, const float* z,
          const float* h, int64_t h_batch_stride, float* h_out, float* h_copy, int64_t copy_batch_stride, int N, int Cin, int H, int W, int C,
          int shape, int wx, void* stream);
}  // namespace vo_f16_pieces
