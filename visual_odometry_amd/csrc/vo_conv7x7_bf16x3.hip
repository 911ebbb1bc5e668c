// 7x7 padding-3 convolution (stride 1 or 2) of a ONE- or TWO-channel float32 input (the encoders' stems
// on a gray frame, the motion encoders' first flow layer) with every float32 product assembled from
// three 16-bit matrix-core terms (xl*wh + xh*wl + xh*wh, float32 accumulation; bf16 or float16 pieces,
// see vo_pieces.h), and optionally the plane sums an instance norm needs.
//
// With so few input channels the layer is bound by the store of its 64+ output channels, so the
// kernel is built around the output stream:
//   workgroup = 4 waves = 4 consecutive output rows x 64 output channels; it walks its rows in
//   segments of 64 columns, so weights are fetched once per workgroup and plane sums leave as ONE pair
//   of float64 atomics per channel and workgroup (per 4 full rows, not per tile);
//   wave w = output row 4*strip + w: 2 (channel tiles of 32) x 2 (pixel tiles of 32) accumulators.
// Reduction index of one matrix instruction (k = 16): two filter columns s = 2j, 2j+1 (one per lane
// half) x eight filter rows r = 0..7 (row 7 and column 7 carry zero weights).  The B fragment of a
// lane is therefore a COLUMN of eight input rows at one input column: while a segment is staged, each
// (output row, input column) pair gets its eight rows gathered, split into high and low bf16 parts and
// written as one 16-byte LDS entry per part; a lane's fragment for any filter column is then a single
// aligned 16-byte read, lanes on consecutive entries (stride 2: even and odd columns stored apart).
#include "vo_common.h"
#include "vo_pieces.h"

namespace {

constexpr int KB = 64;                        // output channels per workgroup
constexpr int SEG = 64;                       // output columns per segment
constexpr int NT = 256;
constexpr int A_STEP = 2 * 2 * 64 * 16;       // bytes of weights per k-step: [part][channel tile][lane][16]

// Weights (K,C,7,7) float32 -> high and low bf16 parts in A-operand order:
//   Ag[kb][c*4 + j][part][mt][lane][r]:  k = kb*64 + mt*32 + (lane & 31),  s = 2j + (lane >> 5); r = 7 or s = 7: zero
template <typename P>
__global__ __launch_bounds__(256) void filter7_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ Ag, int K,
                                                            int C, int Kpad, float wscale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Kpad * C * 64) return;
    const int r = (int)(idx & 7), s = (int)((idx >> 3) & 7), c = (int)((idx >> 6) % C), k = (int)((idx >> 6) / C);
    const float v = (k < K && r < 7 && s < 7) ? w[(((int64_t)k * C + c) * 7 + r) * 7 + s] : 0.0f;
    unsigned hi, lo;
    P::cut_weight(v, wscale, hi, lo);
    const int kb = k >> 6, mt = (k >> 5) & 1, lane = (s & 1) * 32 + (k & 31), j = s >> 1;
    const int64_t base = (((int64_t)kb * C + c) * 4 + j) * 4;                  // [part][mt]
    Ag[((base + 0 * 2 + mt) * 64 + lane) * 8 + r] = (unsigned short)hi;
    Ag[((base + 1 * 2 + mt) * 64 + lane) * 8 + r] = (unsigned short)lo;
}

struct Epi { const float* scale; const float* shift; int relu; double* stats; float unscale; };

// S: stride, CIN: input channels (1 or 2).  (H, W): input size, (Ho, Wo): output size.
template <typename P, int S, int CIN, bool STATS>
__global__ __launch_bounds__(NT, 2) void conv7_split_kernel(const float* __restrict__ x, const unsigned short* __restrict__ Ag,
                                                            float* __restrict__ y, Epi ep, int H, int W, int Ho, int Wo, int K,
                                                            int strips, int kblocks) {
    constexpr int PW = (SEG - 1) * S + 8;                    // input columns a segment touches (filter column 7 included)
    constexpr int PWE = (PW + 1) / 2 * 2;                    // entries per row of the LDS image (even)
    constexpr int ENTRIES = CIN * 4 * PW;                    // (channel, output row, input column)
    constexpr int ROUNDS = (ENTRIES + NT - 1) / NT;
    constexpr int T_PART = CIN * 4 * PWE * 16;               // bytes of one part (high or low) of the staged segment
    constexpr int A_BYTES = CIN * 4 * A_STEP;
    __shared__ __align__(16) unsigned char lds[A_BYTES + 2 * T_PART + KB * 8];
    unsigned char* t_lds = lds + A_BYTES;
    float2* affine = reinterpret_cast<float2*>(lds + A_BYTES + 2 * T_PART);
    double* red = reinterpret_cast<double*>(t_lds);          // [2][4 waves][64], after the last segment
    static_assert(2 * T_PART >= 2 * 4 * 64 * 8, "the reduction buffer does not fit the staging area");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int kb = bid % kblocks; bid /= kblocks;
    const int strip = bid % strips;
    const int n = bid / strips;
    const int py = strip * 4 + wave;                         // this wave's output row
    const int nseg = (Wo + SEG - 1) / SEG;
    const int HW4 = H * W * 4;

    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (int64_t)n * CIN * H * W), 0, CIN * HW4, 0x00020000);
    // weights of this channel block: a linear copy into LDS, once per workgroup
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(Ag + (int64_t)kb * (A_BYTES / 2));
        for (int i = tid; i < A_BYTES / 16; i += NT) reinterpret_cast<u32x4*>(lds)[i] = src[i];
        if (tid < KB) {
            const int k = kb * KB + tid;
            const bool kok = k < K;
            affine[tid] = make_float2((ep.scale && kok) ? ep.scale[k] : 1.0f, (ep.shift && kok) ? ep.shift[k] : 0.0f);
        }
    }

    // staging entries of this thread: (channel, output row of the strip, input column of the segment)
    int e_row[ROUNDS], e_lds[ROUNDS], e_pc[ROUNDS];
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        const int e = tid + i * NT;
        const int c = e / (4 * PW), rem = e - c * (4 * PW), rw = rem / PW, pc = rem - rw * PW;
        const int iy0 = (strip * 4 + rw) * S - 3;            // input row of filter row 0
        e_row[i] = e < ENTRIES ? c * H + iy0 : (int)0x40000000;              // in rows of the (channel-major) input
        e_pc[i] = pc;
        e_lds[i] = ((c * 4 + rw) * PWE + (S == 1 ? pc : (pc & 1) * (PWE / 2) + (pc >> 1))) * 16;
    }
    float raw[ROUNDS][8];
    auto load_segment = [&](int seg) {
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            if (i * NT + NT > ENTRIES && tid + i * NT >= ENTRIES) continue;
            const int ix = seg * SEG * S - 3 + e_pc[i];
            const int c = (tid + i * NT) / (4 * PW);
            const int iy0 = e_row[i] - c * H;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int iy = iy0 + r;
                const bool ok = ix >= 0 && ix < W && iy >= 0 && iy < H;
                raw[i][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, ok ? ((e_row[i] + r) * W + ix) * 4 : (int)0x80000000, 0, 0));
            }
        }
    };
    auto store_segment = [&]() {
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            if (i * NT + NT > ENTRIES && tid + i * NT >= ENTRIES) continue;
            u32x4 hi, lo;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                unsigned h, l;
                P::cut(raw[i][2 * jj], raw[i][2 * jj + 1], h, l);
                hi[jj] = h;
                lo[jj] = l;
            }
            *reinterpret_cast<u32x4*>(t_lds + e_lds[i]) = hi;
            *reinterpret_cast<u32x4*>(t_lds + T_PART + e_lds[i]) = lo;
        }
    };

    // per-lane plane sums over every segment of the row (float32 over the lane's few pixels only)
    float a1[STATS ? 2 : 1][16], a2[STATS ? 2 : 1][16];
#pragma unroll
    for (int mt = 0; mt < (STATS ? 2 : 1); ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) a1[mt][e] = a2[mt][e] = 0.0f;

    const int nn = lane & 31, lh = lane >> 5;
    float* yn = y + (int64_t)n * K * Ho * Wo;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yn, 0, K * Ho * Wo * 4, 0x00020000);
    const unsigned char* a_base = lds + lane * 16;
    load_segment(0);
    for (int seg = 0; seg < nseg; ++seg) {
        store_segment();
        __syncthreads();
        if (seg + 1 < nseg) load_segment(seg + 1);           // in flight during the multiply phase and the stores

        f32x16 acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][t][e] = 0.0f;
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 fa[2][2], fb[2][2];
#pragma unroll
                for (int part = 0; part < 2; ++part) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        fa[part][mt] = *reinterpret_cast<const u32x4*>(a_base + (c * 4 + j) * A_STEP + (part * 2 + mt) * 1024);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        // input column of (pixel t*32 + nn, filter column 2j + lh), relative to the segment's first
                        const int pos = S == 1 ? (t * 32 + nn) + 2 * j + lh : lh * (PWE / 2) + (t * 32 + nn) + j;
                        fb[part][t] = *reinterpret_cast<const u32x4*>(t_lds + part * T_PART + ((c * 4 + wave) * PWE + pos) * 16);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc[mt][t] = P::mfma(fa[1][mt], fb[0][t], acc[mt][t]);
                        acc[mt][t] = P::mfma(P::for_low_input(fa[0][mt]), fb[1][t], acc[mt][t]);
                        acc[mt][t] = P::mfma(fa[0][mt], fb[0][t], acc[mt][t]);
                    }
            }

        if constexpr (P::F16) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[mt][t] *= ep.unscale;    // back to the weights' own scale (a power of two)
        }

        // C/D layout of a 32x32 tile: pixel = lane & 31, channel row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5).
        // K is a multiple of 8, so a group of channel rows 8g..8g+7 is valid or not as a whole (wave-uniform);
        // the lane part of a store's address is the same for every channel, the channel part is scalar
        int voff[2];
        bool pok[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int px = seg * SEG + t * 32 + nn;
            pok[t] = py < Ho && px < Wo;
            voff[t] = pok[t] ? ((4 * lh * Ho + py) * Wo + px) * 4 : (int)0x80000000;
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kbase = kb * KB + mt * 32 + 8 * (e >> 2);          // first channel of the group of eight
                if (kbase >= K) continue;
                const int krow = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float2 ss = affine[krow];
                const int soff = (kbase + (e & 3)) * (Ho * Wo * 4);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float v = acc[mt][t][e];
                    if (STATS) {
                        const float u = pok[t] ? v : 0.0f;
                        a1[STATS ? mt : 0][e] += u; a2[STATS ? mt : 0][e] = fmaf(u, u, a2[STATS ? mt : 0][e]);
                    } else {
                        v = fmaf(v, ss.x, ss.y);
                        if (ep.relu) v = fmaxf(v, 0.0f);
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, voff[t], soff, 0);
                }
            }
        __syncthreads();                                     // everyone has read the segment
    }

    if (STATS) {
        // over the 32 pixel lanes by a halving exchange in float64 (each step hands half of the lane's
        // channel rows to the partner lane and adds the partner's half to its own), then over the four
        // waves through LDS: one pair of float64 atomics per channel and workgroup
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            double s1[16], s2[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) { s1[e] = (double)a1[mt][e]; s2[e] = (double)a2[mt][e]; }
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                const int m = 16 >> step, keep = 8 >> step;
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
            if ((lane & 1) == 0) {
                const int e = ((nn >> 4) & 1) * 8 + ((nn >> 3) & 1) * 4 + ((nn >> 2) & 1) * 2 + ((nn >> 1) & 1);
                const int krow = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                red[(0 * 4 + wave) * 64 + krow] = s1[0];
                red[(1 * 4 + wave) * 64 + krow] = s2[0];
            }
        }
        __syncthreads();
        if (tid < 2 * KB) {
            const int which = tid / KB, krow = tid - which * KB;
            const double* rp = red + which * 4 * 64 + krow;
            const double tot = (rp[0] + rp[64]) + (rp[128] + rp[192]);
            const int k = kb * KB + krow;
            if (k < K) atomicAdd(ep.stats + ((int64_t)n * K + k) * 2 + which, tot);
        }
    }
}

template <typename P, int S, int CIN>
void launch7(const float* x, const unsigned short* ag, float* y, const Epi& ep, int N, int H, int W, int Ho, int Wo, int K,
             hipStream_t st) {
    const int strips = (Ho + 3) / 4, kblocks = (K + KB - 1) / KB;
    const dim3 grid((unsigned)((int64_t)N * strips * kblocks));
    if (ep.stats) conv7_split_kernel<P, S, CIN, true><<<grid, dim3(NT), 0, st>>>(x, ag, y, ep, H, W, Ho, Wo, K, strips, kblocks);
    else conv7_split_kernel<P, S, CIN, false><<<grid, dim3(NT), 0, st>>>(x, ag, y, ep, H, W, Ho, Wo, K, strips, kblocks);
}

template <typename P>
void launch7_any(const float* x, const unsigned short* ag, float* y, const Epi& ep, int N, int C, int H, int W, int Ho, int Wo, int K,
                 int stride, hipStream_t st) {
    if (stride == 1 && C == 1) launch7<P, 1, 1>(x, ag, y, ep, N, H, W, Ho, Wo, K, st);
    else if (stride == 1) launch7<P, 1, 2>(x, ag, y, ep, N, H, W, Ho, Wo, K, st);
    else if (C == 1) launch7<P, 2, 1>(x, ag, y, ep, N, H, W, Ho, Wo, K, st);
    else launch7<P, 2, 2>(x, ag, y, ep, N, H, W, Ho, Wo, K, st);
}

}  // namespace

extern "C" int vo_conv7x7_bf16x3_filter(const float* weight, void* arranged, int K, int C, int pieces, int weight_exp, void* stream) {
    VO_REQUIRE(weight && arranged, "vo_conv7x7_bf16x3_filter: null pointer");
    VO_REQUIRE(pieces == VO_PIECES_BF16 || pieces == VO_PIECES_F16, "vo_conv7x7_bf16x3_filter: pieces must be VO_PIECES_BF16 or VO_PIECES_F16");
    VO_REQUIRE(weight_exp >= -100 && weight_exp <= 100 && (pieces == VO_PIECES_F16 || weight_exp == 0),
               "vo_conv7x7_bf16x3_filter: weight_exp out of range (0 with bf16 pieces)");
    VO_REQUIRE(K > 0 && (C == 1 || C == 2), "vo_conv7x7_bf16x3_filter: bad sizes K=%d C=%d (one or two input channels)", K, C);
    const int Kpad = (K + KB - 1) / KB * KB;
    const int64_t n = (int64_t)Kpad * C * 64;
    if (pieces == VO_PIECES_F16)
        filter7_split_kernel<PiecesF16><<<dim3((unsigned)vo_ceil_div(n, 256)), dim3(256), 0, vo_stream(stream)>>>(
            weight, static_cast<unsigned short*>(arranged), K, C, Kpad, ldexpf(1.0f, weight_exp));
    else
        filter7_split_kernel<PiecesBf16><<<dim3((unsigned)vo_ceil_div(n, 256)), dim3(256), 0, vo_stream(stream)>>>(
            weight, static_cast<unsigned short*>(arranged), K, C, Kpad, 1.0f);
    VO_CHECK_LAUNCH("vo_conv7x7_bf16x3_filter");
    return VO_OK;
}

extern "C" int vo_conv7x7_bf16x3(const float* x, const void* arranged, float* y, const float* scale, const float* shift, int relu,
                                 double* stats, int N, int C, int H, int W, int K, int stride, int pieces, int weight_exp, void* stream) {
    VO_REQUIRE(x && arranged && y, "vo_conv7x7_bf16x3: null pointer");
    VO_REQUIRE(pieces == VO_PIECES_BF16 || pieces == VO_PIECES_F16, "vo_conv7x7_bf16x3: pieces must be VO_PIECES_BF16 or VO_PIECES_F16");
    VO_REQUIRE(weight_exp >= -100 && weight_exp <= 100 && (pieces == VO_PIECES_F16 || weight_exp == 0),
               "vo_conv7x7_bf16x3: weight_exp out of range (0 with bf16 pieces)");
    VO_REQUIRE(N > 0 && (C == 1 || C == 2) && H > 0 && W > 0 && K > 0, "vo_conv7x7_bf16x3: bad sizes (one or two input channels)");
    VO_REQUIRE(stride == 1 || stride == 2, "vo_conv7x7_bf16x3: stride must be 1 or 2");
    if (K % 8 != 0) return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv7x7_bf16x3: K=%d is not a multiple of 8", K);
    VO_REQUIRE(!(scale && !shift), "vo_conv7x7_bf16x3: scale without shift");
    VO_REQUIRE(!(stats && (shift || relu)), "vo_conv7x7_bf16x3: statistics are of the raw output, no epilogue with them");
    VO_REQUIRE(static_cast<const void*>(x) != static_cast<const void*>(y), "vo_conv7x7_bf16x3: the output must not alias the input");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    if ((int64_t)C * H * W * 4 >= 0x3fffffffLL || (int64_t)K * Ho * Wo * 4 >= 0x7fffffffLL)
        return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv7x7_bf16x3: one image of the input exceeds 1 GiB or of the output 2 GiB");
    const int64_t blocks = (int64_t)N * ((Ho + 3) / 4) * ((K + KB - 1) / KB);
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv7x7_bf16x3: grid too large");
    const Epi ep{scale, shift, relu, stats, pieces == VO_PIECES_F16 ? ldexpf(1.0f, -weight_exp) : 1.0f};
    const unsigned short* ag = static_cast<const unsigned short*>(arranged);
    hipStream_t st = vo_stream(stream);
    if (pieces == VO_PIECES_F16) launch7_any<PiecesF16>(x, ag, y, ep, N, C, H, W, Ho, Wo, K, stride, st);
    else launch7_any<PiecesBf16>(x, ag, y, ep, N, C, H, W, Ho, Wo, K, stride, st);
    VO_CHECK_LAUNCH("vo_conv7x7_bf16x3");
    return VO_OK;
}
