// All-pairs correlation volume: batched C[g] = (A[g] * B[g]^T) / sqrt(K) on the
// gfx950 matrix cores with float32 operands (v_mfma_f32_32x32x2_f32, which is an
// exact float32 multiply-accumulate chain, so the volume carries no precision
// loss against the reference's float32 matmul).
//
// A[g]: M x K, B[g]: N x K, both row-major with K contiguous (pixel-major feature
// maps), C[g]: M x N.  RAFT-Flow:   g = pair, M = N = H*W/64 pixels.
//                       RAFT-Stereo: g = (pair, image row), M = N = W/4 pixels.
//
// Workgroup = WAVES_M x WAVES_N waves, each wave owning TM x TN MFMA tiles of
// 32x32.  Feature tiles are staged global -> registers -> LDS (rows padded to
// BK+4 floats so the 16-byte fragment reads are bank-conflict free) with the next
// K-slice's global loads in flight while the current one is multiplied.
//
// Fragment trick: the MFMA takes A[i][k] from lane (i = lane&31, k = lane>>5).
// Which physical k feeds which (half, step) slot is irrelevant as long as A and B
// agree, so each lane reads FOUR consecutive k values with one ds_read_b128
// (k = 8*group + 4*(lane>>5) + step) and feeds them to four successive MFMAs.
//
// SPLIT variant (vo_corr_volume_*_bf16x3): the same tiling with every float32 product assembled from three
// bf16 matrix-core terms (al*bh + ah*bl + ah*bh, float32 accumulation, 16 significant bits per
// product): the staged slices are split into high and low bf16 parts on their way into LDS and a lane's
// fragment is eight consecutive k of one part (one 16-byte read, v_mfma_f32_32x32x16_bf16).  The
// bf16 instruction runs at 16 times the float32 instruction's rate, so the kernel's bound moves from
// the matrix cores to the store of the volume.
//
// Float16 pieces (pieces = VO_PIECES_F16): the same three terms on v_mfma_f32_32x32x16_f16 with 11 + 11 = 22
// significant bits per operand.  Feature maps have no known scale, so each IMAGE's map is multiplied by the power of
// two that puts its largest |x| into [2^13, 2^14) before it is cut (high = f16(x'), low = f16(x' - high): normal
// float16 numbers for every element down to 2^-16 of the largest, an absolute error below 2^-38 of the largest under
// that) and the volume is multiplied by the two inverse powers in the store.  The largest |x| per image is found by
// a pass of its own (rows_amax_kernel: float bits under an unsigned atomic maximum) and handed to the kernels as an
// array in device memory, so no value travels through the host.  Per image, not per batch: a pair's volume does not
// depend on which other pairs share its launch.
#include "vo_common.h"
#include "vo_pieces.h"

#include <algorithm>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PAD = 4;   // floats of row padding in LDS

// e such that amax * 2^e lies in [2^13, 2^14), from amax's float bits (0 for zero, subnormal, infinite or NaN maxima:
// the map is then cut as it is); clamped so that 2^e and 2^-e are normal float32 numbers
__device__ __forceinline__ int feature_exp(unsigned amax_bits) {
    const int E = (int)((amax_bits >> 23) & 0xffu);
    if (E == 0 || E == 255) return 0;
    return min(140 - E, 120);
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }   // -126 <= e <= 127

// two scaled values -> their float16 high and low parts (the low part as it is: the scaling keeps it normal)
__device__ __forceinline__ void cut_f16_scaled(float a, float b, unsigned& hi, unsigned& lo) {
    const f32x2 v = {a, b};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f16x2 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// Bijective XCD-aware remap: the dispatcher deals consecutive workgroups round
// robin over the 8 XCDs; give each XCD one contiguous range of the work list so
// that the feature rows it touches stay in its own L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// F16 (SPLIT only): float16 pieces; amaxA / amaxB hold the largest |x| (float bits) of the image that group g belongs
// to, image = g / per_image (one group per image for the flow volume, one per image row for the stereo volume).
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool SPLIT, bool PRE = false, bool F16 = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, WAVES_M * WAVES_N >= 8 ? 1 : 2) void corr_abt_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
    int64_t strideA, int64_t strideB, int64_t strideC, int tiles_m, int tiles_n, unsigned nwg, float divisor,
    const unsigned* __restrict__ amaxA, const unsigned* __restrict__ amaxB, int per_image) {
    using P = std::conditional_t<F16, PiecesF16, PiecesBf16>;
    static_assert(!F16 || SPLIT, "float16 pieces are a split arithmetic");
    constexpr int THREADS = 64 * WAVES_M * WAVES_N;
    constexpr int TM = BM / (32 * WAVES_M);
    constexpr int TN = BN / (32 * WAVES_N);
    constexpr int LDS_STRIDE = BK + PAD;
    constexpr int A_F4 = BM * BK / 4 / THREADS;      // float4 loads per thread per K-slice
    constexpr int B_F4 = BN * BK / 4 / THREADS;
    static_assert(BM % (32 * WAVES_M) == 0 && BN % (32 * WAVES_N) == 0, "tile / wave mismatch");
    static_assert((BM * BK / 4) % THREADS == 0 && (BN * BK / 4) % THREADS == 0, "staging mismatch");
    static_assert(BK % 8 == 0, "BK must be a multiple of 8");
    static_assert(!SPLIT || BK % 16 == 0, "the bf16 instruction takes 16 k per step");
    static_assert(!PRE || SPLIT, "pre-split rows feed the bf16 instruction");
    // SPLIT: a row holds BK high parts then BK low parts (2 bytes each, as many bytes as BK floats); the
    // same padding keeps the 16-byte fragment reads of 32 consecutive rows on distinct banks

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                   // [2][BM][LDS_STRIDE]
    float* Bs = smem + 2 * BM * LDS_STRIDE;             // [2][BN][LDS_STRIDE]

    // Work order inside one matrix: row-tiles are taken in groups of GROUP_M and a
    // group sweeps the column-tiles with the row-tile index fastest, so the ~64
    // workgroups resident on an XCD share GROUP_M A-tiles and ~8 B-tiles (about 2 MB,
    // inside the 4 MB L2) instead of streaming all of B once per row-tile.
    constexpr int GROUP_M = 8;       // measured: 1 -> 893 us, 4 -> 877, 8 -> 870, 16 -> 903 (8 pairs of 640x480)
    const unsigned work = xcd_remap(blockIdx.x, nwg);
    const unsigned per_g = (unsigned)tiles_n * tiles_m;
    const int g = work / per_g;
    const unsigned w_in = work - g * per_g;
    const unsigned group_sz = GROUP_M * tiles_n;
    const unsigned grp = w_in / group_sz;
    const int rows_in_grp = min(GROUP_M, tiles_m - (int)grp * GROUP_M);
    const unsigned w_grp = w_in - grp * group_sz;
    const int tm = grp * GROUP_M + w_grp % rows_in_grp;
    const int tn = w_grp / rows_in_grp;
    const float* Ag = A + g * strideA;
    const float* Bg = B + g * strideB;
    float* Cg = C + g * strideC;
    const int row0 = tm * BM, col0 = tn * BN;
    int eA = 0, eB = 0;
    if constexpr (F16) {
        eA = feature_exp(amaxA[g / per_image]);
        eB = feature_exp(amaxB[g / per_image]);
    }
    const float sA = pow2f(eA), sB = pow2f(eB);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lr = lane & 31, lh = lane >> 5;

    // staging: float4 slot f -> (row f / (BK/4), column group f % (BK/4)); rows past
    // the matrix edge re-read the last valid row (their products are never stored)
    float4 ra[2][A_F4], rb[2][B_F4];             // two slices in flight
#define VO_LOAD_SLICE(set, k0)                                                                                   \
    do {                                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < A_F4; ++i_) {                                               \
            const int f_ = tid + i_ * THREADS;                                                              \
            const int r_ = min(row0 + f_ / (BK / 4), M - 1);                                                \
            ra[set][i_] = *reinterpret_cast<const float4*>(Ag + (int64_t)r_ * K + (k0) + (f_ % (BK / 4)) * 4);   \
        }                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < B_F4; ++i_) {                                               \
            const int f_ = tid + i_ * THREADS;                                                              \
            const int r_ = min(col0 + f_ / (BK / 4), N - 1);                                                \
            rb[set][i_] = *reinterpret_cast<const float4*>(Bg + (int64_t)r_ * K + (k0) + (f_ % (BK / 4)) * 4);   \
        }                                                                                                   \
    } while (0)
#define VO_STORE_SLICE(set, buf)                                                                               \
    do {                                                                                                    \
        float* as_ = As + (buf) * BM * LDS_STRIDE;                                                          \
        float* bs_ = Bs + (buf) * BN * LDS_STRIDE;                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < A_F4; ++i_) {                                               \
            const int f_ = tid + i_ * THREADS;                                                              \
            store_f4(as_ + (f_ / (BK / 4)) * LDS_STRIDE, f_ % (BK / 4), ra[set][i_], sA);                      \
        }                                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < B_F4; ++i_) {                                               \
            const int f_ = tid + i_ * THREADS;                                                              \
            store_f4(bs_ + (f_ / (BK / 4)) * LDS_STRIDE, f_ % (BK / 4), rb[set][i_], sB);                      \
        }                                                                                                   \
    } while (0)
    // four consecutive k of one row: as they are, or as four high parts (8 bytes at 2*k) and four low
    // parts (8 bytes at 2*BK + 2*k)
    auto store_f4 = [](float* row, int q, float4 v, float s) {
        if constexpr (SPLIT && !PRE) {
            unsigned h0, h1, l0, l1;
            if constexpr (F16) {
                cut_f16_scaled(v.x * s, v.y * s, h0, l0);
                cut_f16_scaled(v.z * s, v.w * s, h1, l1);
            } else {
                h0 = pack_bf16(v.x, v.y), h1 = pack_bf16(v.z, v.w);
                l0 = pack_bf16(v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u));
                l1 = pack_bf16(v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u));
            }
            unsigned char* r = reinterpret_cast<unsigned char*>(row);
            *reinterpret_cast<uint2*>(r + q * 8) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(r + 2 * BK + q * 8) = make_uint2(l0, l1);
        } else {
            *reinterpret_cast<float4*>(row + q * 4) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;

    // one K-slice of MFMAs out of LDS buffer `buf`
#define VO_MMA_SLICE(buf)                                                                                          \
    do {                                                                                                           \
        const float* as_ = As + (buf) * BM * LDS_STRIDE + (wm * TM * 32 + lr) * LDS_STRIDE + 4 * lh;               \
        const float* bs_ = Bs + (buf) * BN * LDS_STRIDE + (wn * TN * 32 + lr) * LDS_STRIDE + 4 * lh;               \
        _Pragma("unroll") for (int kg = 0; kg < BK / 8; ++kg) {                                                    \
            float4 fa[TM], fb[TN];                                                                                 \
            _Pragma("unroll") for (int m = 0; m < TM; ++m)                                                         \
                fa[m] = *reinterpret_cast<const float4*>(as_ + m * 32 * LDS_STRIDE + kg * 8);                      \
            _Pragma("unroll") for (int n = 0; n < TN; ++n)                                                         \
                fb[n] = *reinterpret_cast<const float4*>(bs_ + n * 32 * LDS_STRIDE + kg * 8);                      \
            _Pragma("unroll") for (int m = 0; m < TM; ++m) _Pragma("unroll") for (int n = 0; n < TN; ++n) {        \
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m].x, fb[n].x, acc[m][n], 0, 0, 0);            \
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m].y, fb[n].y, acc[m][n], 0, 0, 0);            \
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m].z, fb[n].z, acc[m][n], 0, 0, 0);            \
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m].w, fb[n].w, acc[m][n], 0, 0, 0);            \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)

    // SPLIT: lane (row lr, k-half lh) reads k = 16*kg + 8*lh .. +7 of each part
    auto mma_split = [&](int buf) {
        const unsigned char* as_ = reinterpret_cast<const unsigned char*>(As + buf * BM * LDS_STRIDE + (wm * TM * 32 + lr) * LDS_STRIDE) + 16 * lh;
        const unsigned char* bs_ = reinterpret_cast<const unsigned char*>(Bs + buf * BN * LDS_STRIDE + (wn * TN * 32 + lr) * LDS_STRIDE) + 16 * lh;
#pragma unroll
        for (int kg = 0; kg < BK / 16; ++kg) {
            u32x4 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int part = 0; part < 2; ++part) {
#pragma unroll
                for (int m = 0; m < TM; ++m)
                    fa[part][m] = *reinterpret_cast<const u32x4*>(as_ + m * 32 * LDS_STRIDE * 4 + part * 2 * BK + kg * 32);
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    fb[part][n] = *reinterpret_cast<const u32x4*>(bs_ + n * 32 * LDS_STRIDE * 4 + part * 2 * BK + kg * 32);
            }
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    // the small terms first, so that they are not rounded away against the large one
                    acc[m][n] = P::mfma(fa[1][m], fb[0][n], acc[m][n]);
                    acc[m][n] = P::mfma(fa[0][m], fb[1][n], acc[m][n]);
                    acc[m][n] = P::mfma(fa[0][m], fb[0][n], acc[m][n]);
                }
        }
    };
#define VO_MMA(buf)                       \
    do {                                  \
        if constexpr (SPLIT) mma_split(buf); \
        else VO_MMA_SLICE(buf);           \
    } while (0)

    // Slice s: global loads into register set s&1 two slices ahead (with bf16 terms a slice's multiply
    // phase is shorter than the memory latency), LDS buffer s&1.  K is a multiple of 2*BK.
    const int slices = K / BK;
    VO_LOAD_SLICE(0, 0);
    VO_LOAD_SLICE(1, BK);
    VO_STORE_SLICE(0, 0);
    __syncthreads();
    for (int s = 0; s < slices; s += 2) {
        if (s + 2 < slices) VO_LOAD_SLICE(0, (s + 2) * BK);
        VO_MMA(0);
        VO_STORE_SLICE(1, 1);                 // slice s+1; the other buffer: nobody reads it in this phase
        __syncthreads();
        if (s + 3 < slices) VO_LOAD_SLICE(1, (s + 3) * BK);
        VO_MMA(1);
        if (s + 2 < slices) {
            VO_STORE_SLICE(0, 0);
            __syncthreads();
        }
    }
#undef VO_MMA
#undef VO_MMA_SLICE
#undef VO_LOAD_SLICE
#undef VO_STORE_SLICE

    // SPLIT: a multiplication by the rounded reciprocal instead of the float32 path's division -- the same bits when
    // sqrt(D) is a power of two (D = 256), one more rounding of 2^-24 otherwise, far inside three bf16 terms' 16 bits
    // F16: the two inverse powers of the operands' scaling ride along (exact)
    const float inv_divisor = (1.0f / divisor) * pow2f(-eA), unscale_b = pow2f(-eB);
    auto fin = [&](float v) {
        if constexpr (F16) return v * inv_divisor * unscale_b;
        else if constexpr (SPLIT) return v * inv_divisor;
        else return v / divisor;
    };
    // C/D layout of the 32x32 MFMA: column = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
    if ((N & 3) == 0 && (reinterpret_cast<uintptr_t>(Cg) & 15) == 0) {
        // Every tile passes through a 4 KiB LDS slot of its wave, after which a lane holds four consecutive
        // columns of one row: 16-byte stores, eight lanes per 128-byte segment, a quarter of the store
        // instructions (the kernel's bound is the store of the volume).
        __syncthreads();                                         // every wave is done with the operand tiles
        float* slot = smem + wave * 1024;
        const int q = lane >> 3, px4 = (lane & 7) * 4;
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    slot[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + lr] = fin(acc[m][n][e]);
                const int col = col0 + (wn * TN + n) * 32 + px4;
                const int rbase = row0 + (wm * TM + m) * 32 + q;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(slot + (q + 8 * j) * 32 + px4);
                    const int row = rbase + 8 * j;
                    if (row < M && col < N) {
                        f32x4* dst = reinterpret_cast<f32x4*>(Cg + (int64_t)row * N + col);
                        *dst = t;
                    }
                }
            }
        return;
    }
    // Per register the two wave halves each write 32 consecutive floats of one row.
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = col0 + (wn * TN + n) * 32 + lr;
            const int rbase = row0 + (wm * TM + m) * 32 + 4 * lh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = rbase + (e & 3) + 8 * (e >> 2);
                if (row < M && col < N) Cg[(int64_t)row * N + col] = fin(acc[m][n][e]);
            }
        }
}

// The large volume on rows split beforehand (blocks of 32 k): 256 x 256 tile, 8 waves of 64 rows x 128
// columns, one workgroup per CU walking a list of tiles.
//   * a wave reads 12 fragments per 24 matrix instructions (each A fragment feeds four column tiles, each B
//     fragment two row tiles), fetched one (k-step, column tile) step ahead of their use;
//   * K-slices of 32: one slice ahead in registers, two LDS buffers, one barrier per slice;
//   * the first slice of the NEXT tile is requested during the last slice of the current one, so its memory
//     latency passes under the matrix instructions and the store of the finished tile;
//   * every 32 x 32 block goes through a 4 KiB LDS slot of its wave: 16-byte stores of whole 128-byte segments.
constexpr int BIG = 256, BIG_ROWB = (32 + PAD) * 4, BIG_BUF = 2 * BIG * BIG_ROWB;
// F16: float16 pieces of maps scaled per image (split_rows_kernel<true>); amaxA[g] / amaxB[g] undo the scaling.
template <bool F16>
__global__ __launch_bounds__(512, 1) void corr_presplit_big_kernel(
    const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, float* __restrict__ C, int M, int N, int K,
    int64_t strideA, int64_t strideB, int64_t strideC, int tiles_m, int tiles_n, unsigned nwork, float inv_divisor,
    const unsigned* __restrict__ amaxA, const unsigned* __restrict__ amaxB) {
    using P = std::conditional_t<F16, PiecesF16, PiecesBf16>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_big[];      // [2][A rows | B rows][BIG_ROWB]
    constexpr int GROUP_M = 8;
    const int rowb = K * 4;                                          // bytes of one feature row
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;

    // Work list: one contiguous range per XCD (workgroups are dealt round robin over the 8 XCDs), the
    // workgroups of an XCD taking its items in turn, so that at any time they work on neighbouring tiles.
    const unsigned xcd = blockIdx.x & 7u, per_xcd = gridDim.x >> 3;              // the host launches a multiple of 8
    const unsigned q8 = nwork >> 3, r8 = nwork & 7u;
    const unsigned range0 = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned range1 = range0 + q8 + (xcd < r8 ? 1u : 0u);

    int g, row0, col0;
    int goff_a[4], goff_b[4], loff[4];
    __amdgpu_buffer_rsrc_t ars, brs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 512;
        loff[i] = (f >> 3) * BIG_ROWB + (f & 7) * 16;
    }
    // the tile of work item w: row tiles in groups of GROUP_M, a group sweeping the column tiles with the
    // row tile fastest (neighbouring items share B tiles and a few A tiles through L2)
    auto setup = [&](unsigned w) {
        const unsigned per_g = (unsigned)tiles_n * tiles_m;
        g = w / per_g;
        const unsigned w_in = w - g * per_g;
        const unsigned group_sz = GROUP_M * tiles_n;
        const unsigned grp = w_in / group_sz;
        const int rows_in_grp = min(GROUP_M, tiles_m - (int)grp * GROUP_M);
        const unsigned w_grp = w_in - grp * group_sz;
        row0 = (grp * GROUP_M + w_grp % rows_in_grp) * BIG;
        col0 = (w_grp / rows_in_grp) * BIG;
        ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(A + g * strideA * 4), 0, M * rowb, 0x00020000);
        brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(B + g * strideB * 4), 0, N * rowb, 0x00020000);
        // staging: 16-byte piece f -> (row f >> 3, piece f & 7); rows past the edge re-read the last valid row
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + i * 512, r = f >> 3, q = f & 7;
            goff_a[i] = min(row0 + r, M - 1) * rowb + q * 16;
            goff_b[i] = min(col0 + r, N - 1) * rowb + q * 16;
        }
    };
    u32x4 ra[4], rb[4];
    auto load_slice = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(ars, goff_a[i], s * 128, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(brs, goff_b[i], s * 128, 0);
    };
    auto store_slice = [&](int buf) {
        unsigned char* base = lds_big + buf * BIG_BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(base + loff[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(base + BIG * BIG_ROWB + loff[i]) = rb[i];
    };

    const int a_frag = (wm * 64 + lr) * BIG_ROWB + 16 * lh;                      // + m*32 rows, + part*64, + kg*32
    const int b_frag = BIG * BIG_ROWB + (wn * 128 + lr) * BIG_ROWB + 16 * lh;
    const int slices = K / 32;                                                   // even (the host checks K % 64)
    // transposition slots (half a 32 x 32 block each) behind the two operand buffers: the buffers receive the
    // next tile's slices while other waves may still be storing
    float* slot = reinterpret_cast<float*>(lds_big + 2 * BIG_BUF) + wave * 512;
    const int q = lane >> 3, px4 = (lane & 7) * 4;

    unsigned w = range0 + (blockIdx.x >> 3);
    if (w >= range1) return;
    setup(w);
    load_slice(0);
    store_slice(0);
    load_slice(1);
    __syncthreads();

    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;

    // One body for every slice s (loads and LDS stores outside any branch, so that their waits stay counted):
    // slice s+1, requested a whole slice earlier, goes from the registers into the buffer that was read
    // last (everyone has passed the barrier behind those reads), slice s+2 is requested into the same
    // registers, slice s is multiplied.  Slices past a tile's end are the first two of the workgroup's next
    // tile (on its last tile: of that tile again -- loaded, stored, never used).
    auto slice_body = [&](int s) {
        store_slice((s + 1) & 1);
        if (s + 2 == slices) setup(w + per_xcd < range1 ? w + per_xcd : w);
        load_slice(s + 2 < slices ? s + 2 : s + 2 - slices);
        const unsigned char* base = lds_big + (s & 1) * BIG_BUF;
        auto read_a = [&](int kg, int part, int m) {
            return *reinterpret_cast<const u32x4*>(base + a_frag + m * 32 * BIG_ROWB + part * 64 + kg * 32);
        };
        auto read_b = [&](int kg, int part, int n) {
            return *reinterpret_cast<const u32x4*>(base + b_frag + n * 32 * BIG_ROWB + part * 64 + kg * 32);
        };
        // 8 steps of (k-step, column tile), 6 matrix instructions each; a step issues the reads of the next one
        u32x4 fa[2][2][2], fb[2][2];
#pragma unroll
        for (int part = 0; part < 2; ++part) {
#pragma unroll
            for (int m = 0; m < 2; ++m) fa[0][part][m] = read_a(0, part, m);
            fb[0][part] = read_b(0, part, 0);
        }
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int kg = st >> 2, n = st & 3, cb = st & 1;
            if (st + 1 < 8) {
                fb[cb ^ 1][0] = read_b((st + 1) >> 2, 0, (st + 1) & 3);
                fb[cb ^ 1][1] = read_b((st + 1) >> 2, 1, (st + 1) & 3);
            }
            if (kg == 0) fa[1][n >> 1][n & 1] = read_a(1, n >> 1, n & 1);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                // the small terms first, so that they are not rounded away against the large one
                acc[m][n] = P::mfma(fa[kg][1][m], fb[cb][0], acc[m][n]);
                acc[m][n] = P::mfma(fa[kg][0][m], fb[cb][1], acc[m][n]);
                acc[m][n] = P::mfma(fa[kg][0][m], fb[cb][0], acc[m][n]);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };

    // The loop starts with the store of the PREVIOUS tile (of nothing on the first pass: a descriptor of zero
    // bytes drops every store), so that the multiplication of a tile is always entered from behind the same
    // 32 branch-free stores: the second slice of the tile, requested before them, is then waited for with a
    // count that leaves those stores in flight, and they drain under the first slice's matrix instructions.
    int st_g = 0, st_row0 = 0, st_col0 = 0, st_bytes = 0;
    float st_k1 = inv_divisor, st_k2 = 1.0f;                         // F16: with the group's two inverse powers of two
    for (;;) {
        const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(C + st_g * strideC, 0, st_bytes, 0x00020000);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int half = 0; half < 2; ++half) {               // rows 0..15 of the block, then rows 16..31
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if constexpr (F16) slot[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + lr] = acc[m][n][8 * half + e] * st_k1 * st_k2;
                        else slot[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + lr] = acc[m][n][8 * half + e] * inv_divisor;
                        acc[m][n][8 * half + e] = 0.0f;
                    }
                    const int col = st_col0 + wn * 128 + n * 32 + px4;
                    const int rbase = st_row0 + wm * 64 + m * 32 + 16 * half + q;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const u32x4 t = *reinterpret_cast<const u32x4*>(slot + (q + 8 * j) * 32 + px4);
                        const int row = rbase + 8 * j;
                        __builtin_amdgcn_raw_buffer_store_b128(t, crs, (row < M && col < N) ? (row * N + col) * 4 : (int)0x80000000, 0, 2);   // nontemporal: the volume streams out, the feature tiles stay in L2
                    }
                }
        if (w >= range1) break;
        st_g = g; st_row0 = row0; st_col0 = col0; st_bytes = M * N * 4;
        if constexpr (F16) {
            st_k1 = inv_divisor * pow2f(-feature_exp(amaxA[g]));
            st_k2 = pow2f(-feature_exp(amaxB[g]));
        }
        slice_body(0);
        for (int s = 1; s < slices; ++s) slice_body(s);
        w += per_xcd;
    }
}

// Rows of K float32 -> the same 4K bytes per row holding bf16 pieces in blocks of BLK k:
//   [BLK high parts | BLK low parts] per block, high = bf16(x), low = bf16(x - high).
// A K-slice of BLK k of one row is then 4*BLK contiguous bytes, laid out as the split kernels keep it in LDS.
// F16: float16 pieces of x * 2^e, e from the largest |x| of the row's image (amax[image], quads_per_image quads each).
template <bool F16>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, unsigned char* __restrict__ out,
                                                         int64_t quads, int blk_quads, const unsigned* __restrict__ amax,
                                                         int64_t quads_per_image) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // four consecutive k of one row
    if (i >= quads) return;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    unsigned h0, h1, l0, l1;
    if constexpr (F16) {
        const float s = pow2f(feature_exp(amax[i / quads_per_image]));
        cut_f16_scaled(v.x * s, v.y * s, h0, l0);
        cut_f16_scaled(v.z * s, v.w * s, h1, l1);
    } else {
        h0 = pack_bf16(v.x, v.y), h1 = pack_bf16(v.z, v.w);
        l0 = pack_bf16(v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u));
        l1 = pack_bf16(v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u));
    }
    const int64_t block = i / blk_quads;
    const int q = (int)(i - block * blk_quads);
    unsigned char* o = out + block * blk_quads * 16;
    *reinterpret_cast<uint2*>(o + q * 8) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(o + blk_quads * 8 + q * 8) = make_uint2(l0, l1);
}

// Largest |x| of each image as float bits (non-negative floats order like unsigned integers; a NaN's bits exceed
// every number's, so a map with a NaN reports one).  amax is zeroed beforehand (vo_zero_words); one atomic per
// workgroup, a few dozen workgroups per image.
__global__ __launch_bounds__(256) void rows_amax_kernel(const float* __restrict__ x, unsigned* __restrict__ amax, int64_t quads_per_image) {
    __shared__ unsigned part[4];
    const float4* p = reinterpret_cast<const float4*>(x) + (int64_t)blockIdx.y * quads_per_image;
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads_per_image; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu,
                max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(part[0], part[1]), max(part[2], part[3]));
        if (m) atomicMax(amax + blockIdx.y, m);
    }
}

// the float16 pieces' per-image maxima: amaxA / amaxB (device, float bits), per_image groups share one entry
struct Amax { const unsigned* a = nullptr; const unsigned* b = nullptr; int per_image = 1; };

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool SPLIT, bool PRE = false, bool F16 = false>
int launch_corr(const float* A, const float* B, float* C, int groups, int M, int N, int K, int64_t sA,
                int64_t sB, int64_t sC, float divisor, hipStream_t stream, const char* name, Amax am = Amax{}) {
    const int tiles_m = (int)vo_ceil_div(M, BM), tiles_n = (int)vo_ceil_div(N, BN);
    const int64_t nwg = (int64_t)tiles_m * tiles_n * groups;
    if (nwg > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "%s: %lld workgroups", name, (long long)nwg);
    constexpr size_t lds = 2 * (size_t)(BM + BN) * (BK + PAD) * sizeof(float);
    static_assert(lds >= (size_t)WAVES_M * WAVES_N * 4096, "the store path's transposition slots do not fit");
    auto kern = corr_abt_kernel<BM, BN, BK, WAVES_M, WAVES_N, SPLIT, PRE, F16>;
    if (lds > 48 * 1024) {
        // the attribute is kept per device and setting it is a cheap host call: set it on every
        // launch rather than remember it in a flag that another device or thread would not share
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "%s: LDS attribute: %s", name, hipGetErrorString(e));
    }
    kern<<<dim3((unsigned)nwg), dim3(64 * WAVES_M * WAVES_N), lds, stream>>>(A, B, C, M, N, K, sA, sB, sC, tiles_m,
                                                                             tiles_n, (unsigned)nwg, divisor, am.a, am.b, am.per_image);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e));
    return VO_OK;
}

template <bool F16>
int launch_presplit_big(const float* A, const float* B, float* C, int groups, int M, int N, int K, int64_t sA, int64_t sB,
                        int64_t sC, float divisor, hipStream_t stream, const char* name, Amax am) {
    const int tiles_m = (int)vo_ceil_div(M, BIG), tiles_n = (int)vo_ceil_div(N, BIG);
    const int64_t nwork = (int64_t)tiles_m * tiles_n * groups;
    if (nwork > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "%s: %lld tiles", name, (long long)nwork);
    constexpr int lds = 2 * BIG_BUF + 8 * 2048;
    static_assert(lds <= 160 * 1024, "the operand buffers and the slots exceed the LDS of a CU");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(corr_presplit_big_kernel<F16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "%s: LDS attribute: %s", name, hipGetErrorString(e));
    // one workgroup per CU (a multiple of 8, one share per XCD); a workgroup with no item returns at once
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    const unsigned grid = (unsigned)std::min<int64_t>((cus + 7) / 8 * 8, (nwork + 7) / 8 * 8);
    corr_presplit_big_kernel<F16><<<dim3(grid), dim3(512), lds, stream>>>(
        reinterpret_cast<const unsigned char*>(A), reinterpret_cast<const unsigned char*>(B), C, M, N, K, sA, sB, sC, tiles_m,
        tiles_n, (unsigned)nwork, 1.0f / divisor, am.a, am.b);
    e = hipGetLastError();
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e));
    return VO_OK;
}

// mode 0: float32 products; 1: float32 rows split into 16-bit pieces inside the kernel; 2: rows already split
// (split_rows_kernel's format, blocks of 32 k for the large tile and of 16 k for the small one).
// pieces (modes 1, 2): VO_PIECES_BF16, or VO_PIECES_F16 with the per-image maxima in `am`.
int corr_abt(const float* A, const float* B, float* C, int groups, int M, int N, int K, int64_t sA, int64_t sB,
             int64_t sC, hipStream_t stream, const char* name, int mode = 0, int force_small = -1, int pieces = VO_PIECES_BF16,
             Amax am = Amax{}) {
    if (!A || !B || !C) return vo_fail(VO_ERR_INVALID_ARG, "%s: null pointer", name);
    if (groups <= 0 || M <= 0 || N <= 0 || K <= 0)
        return vo_fail(VO_ERR_INVALID_ARG, "%s: non-positive size (groups=%d M=%d N=%d K=%d)", name, groups, M, N, K);
    if (K % 32 != 0) return vo_fail(VO_ERR_UNSUPPORTED, "%s: feature dim %d is not a multiple of 32", name, K);
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15)
        return vo_fail(VO_ERR_INVALID_ARG, "%s: feature maps must be 16-byte aligned", name);
    if (pieces != VO_PIECES_BF16 && pieces != VO_PIECES_F16) return vo_fail(VO_ERR_INVALID_ARG, "%s: pieces must be VO_PIECES_BF16 or VO_PIECES_F16", name);
    const bool f16 = mode != 0 && pieces == VO_PIECES_F16;
    if (f16 && (!am.a || !am.b || am.per_image <= 0 || groups % am.per_image != 0))
        return vo_fail(VO_ERR_INVALID_ARG, "%s: float16 pieces need the per-image maxima of both feature maps", name);
    const float divisor = sqrtf((float)K);      // corr / sqrt(dim), divided as the reference does
    // Small matrices (one image row of RAFT-Stereo) get one 160-wide tile per
    // workgroup; the big flow volume gets 128x128 tiles.
    // the kernels take K in pairs of slices: 2 x 16 for the small tile, 2 x 32 for the large one
    bool small = (M <= 512 && N <= 512) || K % 64 != 0;
    if (force_small >= 0) {
        if (!force_small && K % 64 != 0) return vo_fail(VO_ERR_UNSUPPORTED, "%s: feature dim %d is not a multiple of 64", name, K);
        small = force_small != 0;
    }
    if (mode == 2 && !small && (N & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && (sC & 3) == 0 &&
        (int64_t)std::max(M, N) * K * 4 < 0x7fffffffLL && (int64_t)M * N * 4 < 0x7fffffffLL && (!f16 || am.per_image == 1)) {
        return f16 ? launch_presplit_big<true>(A, B, C, groups, M, N, K, sA, sB, sC, divisor, stream, name, am)
                   : launch_presplit_big<false>(A, B, C, groups, M, N, K, sA, sB, sC, divisor, stream, name, am);
    }
#define VO_CORR_LAUNCH(PRE_)                                                                                                            \
    do {                                                                                                                                \
        if (small)                                                                                                                      \
            return f16 ? launch_corr<160, 160, 16, 1, 5, true, PRE_, true>(A, B, C, groups, M, N, K, sA, sB, sC, divisor, stream, name, am) \
                       : launch_corr<160, 160, 16, 1, 5, true, PRE_>(A, B, C, groups, M, N, K, sA, sB, sC, divisor, stream, name);      \
        return f16 ? launch_corr<128, 128, 32, 2, 2, true, PRE_, true>(A, B, C, groups, M, N, K, sA, sB, sC, divisor, stream, name, am)     \
                   : launch_corr<128, 128, 32, 2, 2, true, PRE_>(A, B, C, groups, M, N, K, sA, sB, sC, divisor, stream, name);          \
    } while (0)
    if (mode == 2) VO_CORR_LAUNCH(true);
    if (mode == 1) VO_CORR_LAUNCH(false);
#undef VO_CORR_LAUNCH
    if (small)
        return launch_corr<160, 160, 16, 1, 5, false>(A, B, C, groups, M, N, K, sA, sB, sC, divisor, stream, name);
    return launch_corr<128, 128, 32, 2, 2, false>(A, B, C, groups, M, N, K, sA, sB, sC, divisor, stream, name);
}

}  // namespace

extern "C" int vo_corr_volume_2d(const float* fmap1, const float* fmap2, float* corr, int batch, int H, int W,
                                 int D, void* stream) {
    VO_REQUIRE(H > 0 && W > 0, "vo_corr_volume_2d: bad size H=%d W=%d", H, W);
    const int64_t n = (int64_t)H * W;
    VO_REQUIRE(n <= (1 << 20), "vo_corr_volume_2d: %lld pixels per feature map is beyond the supported 2^20", (long long)n);
    return corr_abt(fmap1, fmap2, corr, batch, (int)n, (int)n, D, n * D, n * D, n * n, vo_stream(stream),
                    "vo_corr_volume_2d");
}

extern "C" int vo_corr_volume_1d(const float* fmap1, const float* fmap2, float* corr, int batch, int H, int W1,
                                 int W2, int D, void* stream) {
    VO_REQUIRE(batch > 0 && H > 0, "vo_corr_volume_1d: bad size batch=%d H=%d", batch, H);
    VO_REQUIRE((int64_t)batch * H <= 0x7fffffffLL, "vo_corr_volume_1d: batch*H too large");
    return corr_abt(fmap1, fmap2, corr, batch * H, W1, W2, D, (int64_t)W1 * D, (int64_t)W2 * D, (int64_t)W1 * W2,
                    vo_stream(stream), "vo_corr_volume_1d");
}

extern "C" int vo_corr_volume_2d_bf16x3(const float* fmap1, const float* fmap2, float* corr, int batch, int H, int W,
                                        int D, int pieces, const unsigned* amax1, const unsigned* amax2, void* stream) {
    VO_REQUIRE(H > 0 && W > 0, "vo_corr_volume_2d_bf16x3: bad size H=%d W=%d", H, W);
    const int64_t n = (int64_t)H * W;
    VO_REQUIRE(n <= (1 << 20), "vo_corr_volume_2d_bf16x3: %lld pixels per feature map is beyond the supported 2^20", (long long)n);
    return corr_abt(fmap1, fmap2, corr, batch, (int)n, (int)n, D, n * D, n * D, n * n, vo_stream(stream),
                    "vo_corr_volume_2d_bf16x3", 1, -1, pieces, Amax{amax1, amax2, 1});
}

extern "C" int vo_corr_volume_1d_bf16x3(const float* fmap1, const float* fmap2, float* corr, int batch, int H, int W1,
                                        int W2, int D, int pieces, const unsigned* amax1, const unsigned* amax2, void* stream) {
    VO_REQUIRE(batch > 0 && H > 0, "vo_corr_volume_1d_bf16x3: bad size batch=%d H=%d", batch, H);
    VO_REQUIRE((int64_t)batch * H <= 0x7fffffffLL, "vo_corr_volume_1d_bf16x3: batch*H too large");
    return corr_abt(fmap1, fmap2, corr, batch * H, W1, W2, D, (int64_t)W1 * D, (int64_t)W2 * D, (int64_t)W1 * W2,
                    vo_stream(stream), "vo_corr_volume_1d_bf16x3", 1, -1, pieces, Amax{amax1, amax2, H});
}

extern "C" int vo_feature_rows_amax(const float* x, unsigned* amax, int images, int64_t elems_per_image, void* stream) {
    VO_REQUIRE(x && amax, "vo_feature_rows_amax: null pointer");
    VO_REQUIRE(images > 0 && images <= 65535 && elems_per_image > 0, "vo_feature_rows_amax: bad sizes images=%d elems=%lld", images, (long long)elems_per_image);
    VO_REQUIRE(elems_per_image % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "vo_feature_rows_amax: images of whole, 16-byte aligned groups of four values");
    hipStream_t st = vo_stream(stream);
    hipError_t e = vo_zero_words(amax, images, st);
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "vo_feature_rows_amax: %s", hipGetErrorString(e));
    const int64_t quads = elems_per_image / 4;
    const unsigned segs = (unsigned)std::min<int64_t>(vo_ceil_div(quads, 256 * 32), 64);          // 32 KiB or more per workgroup
    rows_amax_kernel<<<dim3(segs, (unsigned)images), dim3(256), 0, st>>>(x, amax, quads);
    e = hipGetLastError();
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "vo_feature_rows_amax: %s", hipGetErrorString(e));
    return VO_OK;
}

extern "C" int vo_split_feature_rows(const float* x, void* out, int64_t rows, int D, int block, int pieces, const unsigned* amax,
                                     int64_t rows_per_image, void* stream) {
    VO_REQUIRE(x && out, "vo_split_feature_rows: null pointer");
    VO_REQUIRE(rows > 0 && D > 0, "vo_split_feature_rows: bad sizes rows=%lld D=%d", (long long)rows, D);
    VO_REQUIRE(block == 16 || block == 32, "vo_split_feature_rows: block must be 16 or 32, got %d", block);
    VO_REQUIRE(D % block == 0, "vo_split_feature_rows: D=%d is not a multiple of the block %d", D, block);
    VO_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "vo_split_feature_rows: tensors must be 16-byte aligned");
    VO_REQUIRE(static_cast<const void*>(x) != out, "vo_split_feature_rows: not in place");
    VO_REQUIRE(pieces == VO_PIECES_BF16 || pieces == VO_PIECES_F16, "vo_split_feature_rows: pieces must be VO_PIECES_BF16 or VO_PIECES_F16");
    const int64_t quads = rows * D / 4;
    const dim3 grid((unsigned)vo_ceil_div(quads, 256));
    if (pieces == VO_PIECES_F16) {
        VO_REQUIRE(amax && rows_per_image > 0 && rows % rows_per_image == 0,
                   "vo_split_feature_rows: float16 pieces need the per-image maxima and rows in whole images (rows=%lld, per image %lld)",
                   (long long)rows, (long long)rows_per_image);
        split_rows_kernel<true><<<grid, dim3(256), 0, vo_stream(stream)>>>(x, static_cast<unsigned char*>(out), quads, block / 4, amax, rows_per_image * D / 4);
    } else {
        split_rows_kernel<false><<<grid, dim3(256), 0, vo_stream(stream)>>>(x, static_cast<unsigned char*>(out), quads, block / 4, nullptr, 1);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "vo_split_feature_rows: %s", hipGetErrorString(e));
    return VO_OK;
}

extern "C" int vo_corr_volume_2d_presplit(const void* fmap1, const void* fmap2, float* corr, int batch, int H, int W,
                                          int D, int pieces, const unsigned* amax1, const unsigned* amax2, void* stream) {
    VO_REQUIRE(H > 0 && W > 0, "vo_corr_volume_2d_presplit: bad size H=%d W=%d", H, W);
    const int64_t n = (int64_t)H * W;
    VO_REQUIRE(n <= (1 << 20), "vo_corr_volume_2d_presplit: %lld pixels per feature map is beyond the supported 2^20", (long long)n);
    return corr_abt(static_cast<const float*>(fmap1), static_cast<const float*>(fmap2), corr, batch, (int)n, (int)n, D, n * D,
                    n * D, n * n, vo_stream(stream), "vo_corr_volume_2d_presplit", 2, 0, pieces, Amax{amax1, amax2, 1});
}

extern "C" int vo_corr_volume_1d_presplit(const void* fmap1, const void* fmap2, float* corr, int batch, int H, int W1,
                                          int W2, int D, int pieces, const unsigned* amax1, const unsigned* amax2, void* stream) {
    VO_REQUIRE(batch > 0 && H > 0, "vo_corr_volume_1d_presplit: bad size batch=%d H=%d", batch, H);
    VO_REQUIRE((int64_t)batch * H <= 0x7fffffffLL, "vo_corr_volume_1d_presplit: batch*H too large");
    return corr_abt(static_cast<const float*>(fmap1), static_cast<const float*>(fmap2), corr, batch * H, W1, W2, D,
                    (int64_t)W1 * D, (int64_t)W2 * D, (int64_t)W1 * W2, vo_stream(stream), "vo_corr_volume_1d_presplit", 2, 1, pieces,
                    Amax{amax1, amax2, H});
}
