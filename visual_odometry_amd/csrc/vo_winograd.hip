// 3x3 stride-1 convolution (padding 1) in float32 as Winograd F(2x2, 3x3) with the
// sixteen transformed-domain products on the matrix cores.
//
// Both networks spend most of their time in 3x3 convolutions of the encoders and update
// blocks (flow/flow_map_RAFT.py:42, stereo/stereo_disparity_RAFT.py:46 build them).  The
// Winograd form needs 16 multiplies per 2x2 output tile and channel pair instead of 36;
// for each of the 16 tile positions xi the reduction over input channels is a GEMM
//     M_xi[k, t] = sum_c U_xi[k, c] * V_xi[c, t]        (k: output channel, t: tile)
// which runs on v_mfma_f32_32x32x2_f32.  One workgroup (8 waves) owns an 8x8 block of
// tiles (16x16 output pixels) and 64 output channels; wave w owns positions 2w, 2w+1.
// Input channels stream through in chunks of 8: the raw 18x18 patch of a chunk lands in
// LDS, every lane transforms one (channel, tile) into its 16 positions (B^T d B), the
// waves multiply, and after the last chunk the accumulators go through LDS once more for
// the output transform (A^T M A).  Transformed filters are precomputed in the exact
// register order of the MFMA A operand, so they are read straight from L2, coalesced.
#include "vo_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CB = 8;                  // input channels per chunk
constexpr int KB = 64;                 // output channels per workgroup
constexpr int RAW_W = 20;              // padded row length of the raw patch (18 used)
constexpr int RAW_C = 18 * RAW_W;      // floats per channel of the raw patch
constexpr int V_FLOATS = 16 * CB * 64; // transformed chunk: [xi][c][tile]
constexpr int RAW_FLOATS = CB * RAW_C;
constexpr int RAW_PER_LANE = (CB * 18 * 18 + 511) / 512;   // 6
constexpr int MAX_NORM_C = 512;        // input channels supported by the normalise-on-load variant

// U = G g G^T for every (k, c), written as
//   Ug[kb][chunk][xi][i][lane][ks]:  k = kb*64 + i*32 + (lane & 31),  c = chunk*8 + 2*ks + (lane >> 5)
// (zero where k >= K): the four A fragments a lane needs for one (position, k-tile, chunk)
// are one 16-byte load.
__global__ __launch_bounds__(256) void filter_kernel(const float* __restrict__ g, float* __restrict__ Ug, int K, int C,
                                                     int Kpad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Kpad * C) return;
    const int k = (int)(idx / C), c = (int)(idx % C);
    float w[3][3];
    for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s) w[r][s] = k < K ? g[(((int64_t)k * C + c) * 3 + r) * 3 + s] : 0.0f;
    float t[4][3];
    for (int s = 0; s < 3; ++s) {
        t[0][s] = w[0][s];
        t[1][s] = 0.5f * (w[0][s] + w[1][s] + w[2][s]);
        t[2][s] = 0.5f * (w[0][s] - w[1][s] + w[2][s]);
        t[3][s] = w[2][s];
    }
    const int kb = k >> 6, i = (k >> 5) & 1, kl = k & 31, chunk = c >> 3, ks = (c & 7) >> 1, kk = c & 1;
    const int nchunks = C / CB;
    for (int a = 0; a < 4; ++a) {
        const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]),
                    u3 = t[a][2];
        const float u[4] = {u0, u1, u2, u3};
        for (int b = 0; b < 4; ++b) {
            const int xi = 4 * a + b;
            const int64_t o = ((((((int64_t)kb * nchunks + chunk) * 16 + xi) * 2 + i) * 64) + kk * 32 + kl) * 4 + ks;
            Ug[o] = u[b];
        }
    }
}

// Optional per-channel epilogue applied at the store: v = y*scale[k] + shift[k]; ReLU;
// then the residual join max(v + skip, 0).  Null pointers / relu = 0 switch stages off.
struct Epilogue { const float* scale; const float* shift; const float* skip; int relu; int64_t y_batch_stride; double* stats; };

// Optional normalisation of the INPUT while it is staged: x^ = relu((x - mean) * rstd) per (image,
// channel), mean / rstd from the plane sums a producing launch left in `stats` (N,C,2): the
// instance norm + ReLU between two convolutions without a pass of its own.
struct InputNorm { const double* stats; float eps; };

template <bool NORM_IN>
__global__ __launch_bounds__(512) void wino_kernel(const float* __restrict__ x, const float* __restrict__ Ug,
                                                   float* __restrict__ y, Epilogue ep, InputNorm in, int C, int H, int W,
                                                   int K, int blocks_x, int blocks_y, int kblocks) {
    // V and the raw patch are double-buffered (chunk c+1 is transformed while chunk c is
    // multiplied); after the loop the first 64 KB hold M for the output transform.
    __shared__ __align__(16) float lds[2 * V_FLOATS + 2 * RAW_FLOATS + (NORM_IN ? 2 * MAX_NORM_C : 0)];
    float* Vs = lds;
    float* raw = lds + 2 * V_FLOATS;
    float2* nrm = reinterpret_cast<float2*>(lds + 2 * V_FLOATS + 2 * RAW_FLOATS);     // (scale, shift) per input channel

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The dispatcher deals consecutive workgroups round robin over the 8 XCDs.  Give each XCD
    // one contiguous range of the (image, block row, block column, channel block) list, so
    // that blocks sharing halo rows and 128-byte lines hit in the same L2.
    int bid;
    {
        const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, q = nwg >> 3, r = nwg & 7u;
        bid = (int)(((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3));
    }
    const int kb = bid % kblocks; bid /= kblocks;
    const int bx = bid % blocks_x; bid /= blocks_x;
    const int by = bid % blocks_y;
    const int n = bid / blocks_y;
    const int nchunks = C / CB;
    const int y0 = by * 16 - 1, x0 = bx * 16 - 1;                // top-left of the raw patch (padding 1)
    const float* xn = x + (int64_t)n * C * H * W;

    // where this lane's raw-patch elements live (same for every chunk)
    int raw_lds[RAW_PER_LANE];
    int raw_glb[RAW_PER_LANE];
    unsigned raw_ch = 0;                                         // channel within the chunk, 3 bits per element
#pragma unroll
    for (int i = 0; i < RAW_PER_LANE; ++i) {
        const int idx = tid + i * 512;
        const int c = idx / 324, rem = idx - c * 324, r = rem / 18, col = rem - r * 18;
        const int gy = y0 + r, gx = x0 + col;
        const bool ok = idx < CB * 324 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        raw_lds[i] = idx < CB * 324 ? c * RAW_C + r * RAW_W + col : -1;
        raw_glb[i] = ok ? (c * H + gy) * W + gx : -1;
        raw_ch |= (unsigned)(c & 7) << (3 * i);
    }
    float rv[RAW_PER_LANE];
    auto load_raw = [&](int chunk) {
        const float* xc = xn + (int64_t)chunk * CB * H * W;
#pragma unroll
        for (int i = 0; i < RAW_PER_LANE; ++i) rv[i] = raw_glb[i] >= 0 ? xc[raw_glb[i]] : 0.0f;
    };
    if (NORM_IN) {
        const double hw = (double)H * (double)W;
        for (int c = tid; c < C; c += 512) {
            const double* st = in.stats + ((int64_t)n * C + c) * 2;
            const double mean = st[0] / hw;
            double var = st[1] / hw - mean * mean;
            if (var < 0.0) var = 0.0;
            const float r = (float)(1.0 / sqrt(var + (double)in.eps));
            nrm[c] = make_float2(r, -(float)mean * r);
        }
        __syncthreads();
    }
    auto store_raw = [&](int buf, int chunk) {
        float* rb = raw + buf * RAW_FLOATS;
#pragma unroll
        for (int i = 0; i < RAW_PER_LANE; ++i)
            if (raw_lds[i] >= 0) {
                float v = rv[i];
                if (NORM_IN) {                                   // padding stays zero: it pads the NORMALISED tensor
                    const float2 ss = nrm[chunk * CB + ((raw_ch >> (3 * i)) & 7)];
                    v = raw_glb[i] >= 0 ? fmaxf(v * ss.x + ss.y, 0.0f) : 0.0f;
                }
                rb[raw_lds[i]] = v;
            }
    };
    const float* Uw = Ug + (((int64_t)kb * nchunks) * 16 + 2 * wave) * 512 + lane * 4;  // + chunk*8192 + (xl*2+i)*256
    float a_cur[16], a_nxt[16];                                  // [(xl*2+i)*4 + ks]
    auto load_a = [&](int chunk, float* a) {
        const float* p = Uw + (int64_t)chunk * 8192;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(p + j * 256);
            a[4 * j] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
        }
    };

    f32x16 acc[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][i][j][e] = 0.0f;

    const int tc = tid >> 6, tt = tid & 63, tty = tt >> 3, ttx = tt & 7;       // transform role: (channel, tile)
    const int kk = lane >> 5, nn = lane & 31;
    auto transform = [&](int buf) {                              // B^T d B of one (channel, tile)
        const float* p = raw + buf * RAW_FLOATS + tc * RAW_C + (2 * tty) * RAW_W + 2 * ttx;
        float d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float2 lo = *reinterpret_cast<const float2*>(p + r * RAW_W);
            const float2 hi = *reinterpret_cast<const float2*>(p + r * RAW_W + 2);
            d[r][0] = lo.x; d[r][1] = lo.y; d[r][2] = hi.x; d[r][3] = hi.y;
        }
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j];
            t[1][j] = d[1][j] + d[2][j];
            t[2][j] = d[2][j] - d[1][j];
            t[3][j] = d[1][j] - d[3][j];
        }
        float* vo = Vs + buf * V_FLOATS + tc * 64 + tt;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            vo[(4 * a + 0) * (CB * 64)] = t[a][0] - t[a][2];
            vo[(4 * a + 1) * (CB * 64)] = t[a][1] + t[a][2];
            vo[(4 * a + 2) * (CB * 64)] = t[a][2] - t[a][1];
            vo[(4 * a + 3) * (CB * 64)] = t[a][1] - t[a][3];
        }
    };
    auto multiply = [&](int buf) {
#pragma unroll
        for (int xl = 0; xl < 2; ++xl) {
            const float* vb = Vs + buf * V_FLOATS + (2 * wave + xl) * (CB * 64) + kk * 64 + 2 * nn;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float2 b = *reinterpret_cast<const float2*>(vb + ks * 128);
                const float a0 = a_cur[(xl * 2 + 0) * 4 + ks], a1 = a_cur[(xl * 2 + 1) * 4 + ks];
                acc[xl][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b.x, acc[xl][0][0], 0, 0, 0);
                acc[xl][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b.y, acc[xl][0][1], 0, 0, 0);
                acc[xl][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b.x, acc[xl][1][0], 0, 0, 0);
                acc[xl][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b.y, acc[xl][1][1], 0, 0, 0);
            }
        }
    };

    load_raw(0);
    load_a(0, a_nxt);
    store_raw(0, 0);
    __syncthreads();
    if (nchunks > 1) load_raw(1);
    transform(0);
    for (int chunk = 0; chunk + 1 < nchunks; ++chunk) {
        const int cur = chunk & 1;
#pragma unroll
        for (int j = 0; j < 16; ++j) a_cur[j] = a_nxt[j];
        store_raw(cur ^ 1, chunk + 1);                           // chunk+1, prefetched one iteration ago
        __syncthreads();                                         // V[cur] complete, raw[cur^1] visible
        load_a(chunk + 1, a_nxt);
        if (chunk + 2 < nchunks) load_raw(chunk + 2);
        // chunk+1 -> V[cur^1] and the MFMAs on V[cur] are independent: half of the waves take
        // them in the opposite order, so LDS/VALU work and matrix work overlap on a SIMD
        if (wave < 4) { transform(cur ^ 1); multiply(cur); }
        else          { multiply(cur); transform(cur ^ 1); }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) a_cur[j] = a_nxt[j];
    __syncthreads();
    multiply((nchunks - 1) & 1);

    // output transform, 16 output channels per round: M[xi][kq][tile] through LDS
    float* yn = y + (int64_t)n * ep.y_batch_stride;              // the output may be a channel slice of a wider tensor
    const bool quad_store = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (ep.y_batch_stride & 3) == 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        __syncthreads();                                         // previous readers of lds are done
#pragma unroll
        for (int xl = 0; xl < 2; ++xl)
#pragma unroll
            for (int eh = 0; eh < 8; ++eh) {
                const int e = (q & 1) * 8 + eh;
                const int kq = (e & 3) + 8 * ((e >> 2) & 1) + 4 * kk;
                float2 v;
                v.x = acc[xl][q >> 1][0][e];
                v.y = acc[xl][q >> 1][1][e];
                *reinterpret_cast<float2*>(lds + (2 * wave + xl) * 1024 + kq * 64 + 2 * nn) = v;
            }
        __syncthreads();
        {   // one (channel, pair of horizontally adjacent tiles) per lane: 16 bytes per output row
            const int tp = tid & 31, kq = tid >> 5;
            const int t = 2 * tp;
            const int k = kb * KB + q * 16 + kq;
            float m[2][16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                const float2 v = *reinterpret_cast<const float2*>(lds + xi * 1024 + kq * 64 + t);
                m[0][xi] = v.x; m[1][xi] = v.y;
            }
            float o[2][4];                                       // [row][x]
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float s0[4], s1[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    s0[b] = m[j][b] + m[j][4 + b] + m[j][8 + b];
                    s1[b] = m[j][4 + b] - m[j][8 + b] - m[j][12 + b];
                }
                o[0][2 * j] = s0[0] + s0[1] + s0[2]; o[0][2 * j + 1] = s0[1] - s0[2] - s0[3];
                o[1][2 * j] = s1[0] + s1[1] + s1[2]; o[1][2 * j + 1] = s1[1] - s1[2] - s1[3];
            }
            const int oy = by * 16 + 2 * (t >> 3), ox = bx * 16 + 2 * (t & 7);
            if (ep.stats) {
                // per-(image, channel) sum and sum of squares of the raw convolution output, for a
                // following instance norm: this lane's in-image values, reduced over the 32 lanes
                // that share the channel, one pair of float64 atomics per channel and workgroup
                float ps = 0.0f, pq = 0.0f;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (oy + r < H && ox + c < W) { ps += o[r][c]; pq += o[r][c] * o[r][c]; }
                // float32 only over the lane's own few pixels; across lanes in float64, so that the variance
                // E[x^2] - mean^2 formed from these sums survives planes whose mean dwarfs their spread
                double dps = (double)ps, dpq = (double)pq;
#pragma unroll
                for (int m2 = 16; m2 > 0; m2 >>= 1) { dps += __shfl_xor(dps, m2, VO_WAVE); dpq += __shfl_xor(dpq, m2, VO_WAVE); }
                if (tp == 0 && k < K) {
                    double* st = ep.stats + ((int64_t)n * K + k) * 2;
                    atomicAdd(st, dps);
                    atomicAdd(st + 1, dpq);
                }
            }
            if (k < K && oy < H && ox < W) {
                const int64_t off = ((int64_t)k * H + oy) * W + ox;
                float* p = yn + off;
                if (ep.shift) {
                    const float sc = ep.scale ? ep.scale[k] : 1.0f, sh = ep.shift[k];
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[r][c] = o[r][c] * sc + sh;
                }
                if (ep.relu) {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[r][c] = fmaxf(o[r][c], 0.0f);
                }
                if (ep.skip) {
                    const float* sp = ep.skip + (int64_t)n * K * H * W + off;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        if (oy + r >= H) break;
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (ox + c < W) o[r][c] = fmaxf(o[r][c] + sp[(int64_t)r * W + c], 0.0f);
                    }
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    if (oy + r >= H) break;
                    if (quad_store) {                            // W % 4 == 0, y 16-byte aligned: ox + 3 < W always
                        *reinterpret_cast<float4*>(p + (int64_t)r * W) = make_float4(o[r][0], o[r][1], o[r][2], o[r][3]);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (ox + c < W) p[(int64_t)r * W + c] = o[r][c];
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int vo_winograd_filter(const float* weight, float* transformed, int K, int C, void* stream) {
    VO_REQUIRE(weight && transformed, "vo_winograd_filter: null pointer");
    VO_REQUIRE(K > 0 && C > 0, "vo_winograd_filter: bad sizes K=%d C=%d", K, C);
    if (C % CB != 0) return vo_fail(VO_ERR_UNSUPPORTED, "vo_winograd_filter: C=%d is not a multiple of %d", C, CB);
    const int Kpad = (K + KB - 1) / KB * KB;
    const int64_t n = (int64_t)Kpad * C;
    filter_kernel<<<dim3((unsigned)vo_ceil_div(n, 256)), dim3(256), 0, vo_stream(stream)>>>(weight, transformed, K, C, Kpad);
    VO_CHECK_LAUNCH("vo_winograd_filter");
    return VO_OK;
}

extern "C" int vo_conv3x3_winograd(const float* x, const float* transformed, float* y, int64_t y_batch_stride,
                                   const float* scale, const float* shift, int relu, const float* skip, double* stats,
                                   const double* in_stats, float in_eps, int N, int C, int H, int W, int K,
                                   void* stream) {
    VO_REQUIRE(x && transformed && y, "vo_conv3x3_winograd: null pointer");
    VO_REQUIRE(!scale || shift, "vo_conv3x3_winograd: scale without shift");
    VO_REQUIRE(!stats || (!shift && !relu && !skip), "vo_conv3x3_winograd: statistics are of the raw output, no epilogue with them");
    VO_REQUIRE(skip != y, "vo_conv3x3_winograd: skip must not alias the output");
    VO_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && K > 0, "vo_conv3x3_winograd: bad sizes");
    if (y_batch_stride == 0) y_batch_stride = (int64_t)K * H * W;
    VO_REQUIRE(y_batch_stride >= (int64_t)K * H * W, "vo_conv3x3_winograd: y_batch_stride smaller than one item");
    if (C % CB != 0) return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv3x3_winograd: C=%d is not a multiple of %d", C, CB);
    if (in_stats && C > MAX_NORM_C)
        return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv3x3_winograd: normalise-on-load supports at most %d input channels", MAX_NORM_C);
    const int bx = (W + 15) / 16, by = (H + 15) / 16;
    const int64_t blocks = (int64_t)bx * by * N;
    const int kblocks = (K + KB - 1) / KB;
    if (blocks * kblocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv3x3_winograd: grid too large");
    const dim3 grid((unsigned)(blocks * kblocks));
    const Epilogue ep{scale, shift, skip, relu, y_batch_stride, stats};
    if (in_stats)
        wino_kernel<true><<<grid, dim3(512), 0, vo_stream(stream)>>>(x, transformed, y, ep, InputNorm{in_stats, in_eps}, C, H, W,
                                                                     K, bx, by, kblocks);
    else
        wino_kernel<false><<<grid, dim3(512), 0, vo_stream(stream)>>>(x, transformed, y, ep, InputNorm{nullptr, 0.0f}, C, H, W,
                                                                      K, bx, by, kblocks);
    VO_CHECK_LAUNCH("vo_conv3x3_winograd");
    return VO_OK;
}
