// 1x5 and 5x1 stride-1 "same" convolutions in float32 as one-dimensional Winograd F(4, 5): eight
// multiplies per four outputs and channel pair instead of twenty, the eight transformed-domain
// products on the matrix cores.  These are the gate convolutions of RAFT's separable GRU
// (SepConvGRU: 1x5 then 5x1 over [h | x], 384 -> 256 and 384 -> 128 channels every iteration).
//
//   M_xi[k, t] = sum_c U_xi[k, c] * V_xi[c, t],   U = G g,  V = B^T d,  y = A^T M
// with the interpolation points 0, +-1, +-2, +-1/2, inf:
//   B^T rows: [-1 0 21/4 0 -21/4 0 1 0], [0 1 1 -17/4 -17/4 1 1 0], [0 -1 1 17/4 -17/4 -1 1 0],
//             [0 1/2 1/4 -5/2 -5/4 2 1 0], [0 -1/2 1/4 5/2 -5/4 -2 1 0], [0 2 4 -5/2 -5 1/2 1 0],
//             [0 -2 4 5/2 -5 -1/2 1 0], [0 -1 0 21/4 0 -21/4 0 1]
//   A^T rows: [1 1 1 1 1 1 1 0], [0 1 -1 2 -2 1/2 -1/2 0], [0 1 1 4 4 1/4 1/4 0], [0 1 -1 8 -8 1/8 -1/8 1]
// (one-dimensional transforms: the float32 rounding error stays within a few times that of the
// direct sum).  One workgroup of 8 waves owns a 16x16-pixel block (64 tiles of four pixels along
// the filter's axis; 32x16 as a template option) and 64 output channels; wave w owns position w:
// 2 k-tiles x 2 tile groups = 64 accumulator registers, one pair of 16-byte filter loads per chunk
// of 8 input channels.
#include "vo_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CB = 8;                      // input channels per chunk
constexpr int KB = 64;                     // output channels per workgroup
constexpr int NT = 512;                    // threads per workgroup
constexpr int A_CHUNK = 8 * 2 * 64 * 4;    // floats of transformed filter per (channel block, chunk)

// U = G g for every (k, c), in the register order of the MFMA A operand:
//   Ug[kb][chunk][xi][kt][lane][ks]:  k = kb*64 + kt*32 + (lane & 31),  c = chunk*8 + 2*ks + (lane >> 5)
__global__ __launch_bounds__(256) void filter1d_kernel(const float* __restrict__ g, float* __restrict__ Ug, int K, int C,
                                                       int Kpad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Kpad * C) return;
    const int k = (int)(idx / C), c = (int)(idx % C);
    double w[5];
    for (int s = 0; s < 5; ++s) w[s] = k < K ? (double)g[((int64_t)k * C + c) * 5 + s] : 0.0;
    const double G[8][5] = {{-1, 0, 0, 0, 0},
                            {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9},
                            {-2.0 / 9, 2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},
                            {1.0 / 90, 1.0 / 45, 2.0 / 45, 4.0 / 45, 8.0 / 45},
                            {1.0 / 90, -1.0 / 45, 2.0 / 45, -4.0 / 45, 8.0 / 45},
                            {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45},
                            {32.0 / 45, -16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45},
                            {0, 0, 0, 0, 1}};
    const int kb = k >> 6, kt = (k >> 5) & 1, kl = k & 31, chunk = c >> 3, ks = (c & 7) >> 1, kk = c & 1;
    const int nchunks = C / CB;
    for (int xi = 0; xi < 8; ++xi) {
        double u = 0.0;
        for (int s = 0; s < 5; ++s) u += G[xi][s] * w[s];
        const int64_t o = ((((((int64_t)kb * nchunks + chunk) * 8 + xi) * 2 + kt) * 64) + kk * 32 + kl) * 4 + ks;
        Ug[o] = (float)u;
    }
}

// B^T d, eight values in and out (26 operations).
__device__ __forceinline__ void in_1d(const float* d, float* t) {
    t[0] = fmaf(5.25f, d[2] - d[4], d[6] - d[0]);
    t[7] = fmaf(5.25f, d[3] - d[5], d[7] - d[1]);
    const float e1 = fmaf(-4.25f, d[4], d[2] + d[6]), o1 = fmaf(-4.25f, d[3], d[1] + d[5]);
    t[1] = e1 + o1;
    t[2] = e1 - o1;
    const float e2 = fmaf(0.25f, d[2], fmaf(-1.25f, d[4], d[6])), o2 = fmaf(0.5f, d[1], fmaf(-2.5f, d[3], 2.0f * d[5]));
    t[3] = e2 + o2;
    t[4] = e2 - o2;
    const float e3 = fmaf(4.0f, d[2], fmaf(-5.0f, d[4], d[6])), o3 = fmaf(2.0f, d[1], fmaf(-2.5f, d[3], 0.5f * d[5]));
    t[5] = e3 + o3;
    t[6] = e3 - o3;
}
// A^T m, eight values in, four out.
__device__ __forceinline__ void out_1d(const float* m, float* o) {
    const float p1 = m[1] + m[2], q1 = m[1] - m[2], p2 = m[3] + m[4], q2 = m[3] - m[4], p3 = m[5] + m[6], q3 = m[5] - m[6];
    o[0] = m[0] + p1 + p2 + p3;
    o[1] = fmaf(2.0f, q2, fmaf(0.5f, q3, q1));
    o[2] = fmaf(4.0f, p2, fmaf(0.25f, p3, p1));
    o[3] = fmaf(8.0f, q2, fmaf(0.125f, q3, q1)) + m[7];
}

// DIR 0: filter along W (1x5, padding (0,2)); DIR 1: filter along H (5x1, padding (2,0)).
// NTG: groups of 32 tiles per workgroup.  4: a 32x16-pixel block, 128 accumulator registers, one
// workgroup per CU.  2: a 16x16-pixel block, 64 accumulator registers, at most 128 registers and
// 64 KB of LDS, so two workgroups share a CU and the matrix pipe is fed by one while the other
// sits in its per-chunk barrier / LDS round trip.
template <int DIR, int NTG>
__global__ __launch_bounds__(NT, NTG == 2 ? 4 : 2) void wino1d_kernel(const float* __restrict__ x, const float* __restrict__ Ug,
                                                                      float* __restrict__ y, int C, int H, int W, int K,
                                                                      int blocks_x, int blocks_y, int kblocks) {
    constexpr int TILES = 32 * NTG, ROWS = 8 * NTG;              // tiles and pixel rows per workgroup (16 columns)
    constexpr int V_XI = CB * TILES, V_FLOATS = 8 * V_XI;        // per position: [kk][tile][ks], c = 2*ks + kk
    constexpr int PR = DIR == 0 ? ROWS : ROWS + 4, PC = DIR == 0 ? 20 : 16;    // raw patch rows, columns
    constexpr int RAW_C = PR * PC, RAW_FLOATS = CB * RAW_C;
    constexpr int RAW_MAX = CB * ROWS * 20;                      // the larger of the two patch shapes
    constexpr int RAW_PER_LANE = RAW_MAX / NT;
    constexpr int M_FLOATS = 8 * TILES * 16;                     // one round of the output transform: [xi][tile][channel]
    constexpr int LDS_FLOATS = 2 * M_FLOATS > 2 * V_FLOATS + 2 * RAW_MAX ? 2 * M_FLOATS : 2 * V_FLOATS + 2 * RAW_MAX;
    __shared__ __align__(16) float lds[LDS_FLOATS];
    float* Vs = lds;
    float* raw = lds + 2 * V_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid;
    {   // one contiguous range of the block list per XCD (workgroups are dealt round robin over 8)
        const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, q = nwg >> 3, r = nwg & 7u;
        bid = (int)(((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3));
    }
    const int kb = bid % kblocks; bid /= kblocks;
    const int bx = bid % blocks_x; bid /= blocks_x;
    const int by = bid % blocks_y;
    const int n = bid / blocks_y;
    const int nchunks = C / CB;
    const int y0 = by * ROWS - (DIR == 1 ? 2 : 0), x0 = bx * 16 - (DIR == 0 ? 2 : 0);
    const float* xn = x + (int64_t)n * C * H * W;

    // Lanes of the zero padding carry an offset past the extent: a buffer load returns 0.0f there.
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, C * H * W * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Ug + (int64_t)kb * nchunks * A_CHUNK), 0, nchunks * A_CHUNK * 4, 0x00020000);
    const int chunk_bytes = CB * H * W * 4;

    int raw_off[RAW_PER_LANE];
#pragma unroll
    for (int i = 0; i < RAW_PER_LANE; ++i) {
        const int idx = tid + i * NT;
        const int c = idx / RAW_C, rem = idx - c * RAW_C, r = rem / PC, col = rem - r * PC;
        const int gy = y0 + r, gx = x0 + col;
        const bool ok = idx < RAW_FLOATS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        raw_off[i] = ok ? ((c * H + gy) * W + gx) * 4 : (int)0x80000000;
    }
    float rv[RAW_PER_LANE];
    auto load_raw = [&](int chunk) {
        chunk = chunk < nchunks ? chunk : nchunks - 1;           // past the end: reload the last chunk, never used
#pragma unroll
        for (int i = 0; i < RAW_PER_LANE; ++i)
            rv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, raw_off[i], chunk * chunk_bytes, 0));
    };
    auto store_raw = [&](int buf) {                              // the patch is stored linearly: element idx at idx
        float* rb = raw + buf * RAW_MAX + tid;
#pragma unroll
        for (int i = 0; i < RAW_PER_LANE; ++i)
            if (i * NT + NT <= RAW_FLOATS || tid + i * NT < RAW_FLOATS) rb[i * NT] = rv[i];
    };

    // filter fragments of the current and the next chunk: [k-tile] -> four k-steps
    const int a_off = (wave * 2 * 64 + lane) * 16;               // + kt * 1024 + chunk * A_CHUNK * 4 bytes
    u32x4 a_cur[2], a_nxt[2];
    auto load_a = [&](int chunk) {
        chunk = chunk < nchunks ? chunk : nchunks - 1;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
            a_nxt[kt] = __builtin_amdgcn_raw_buffer_load_b128(urs, a_off + kt * 1024, chunk * (A_CHUNK * 4), 0);
    };

    f32x16 acc[2][NTG];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < NTG; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][j][e] = 0.0f;

    // transform role, NTG/2 (channel, tile) per lane and chunk: pass p of wave w takes combination
    // m = w*NTG/2 + p: kk = m & 1 and the 16 tiles of group m >> 1; lane: ks = lane & 3, tile = 16*group + (lane >> 2)
    auto transform = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NTG / 2; ++p) {
            const int combo = (NTG / 2) * wave + p, t_kk = combo & 1, tile = (combo >> 1) * 16 + (lane >> 2);
            const int c = 2 * (lane & 3) + t_kk;
            float d[8], t[8];
            if (DIR == 0) {
                const float* src = raw + buf * RAW_MAX + c * RAW_C + (tile >> 2) * PC + 4 * (tile & 3);
                const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
                d[0] = lo.x; d[1] = lo.y; d[2] = lo.z; d[3] = lo.w; d[4] = hi.x; d[5] = hi.y; d[6] = hi.z; d[7] = hi.w;
            } else {
                const float* src = raw + buf * RAW_MAX + c * RAW_C + (4 * (tile >> 4)) * PC + (tile & 15);
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = src[j * PC];
            }
            in_1d(d, t);
            float* vo = Vs + buf * V_FLOATS + t_kk * (TILES * 4) + (combo >> 1) * 64 + lane;
#pragma unroll
            for (int xi = 0; xi < 8; ++xi) vo[xi * V_XI] = t[xi];
        }
    };
    auto multiply = [&](int buf) {
        const float* vb = Vs + buf * V_FLOATS + wave * V_XI + (lane >> 5) * (TILES * 4) + (lane & 31) * 4;
        float4 b[NTG];
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) b[nt] = *reinterpret_cast<const float4*>(vb + nt * 128);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int nt = 0; nt < NTG; ++nt) {
                const float bv = ks == 0 ? b[nt].x : ks == 1 ? b[nt].y : ks == 2 ? b[nt].z : b[nt].w;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    acc[kt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a_cur[kt][ks]), bv, acc[kt][nt], 0, 0, 0);
            }
    };

    load_raw(0);
    load_a(0);
    store_raw(0);
    load_raw(1);
    __syncthreads();
    transform(0);
    store_raw(1);
    load_raw(2);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
        // V[cur] is complete, raw[cur ^ 1] holds chunk + 1, the registers hold chunk + 2
        a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1];
        load_a(chunk + 1);
        if (chunk + 1 < nchunks) transform(cur ^ 1);
        multiply(cur);
        store_raw(cur);                                          // raw[cur] was consumed one iteration ago
        load_raw(chunk + 3);
        __syncthreads();
    }

    // Output transform, 16 output channels per round, M[xi][tile][channel] through LDS: a lane's four
    // accumulator elements e..e+3 are four consecutive channels (one 16-byte store, its slot inside
    // the tile's 16 channels XOR-swizzled with the tile index against bank conflicts).  M is
    // double-buffered: the stores of round q+1 go out while round q is transformed.
    float* yn = y + (int64_t)n * K * H * W;
    const int kk = lane >> 5, nn = lane & 31;
    auto put_m = [&](int q) {
        float* mb = lds + (q & 1) * M_FLOATS + wave * (TILES * 16);
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt)
#pragma unroll
            for (int eh = 0; eh < 2; ++eh) {
                const int e = (q & 1) * 8 + eh * 4, g = 2 * eh + kk, tile = nt * 32 + nn;
                const f32x16& a = acc[q >> 1][nt];
                *reinterpret_cast<float4*>(mb + tile * 16 + 4 * (g ^ ((tile >> 2) & 3))) = make_float4(a[e], a[e + 1], a[e + 2], a[e + 3]);
            }
    };
    put_m(0);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < 3) put_m(q + 1);
        const int tile = tid & (TILES - 1);
        int oy, ox;
        if (DIR == 0) { oy = by * ROWS + (tile >> 2); ox = bx * 16 + 4 * (tile & 3); }
        else { oy = by * ROWS + 4 * (tile >> 4); ox = bx * 16 + (tile & 15); }
#pragma unroll
        for (int j = 0; j < 16 * TILES / NT; ++j) {
            const int kq = tid / TILES + (NT / TILES) * j;
            const int k = kb * KB + q * 16 + kq;
            const float* mp = lds + (q & 1) * M_FLOATS + tile * 16 + 4 * ((kq >> 2) ^ ((tile >> 2) & 3)) + (kq & 3);
            float m[8], o[4];
#pragma unroll
            for (int xi = 0; xi < 8; ++xi) m[xi] = mp[xi * (TILES * 16)];
            out_1d(m, o);
            if (k < K && oy < H && ox < W) {
                float* p = yn + ((int64_t)k * H + oy) * W + ox;
                if (DIR == 0) {
                    if (ox + 3 < W && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
                        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (ox + c < W) p[c] = o[c];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (oy + r < H) p[(int64_t)r * W] = o[r];
                }
            }
        }
        if (q < 3) __syncthreads();
    }
}

}  // namespace

extern "C" int vo_winograd1d_filter(const float* weight, float* transformed, int K, int C, void* stream) {
    VO_REQUIRE(weight && transformed, "vo_winograd1d_filter: null pointer");
    VO_REQUIRE(K > 0 && C > 0, "vo_winograd1d_filter: bad sizes K=%d C=%d", K, C);
    if (C % CB != 0) return vo_fail(VO_ERR_UNSUPPORTED, "vo_winograd1d_filter: C=%d is not a multiple of %d", C, CB);
    const int Kpad = (K + KB - 1) / KB * KB;
    const int64_t n = (int64_t)Kpad * C;
    filter1d_kernel<<<dim3((unsigned)vo_ceil_div(n, 256)), dim3(256), 0, vo_stream(stream)>>>(weight, transformed, K, C, Kpad);
    VO_CHECK_LAUNCH("vo_winograd1d_filter");
    return VO_OK;
}

extern "C" int vo_conv5_winograd(const float* x, const float* transformed, float* y, int N, int C, int H, int W, int K,
                                 int vertical, void* stream) {
    VO_REQUIRE(x && transformed && y, "vo_conv5_winograd: null pointer");
    VO_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && K > 0, "vo_conv5_winograd: bad sizes");
    VO_REQUIRE(x != y, "vo_conv5_winograd: the output must not alias the input");
    if (C % CB != 0) return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv5_winograd: C=%d is not a multiple of %d", C, CB);
    if ((int64_t)C * H * W * 4 >= 0x7fffffffLL)
        return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv5_winograd: one image of the input exceeds 2 GiB");
    // 16x16-pixel blocks, two workgroups per CU: equal or faster than 32x16 blocks on every shape measured
    // (384 -> 128 @60x80x32: 0.34 vs 0.42 ms, the smaller blocks also even out the last round of workgroups)
    constexpr int NTG = 2;
    const int rows = 8 * NTG;
    const int bx = (W + 15) / 16, by = (H + rows - 1) / rows;
    const int64_t blocks = (int64_t)bx * by * N;
    const int kblocks = (K + KB - 1) / KB;
    if (blocks * kblocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv5_winograd: grid too large");
    const dim3 grid((unsigned)(blocks * kblocks));
    hipStream_t st = vo_stream(stream);
    if (vertical) wino1d_kernel<1, NTG><<<grid, dim3(NT), 0, st>>>(x, transformed, y, C, H, W, K, bx, by, kblocks);
    else wino1d_kernel<0, NTG><<<grid, dim3(NT), 0, st>>>(x, transformed, y, C, H, W, K, bx, by, kblocks);
    VO_CHECK_LAUNCH("vo_conv5_winograd");
    return VO_OK;
}
