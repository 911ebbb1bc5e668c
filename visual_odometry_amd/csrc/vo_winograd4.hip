// 3x3 stride-1 convolution (padding 1) in float32 as Winograd F(4x4, 3x3): 36 multiplies per
// 4x4 output tile and channel pair instead of 144 (the F(2x2, 3x3) form of vo_winograd.hip
// needs 64 for the same 16 pixels), the 36 transformed-domain products on the matrix cores.
//
// Same decomposition as vo_winograd.hip:  M_xi[k, t] = sum_c U_xi[k, c] * V_xi[c, t]  for the 36
// tile positions xi = 6a + b, with  U = G g G^T,  V = B^T d B,  Y = A^T M A  and the usual
// interpolation points 0, +-1, +-2, inf:
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// One workgroup of 12 waves owns 4x8 tiles (16x32 output pixels) and 64 output channels; wave w
// owns positions 3w..3w+2 (96 accumulator registers).  Input channels stream in chunks of 8: the
// raw 18x34 patch of a chunk goes to LDS, every lane transforms one (channel, tile, third of the
// rows of V), the waves multiply, and the accumulators pass through LDS once more for the output
// transform.  On this hardware ordinary vector instructions take issue time away from float32
// matrix instructions of the same SIMD (about 3 cycles each), so the layout is chosen to keep the
// per-chunk instruction count low: 16-byte fragment loads, one barrier per chunk.
#include "vo_common.h"

#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CB = 8;                      // input channels per chunk
constexpr int KB = 64;                     // output channels per workgroup
constexpr int NT = 768;                    // threads per workgroup
constexpr int RAW_C = 18 * 34;             // raw patch of 4x8 (or 8x4) tiles of 4x4 pixels plus the halo: 612 floats per channel, no padding
constexpr int RAW_FLOATS = CB * RAW_C;     // 4896
constexpr int RAW_PER_LANE = (RAW_FLOATS + NT - 1) / NT;     // 7
constexpr int V_XI = CB * 32;              // floats per position: [kk][tile][ks], c = 2*ks + kk
constexpr int V_FLOATS = 36 * V_XI;        // 9216
constexpr int A_CHUNK = 36 * 2 * 64 * 4;   // floats of transformed filter per (channel block, chunk)
constexpr int M_FLOATS = 36 * 32 * 16;     // one round of the output transform: [xi][tile][channel]
constexpr int LDS_FLOATS = 2 * M_FLOATS > 2 * V_FLOATS + 2 * RAW_FLOATS ? 2 * M_FLOATS : 2 * V_FLOATS + 2 * RAW_FLOATS;
constexpr int MAX_NORM_C = 512;
constexpr int RING = 3;                    // filter-fragment loads in flight per wave

// U = G g G^T for every (k, c), in the register order of the MFMA A operand:
//   Ug[kb][chunk][xi][kt][lane][ks]:  k = kb*64 + kt*32 + (lane & 31),  c = chunk*8 + 2*ks + (lane >> 5)
__global__ __launch_bounds__(256) void filter4_kernel(const float* __restrict__ g, float* __restrict__ Ug, int K, int C,
                                                      int Kpad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Kpad * C) return;
    const int k = (int)(idx / C), c = (int)(idx % C);
    double w[3][3];
    for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s) w[r][s] = k < K ? (double)g[(((int64_t)k * C + c) * 3 + r) * 3 + s] : 0.0;
    const double G[6][3] = {{0.25, 0.0, 0.0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    double t[6][3];
    for (int a = 0; a < 6; ++a)
        for (int s = 0; s < 3; ++s) t[a][s] = G[a][0] * w[0][s] + G[a][1] * w[1][s] + G[a][2] * w[2][s];
    const int kb = k >> 6, kt = (k >> 5) & 1, kl = k & 31, chunk = c >> 3, ks = (c & 7) >> 1, kk = c & 1;
    const int nchunks = C / CB;
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
            const double u = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
            const int xi = 6 * a + b;
            const int64_t o = ((((((int64_t)kb * nchunks + chunk) * 36 + xi) * 2 + kt) * 64) + kk * 32 + kl) * 4 + ks;
            Ug[o] = (float)u;
        }
}

struct Epilogue { const float* scale; const float* shift; const float* skip; int relu; int64_t y_batch_stride; double* stats; };
struct InputNorm { const double* stats; float eps; };

// Rows `g` of the one-dimensional input transform B^T d (six values in, two out):
// g = 0 -> rows 0 and 5, g = 1 -> rows 1 and 2, g = 2 -> rows 3 and 4.  Four operations each.
template <int G> __device__ __forceinline__ void in_rows(float d0, float d1, float d2, float d3, float d4, float d5,
                                                         float& ta, float& tb) {
    if (G == 0) {
        ta = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
        tb = fmaf(4.0f, d1, fmaf(-5.0f, d3, d5));
    } else if (G == 1) {
        const float p = fmaf(-4.0f, d2, d4), q = fmaf(-4.0f, d1, d3);
        ta = p + q;
        tb = p - q;
    } else {
        const float p = d4 - d2, q = d3 - d1;
        ta = fmaf(2.0f, q, p);
        tb = fmaf(-2.0f, q, p);
    }
}
// All six rows of the same transform (applied along the other axis).
__device__ __forceinline__ void in_all(const float* t, float* o) {
    o[0] = fmaf(4.0f, t[0], fmaf(-5.0f, t[2], t[4]));
    o[5] = fmaf(4.0f, t[1], fmaf(-5.0f, t[3], t[5]));
    const float p = fmaf(-4.0f, t[2], t[4]), q = fmaf(-4.0f, t[1], t[3]);
    o[1] = p + q;
    o[2] = p - q;
    const float r = t[4] - t[2], s = t[3] - t[1];
    o[3] = fmaf(2.0f, s, r);
    o[4] = fmaf(-2.0f, s, r);
}
// One-dimensional output transform A^T m (six values in, four out).
__device__ __forceinline__ void out_1d(float m0, float m1, float m2, float m3, float m4, float m5, float* o) {
    const float p = m1 + m2, q = m1 - m2, r = m3 + m4, s = m3 - m4;
    o[0] = m0 + p + r;
    o[1] = fmaf(2.0f, s, q);
    o[2] = fmaf(4.0f, r, p);
    o[3] = fmaf(8.0f, s, q) + m5;
}

// TALL: the workgroup's 32 tiles are 8 rows x 4 columns (32x16 pixels) instead of 4 x 8 (16x32 pixels); the
// host picks whichever wastes less of the map (80 pixels of width are 5 blocks of 16 but 3 of 32).
// HALF: the launch covers one block of output channels of which only the lower 32 exist (K mod 64
// in 1..32): the upper k-tile's matrix instructions, filter loads and output rounds are left out.
template <bool NORM_IN, bool HALF, bool TALL>
__global__ __launch_bounds__(NT) void wino4_kernel(const float* __restrict__ x, const float* __restrict__ Ug,
                                                   float* __restrict__ y, Epilogue ep, InputNorm in, int C, int H, int W,
                                                   int K, int blocks_x, int blocks_y, int kblocks, int kb_base) {
    // V and the raw patch are double-buffered; after the loop the array holds two buffers of M
    // (36 positions x 32 tiles x 16 channels) for the output transform.
    __shared__ __align__(16) float lds[LDS_FLOATS + (NORM_IN ? 2 * MAX_NORM_C : 0)];
    float* Vs = lds;
    float* raw = lds + 2 * V_FLOATS;
    float2* nrm = reinterpret_cast<float2*>(lds + LDS_FLOATS);

    constexpr int TX = TALL ? 4 : 8;                             // tile columns of the block; tile t is at (t / TX, t % TX)
    constexpr int BH = TALL ? 32 : 16, BW = TALL ? 16 : 32;      // block size in pixels
    constexpr int PC = BW + 2;                                   // columns of the raw patch
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: scalar
    int bid;
    {   // one contiguous range of the block list per XCD (workgroups are dealt round robin over 8)
        const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, q = nwg >> 3, r = nwg & 7u;
        bid = (int)(((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3));
    }
    const int kb = kb_base + bid % kblocks; bid /= kblocks;
    const int bx = bid % blocks_x; bid /= blocks_x;
    const int by = bid % blocks_y;
    const int n = bid / blocks_y;
    const int nchunks = C / CB;
    const int y0 = by * BH - 1, x0 = bx * BW - 1;
    const float* xn = x + (int64_t)n * C * H * W;

    // Buffer descriptors.  Lanes of the zero padding carry an offset past the extent, which the
    // range check of a buffer load turns into 0.0f, so every load is unconditional.
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, C * H * W * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Ug + (int64_t)kb * nchunks * A_CHUNK), 0, nchunks * A_CHUNK * 4, 0x00020000);
    const int chunk_bytes = CB * H * W * 4;

    int raw_off[RAW_PER_LANE];
    unsigned raw_ch = 0;                                         // channel within the chunk, 3 bits per element
#pragma unroll
    for (int i = 0; i < RAW_PER_LANE; ++i) {
        const int idx = tid + i * NT;
        const int c = idx / RAW_C, rem = idx - c * RAW_C, r = rem / PC, col = rem - r * PC;
        const int gy = y0 + r, gx = x0 + col;
        const bool ok = idx < RAW_FLOATS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        raw_off[i] = ok ? ((c * H + gy) * W + gx) * 4 : (int)0x80000000;
        raw_ch |= (unsigned)(c & 7) << (3 * i);
    }
    float rv[RAW_PER_LANE];
    auto load_raw = [&](int chunk) {
        chunk = chunk < nchunks ? chunk : nchunks - 1;           // past the end: reload the last chunk, never used
#pragma unroll
        for (int i = 0; i < RAW_PER_LANE; ++i)
            rv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, raw_off[i], chunk * chunk_bytes, 0));
    };
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
        __syncthreads();
    }
    auto store_raw = [&](int buf, int chunk) {                   // the patch is stored linearly: element idx at idx
        float* rb = raw + buf * RAW_FLOATS + tid;
        chunk = chunk < nchunks ? chunk : nchunks - 1;
#pragma unroll
        for (int i = 0; i < RAW_PER_LANE; ++i) {
            float v = rv[i];
            if (NORM_IN) {                                       // padding stays zero: it pads the NORMALISED tensor
                const float2 ss = nrm[chunk * CB + ((raw_ch >> (3 * i)) & 7)];
                v = raw_off[i] >= 0 ? fmaxf(fmaf(v, ss.x, ss.y), 0.0f) : 0.0f;
            }
            if (i * NT + NT <= RAW_FLOATS || tid + i * NT < RAW_FLOATS) rb[i * NT] = v;
        }
    };

    // filter fragments: the six 16-byte loads of a chunk ((position, k-tile) step s in slot s); a slot
    // is refilled for the next chunk as soon as its matrix instructions have been issued
    const int a_off = (3 * wave * 2 * 64 + lane) * 16;           // + step * 1024 + chunk * A_CHUNK * 4 bytes
    u32x4 ar[RING];
    auto load_a = [&](int chunk, int step, int slot) {
        chunk = chunk < nchunks ? chunk : nchunks - 1;
        ar[slot] = __builtin_amdgcn_raw_buffer_load_b128(urs, a_off + step * 1024, chunk * (A_CHUNK * 4), 0);
    };

    f32x16 acc[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][j][e] = 0.0f;

    // transform role: waves 4g..4g+3 produce rows `g` of V; inside a group, wave v: kk = v & 1,
    // tiles 16*(v >> 1) .. +15; lane: ks = lane & 3, tile = 16*(v >> 1) + (lane >> 2); c = 2*ks + kk
    const int tg = wave >> 2, tv = wave & 3;
    const int t_kk = tv & 1, t_tile = (tv >> 1) * 16 + (lane >> 2), t_c = 2 * (lane & 3) + t_kk;
    const int t_raw = t_c * RAW_C + (4 * (t_tile / TX)) * PC + 4 * (t_tile % TX);
    const int t_v = t_kk * 128 + (tv >> 1) * 64 + lane;          // + xi * V_XI
    auto transform_g = [&](int buf, auto gtag) {
        constexpr int G = decltype(gtag)::value;
        const float* p = raw + buf * RAW_FLOATS + t_raw;
        float* vo = Vs + buf * V_FLOATS + t_v;
        float t[2][6];                                           // the two rows of B^T d this lane owns
        constexpr int r_lo = G == 0 ? 0 : 1, r_hi = G == 0 ? 5 : 4;
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) {                         // pairs of columns
            float2 d[6];
#pragma unroll
            for (int r = r_lo; r <= r_hi; ++r) d[r] = *reinterpret_cast<const float2*>(p + r * PC + 2 * cp);
            if (G != 0) { d[0] = make_float2(0.0f, 0.0f); d[5] = d[0]; }
            in_rows<G>(d[0].x, d[1].x, d[2].x, d[3].x, d[4].x, d[5].x, t[0][2 * cp], t[1][2 * cp]);
            in_rows<G>(d[0].y, d[1].y, d[2].y, d[3].y, d[4].y, d[5].y, t[0][2 * cp + 1], t[1][2 * cp + 1]);
        }
        constexpr int row_a = G == 0 ? 0 : (G == 1 ? 1 : 3), row_b = G == 0 ? 5 : (G == 1 ? 2 : 4);
        float o[6];
        in_all(t[0], o);
#pragma unroll
        for (int b = 0; b < 6; ++b) vo[(6 * row_a + b) * V_XI] = o[b];
        in_all(t[1], o);
#pragma unroll
        for (int b = 0; b < 6; ++b) vo[(6 * row_b + b) * V_XI] = o[b];
    };
    auto transform = [&](int buf) {
        if (tg == 0) transform_g(buf, std::integral_constant<int, 0>{});
        else if (tg == 1) transform_g(buf, std::integral_constant<int, 1>{});
        else transform_g(buf, std::integral_constant<int, 2>{});
    };
    // multiply: positions 3*wave + xl; the B fragment of a position is one 16-byte read (four k-steps)
    auto multiply_xl = [&](int buf, int chunk, auto xtag) {
        constexpr int xl = decltype(xtag)::value;
        const float4 b = *reinterpret_cast<const float4*>(Vs + buf * V_FLOATS + (3 * wave + xl) * V_XI + lane * 4);
        const float bs[4] = {b.x, b.y, b.z, b.w};
        constexpr int s0 = HALF ? xl : (xl * 2) % RING, s1 = (xl * 2 + 1) % RING;
        // consecutive matrix instructions alternate between the two k-tiles' accumulators
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc[xl][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ar[s0][ks]), bs[ks], acc[xl][0], 0, 0, 0);
            if (!HALF) acc[xl][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ar[s1][ks]), bs[ks], acc[xl][1], 0, 0, 0);
        }
        if (HALF) {                                              // three fragments per chunk: a full chunk of lookahead
            load_a(chunk + 1, 2 * xl, xl);
        } else {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int step = xl * 2 + kt;
                if (step + RING < 6) load_a(chunk, step + RING, step % RING);
                else load_a(chunk + 1, step + RING - 6, step % RING);
            }
        }
    };
    using X0 = std::integral_constant<int, 0>;
    using X1 = std::integral_constant<int, 1>;
    using X2 = std::integral_constant<int, 2>;

    load_raw(0);
#pragma unroll
    for (int s = 0; s < RING; ++s) load_a(0, HALF ? 2 * s : s, s);
    store_raw(0, 0);
    load_raw(1);
    __syncthreads();
    transform(0);
    store_raw(1, 1);
    load_raw(2);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
        // V[cur] is complete, raw[cur ^ 1] holds chunk + 1, the registers hold chunk + 2
        if (chunk + 1 < nchunks) transform(cur ^ 1);
        multiply_xl(cur, chunk, X0{}); multiply_xl(cur, chunk, X1{}); multiply_xl(cur, chunk, X2{});
        store_raw(cur, chunk + 2);                               // raw[cur] was consumed one iteration ago
        load_raw(chunk + 3);
        __syncthreads();
    }

    // Output transform, 16 output channels per round, M[xi][tile][channel] through LDS.  A lane's
    // four accumulator elements e..e+3 are four consecutive channels: one 16-byte store, its slot
    // inside the tile's 16 channels XOR-swizzled with the tile index so that neither these stores
    // nor the scalar reads below collide on banks.  M is double-buffered: the stores of round
    // q+1 go out while round q is transformed, one barrier per round.
    float* yn = y + (int64_t)n * ep.y_batch_stride;
    const bool quad_store = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (ep.y_batch_stride & 3) == 0 &&
                            (reinterpret_cast<uintptr_t>(ep.skip) & 15) == 0;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yn, 0, K * H * W * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(ep.skip ? ep.skip + (int64_t)n * K * H * W : x), 0, ep.skip ? K * H * W * 4 : 0, 0x00020000);
    // the lane index is recomputed here (it costs nothing) instead of being kept alive through the main loop
    const int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int kk = elane >> 5, nn = elane & 31;
    auto put_m = [&](int q) {
        float* mb = lds + (q & 1) * M_FLOATS + (3 * wave) * 512 + nn * 16;
#pragma unroll
        for (int xl = 0; xl < 3; ++xl)
#pragma unroll
            for (int eh = 0; eh < 2; ++eh) {
                const int e = (q & 1) * 8 + eh * 4;
                const int g = 2 * eh + kk;                       // channels 4g..4g+3 of the round
                const f32x16& a = acc[xl][q >> 1];
                *reinterpret_cast<float4*>(mb + xl * 512 + 4 * (g ^ ((nn >> 2) & 3))) = make_float4(a[e], a[e + 1], a[e + 2], a[e + 3]);
            }
    };
    constexpr int rounds = HALF ? 2 : 4;
    // Everything the stores need from memory is requested ahead of time: the wait counter of vector
    // memory operations counts loads and stores in one order, so a load issued after a round's stores
    // could only be waited for together with those stores (a write round trip per round).  Scale and
    // shift of all rounds are loaded before the first store, the residual input of round q+1 at the top
    // of round q.
    const int tile = elane & 31, kq = (2 * wave + (elane >> 5)) & 15;
    const int oy = by * BH + 4 * (tile / TX), ox = bx * BW + 4 * (tile % TX);
    const bool inside = quad_store && oy + 3 < H && ox < W;      // whole tile inside the image, rows 16-byte aligned
    float scq[rounds], shq[rounds];
#pragma unroll
    for (int q = 0; q < rounds; ++q) {
        const int k = kb * KB + q * 16 + kq;
        scq[q] = 1.0f; shq[q] = 0.0f;
        if (ep.shift && k < K) { scq[q] = ep.scale ? ep.scale[k] : 1.0f; shq[q] = ep.shift[k]; }
    }
    u32x4 skq[2][4];
    auto load_skip = [&](auto qtag) {
        constexpr int q = decltype(qtag)::value;
        const int k = kb * KB + q * 16 + kq;
        if (ep.skip && wave < 8 && inside && k < K) {
            const int voff = ((k * H + oy) * W + ox) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) skq[q & 1][r] = __builtin_amdgcn_raw_buffer_load_b128(srs, voff + r * W * 4, 0, 0);
        }
    };
    load_skip(X0{});
    put_m(0);
    __syncthreads();
    using X3 = std::integral_constant<int, 3>;
#pragma unroll
    for (int q = 0; q < rounds; ++q) {
        if (q + 1 < rounds) put_m(q + 1);
        if (q == 0 && rounds > 1) load_skip(X1{});
        if (q == 1 && rounds > 2) load_skip(X2{});
        if (q == 2 && rounds > 3) load_skip(X3{});
        if (wave < 8) {
            const int k = kb * KB + q * 16 + kq;
            const float sc = scq[q], sh = shq[q];
            const u32x4 (&sk)[4] = skq[q & 1];
            const float* mp = lds + (q & 1) * M_FLOATS + tile * 16 + 4 * ((kq >> 2) ^ ((tile >> 2) & 3)) + (kq & 3);
            const bool fast = inside && k < K;
            const int voff = ((k * H + oy) * W + ox) * 4;
            float s[6][4];                                       // A^T m, column by column
#pragma unroll
            for (int b = 0; b < 6; ++b)
                out_1d(mp[(0 + b) * 512], mp[(6 + b) * 512], mp[(12 + b) * 512], mp[(18 + b) * 512], mp[(24 + b) * 512],
                       mp[(30 + b) * 512], s[b]);
            float o[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) out_1d(s[0][r], s[1][r], s[2][r], s[3][r], s[4][r], s[5][r], o[r]);
            if (ep.stats) {
                float ps = 0.0f, pq = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (oy + r < H && ox + c < W) { ps += o[r][c]; pq += o[r][c] * o[r][c]; }
                // float32 only over the lane's own few pixels; across lanes in float64, so that the variance
                // E[x^2] - mean^2 formed from these sums survives planes whose mean dwarfs their spread
                double dps = (double)ps, dpq = (double)pq;
#pragma unroll
                for (int m2 = 16; m2 > 0; m2 >>= 1) { dps += __shfl_xor(dps, m2, VO_WAVE); dpq += __shfl_xor(dpq, m2, VO_WAVE); }
                if (tile == 0 && k < K) {
                    double* st = ep.stats + ((int64_t)n * K + k) * 2;
                    atomicAdd(st, dps);
                    atomicAdd(st + 1, dpq);
                }
            }
            if (fast) {
                // straight-line code, 16-byte accesses
                if (ep.shift) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[r][c] = fmaf(o[r][c], sc, sh);
                }
                if (ep.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[r][c] = fmaxf(o[r][c], 0.0f);
                }
                if (ep.skip) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[r][c] = fmaxf(o[r][c] + __uint_as_float(sk[r][c]), 0.0f);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    u32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = __float_as_uint(o[r][c]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, yrs, voff + r * W * 4, 0, 0);
                }
            } else if (k < K && oy < H && ox < W) {
                const int64_t off = ((int64_t)k * H + oy) * W + ox;
                float* p = yn + off;
                if (ep.shift) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[r][c] = fmaf(o[r][c], sc, sh);
                }
                if (ep.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[r][c] = fmaxf(o[r][c], 0.0f);
                }
                const float* sp = ep.skip ? ep.skip + (int64_t)n * K * H * W + off : nullptr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (oy + r >= H) break;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (ox + c < W) {
                            float v = o[r][c];
                            if (sp) v = fmaxf(v + sp[(int64_t)r * W + c], 0.0f);
                            p[(int64_t)r * W + c] = v;
                        }
                }
            }
        }
        if (q + 1 < rounds) __syncthreads();
    }
}

}  // namespace

extern "C" int vo_winograd4_filter(const float* weight, float* transformed, int K, int C, void* stream) {
    VO_REQUIRE(weight && transformed, "vo_winograd4_filter: null pointer");
    VO_REQUIRE(K > 0 && C > 0, "vo_winograd4_filter: bad sizes K=%d C=%d", K, C);
    if (C % CB != 0) return vo_fail(VO_ERR_UNSUPPORTED, "vo_winograd4_filter: C=%d is not a multiple of %d", C, CB);
    const int Kpad = (K + KB - 1) / KB * KB;
    const int64_t n = (int64_t)Kpad * C;
    filter4_kernel<<<dim3((unsigned)vo_ceil_div(n, 256)), dim3(256), 0, vo_stream(stream)>>>(weight, transformed, K, C, Kpad);
    VO_CHECK_LAUNCH("vo_winograd4_filter");
    return VO_OK;
}

extern "C" int vo_conv3x3_winograd4(const float* x, const float* transformed, float* y, int64_t y_batch_stride,
                                    const float* scale, const float* shift, int relu, const float* skip, double* stats,
                                    const double* in_stats, float in_eps, int N, int C, int H, int W, int K,
                                    void* stream) {
    VO_REQUIRE(x && transformed && y, "vo_conv3x3_winograd4: null pointer");
    VO_REQUIRE(!scale || shift, "vo_conv3x3_winograd4: scale without shift");
    VO_REQUIRE(!stats || (!shift && !relu && !skip), "vo_conv3x3_winograd4: statistics are of the raw output, no epilogue with them");
    VO_REQUIRE(skip != y, "vo_conv3x3_winograd4: skip must not alias the output");
    VO_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && K > 0, "vo_conv3x3_winograd4: bad sizes");
    if (y_batch_stride == 0) y_batch_stride = (int64_t)K * H * W;
    VO_REQUIRE(y_batch_stride >= (int64_t)K * H * W, "vo_conv3x3_winograd4: y_batch_stride smaller than one item");
    if (C % CB != 0) return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv3x3_winograd4: C=%d is not a multiple of %d", C, CB);
    if ((int64_t)C * H * W * 4 >= 0x7fffffffLL || (int64_t)K * H * W * 4 >= 0x7fffffffLL)
        return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv3x3_winograd4: one image of the input or output exceeds 2 GiB");
    if (in_stats && C > MAX_NORM_C)
        return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv3x3_winograd4: normalise-on-load supports at most %d input channels", MAX_NORM_C);
    // block shape: 16x32 pixels, or 32x16 where that covers the map with less waste
    const int64_t wide_px = (int64_t)((W + 31) / 32 * 32) * ((H + 15) / 16 * 16), tall_px = (int64_t)((W + 15) / 16 * 16) * ((H + 31) / 32 * 32);
    const bool tall = tall_px < wide_px;
    const int bx = tall ? (W + 15) / 16 : (W + 31) / 32, by = tall ? (H + 31) / 32 : (H + 15) / 16;
    const int64_t blocks = (int64_t)bx * by * N;
    const int kblocks = (K + KB - 1) / KB;
    if (blocks * kblocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_conv3x3_winograd4: grid too large");
    const Epilogue ep{scale, shift, skip, relu, y_batch_stride, stats};
    const InputNorm nrm{in_stats, in_eps};
    // a last block with at most 32 channels goes out as a launch of its own that skips the empty k-tile
    const int rem = K % KB;
    const int full = (rem >= 1 && rem <= 32) ? kblocks - 1 : kblocks;
    hipStream_t st = vo_stream(stream);
    auto launch = [&](auto shape) {
        constexpr bool TALL = decltype(shape)::value;
#define W4(NORM, HALF) wino4_kernel<NORM, HALF, TALL>
        if (full > 0) {
            const dim3 grid((unsigned)(blocks * full));
            if (in_stats) W4(true, false)<<<grid, dim3(NT), 0, st>>>(x, transformed, y, ep, nrm, C, H, W, K, bx, by, full, 0);
            else W4(false, false)<<<grid, dim3(NT), 0, st>>>(x, transformed, y, ep, nrm, C, H, W, K, bx, by, full, 0);
        }
        if (full < kblocks) {
            const dim3 grid((unsigned)blocks);
            if (in_stats) W4(true, true)<<<grid, dim3(NT), 0, st>>>(x, transformed, y, ep, nrm, C, H, W, K, bx, by, 1, full);
            else W4(false, true)<<<grid, dim3(NT), 0, st>>>(x, transformed, y, ep, nrm, C, H, W, K, bx, by, 1, full);
        }
#undef W4
    };
    if (tall) launch(std::true_type{});
    else launch(std::false_type{});
    VO_CHECK_LAUNCH("vo_conv3x3_winograd4");
    return VO_OK;
}
