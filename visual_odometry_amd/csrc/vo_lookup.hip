// Windowed correlation lookup, called once per refinement iteration.
//
// All (2r+1)^2 taps of one pixel at one level share the same fractional offset
// (the window steps are integers), so the window is the bilinear blend of ONE
// (2r+2)^2 integer patch: each lane gathers that patch row by row, blends it
// horizontally once, and emits the (2r+1)^2 outputs with two FMAs each instead
// of four gathered taps per output.
//
// One lane per (pair, level, pixel), lanes along x: the (2r+1)^2 output planes are
// written fully coalesced (256 B per wave store); the patch reads are 40-byte
// row segments of that pixel's own slice of the volume (gather, served by L2).
// Zero padding outside the slice and pixel-centre coordinates, i.e.
// grid_sample(align_corners=True, padding_mode="zeros") without the normalise /
// un-normalise round trip.
#include "vo_common.h"

namespace {

struct Levels {
    const float* p[VO_MAX_LEVELS];
    int h[VO_MAX_LEVELS];
    int w[VO_MAX_LEVELS];
};

// Level selection by compare/select: indexing the kernel-argument arrays with a
// per-lane level would spill the struct to scratch memory.
__device__ __forceinline__ void pick_level(const Levels& lv, int l, const float*& p, int& h, int& w) {
    p = lv.p[0]; h = lv.h[0]; w = lv.w[0];
    if (l == 1) { p = lv.p[1]; h = lv.h[1]; w = lv.w[1]; }
    if (l == 2) { p = lv.p[2]; h = lv.h[2]; w = lv.w[2]; }
    if (l == 3) { p = lv.p[3]; h = lv.h[3]; w = lv.w[3]; }
}

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));

// floor() of a coordinate as an int that is safe for any input (NaN, inf, huge)
__device__ __forceinline__ int safe_floor(float v, float& frac) {
    const float c = fminf(fmaxf(v, -1048576.0f), 1048576.0f);   // NaN -> -2^20: far outside
    const float f = floorf(c);
    frac = c - f;
    return (int)f;
}

// one patch row of 2R+2 floats starting at column xs (zero outside [0, w))
template <int R>
__device__ __forceinline__ void load_row(const float* __restrict__ row, int xs, int w, bool row_ok, float (&v)[2 * R + 2]) {
    constexpr int T = 2 * R + 2;
    if (!row_ok) {
#pragma unroll
        for (int i = 0; i < T; ++i) v[i] = 0.0f;
        return;
    }
    if (xs >= 0 && xs + T <= w) {                        // common case: whole segment inside
        if constexpr (T == 10) {
            const float4u a = *reinterpret_cast<const float4u*>(row + xs);
            const float4u b = *reinterpret_cast<const float4u*>(row + xs + 4);
            const float2u c = *reinterpret_cast<const float2u*>(row + xs + 8);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
            v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            v[8] = c.x; v[9] = c.y;
        } else {
#pragma unroll
            for (int i = 0; i < T; ++i) v[i] = row[xs + i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int x = xs + i;
            v[i] = (x >= 0 && x < w) ? row[x] : 0.0f;
        }
    }
}

// RPT = window rows (y offsets) per lane.  RPT = 2r+1 gives one lane per (pair,
// level, pixel); smaller RPT splits a pixel's window over (2r+1)/RPT lanes, which
// re-reads the shared patch rows from L1 but multiplies the waves in flight.
template <int R, int RPT>
__global__ __launch_bounds__(256) void lookup2d_kernel(Levels lv, const float* __restrict__ coords,
                                                       float* __restrict__ out, int64_t out_bs, int64_t total, int HW, int levels) {
    constexpr int WIN = 2 * R + 1, T = 2 * R + 2, PARTS = WIN / RPT;
    static_assert(WIN % RPT == 0, "rows per lane must divide the window");
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int p = (int)(t % HW);
    const int part = (int)((t / HW) % PARTS);
    const int l = (int)((t / ((int64_t)HW * PARTS)) % levels);
    const int64_t b = t / ((int64_t)HW * PARTS * levels);

    const float inv = 1.0f / (float)(1 << l);                       // exact power of two
    float fx, fy;
    const int x0 = safe_floor(coords[(b * 2 + 0) * HW + p] * inv, fx);
    const int y0 = safe_floor(coords[(b * 2 + 1) * HW + p] * inv, fy);
    int h, w;
    const float* base;
    pick_level(lv, l, base, h, w);
    const float* img = base + (b * HW + p) * (int64_t)h * w;
    float* o = out + b * out_bs + l * (int64_t)(WIN * WIN) * HW + p;

    const float wx1 = fx, wx0 = 1.0f - fx, wy1 = fy, wy0 = 1.0f - fy;
    const int j0 = part * RPT;
    float prev[WIN], cur[WIN], v[T];
#pragma unroll
    for (int jj = 0; jj <= RPT; ++jj) {
        const int j = j0 + jj;
        const int y = y0 - R + j;
        load_row<R>(img + (int64_t)y * w, x0 - R, w, y >= 0 && y < h, v);
#pragma unroll
        for (int a = 0; a < WIN; ++a) cur[a] = wx0 * v[a] + wx1 * v[a + 1];
        if (jj > 0) {
            // window row j-1 is the y-offset index; the x-offset index a is the slow
            // channel index: channel = a*(2r+1) + (j-1)
#pragma unroll
            for (int a = 0; a < WIN; ++a) o[(int64_t)(a * WIN + (j - 1)) * HW] = wy0 * prev[a] + wy1 * cur[a];
        }
#pragma unroll
        for (int a = 0; a < WIN; ++a) prev[a] = cur[a];
    }
}

template <int R>
__global__ __launch_bounds__(256) void lookup1d_kernel(Levels lv, const float* __restrict__ coords,
                                                       float* __restrict__ out, int64_t out_bs, int64_t total, int HW, int levels) {
    constexpr int WIN = 2 * R + 1, T = 2 * R + 2;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int p = (int)(t % HW);
    const int l = (int)((t / HW) % levels);
    const int64_t b = t / ((int64_t)HW * levels);

    const float inv = 1.0f / (float)(1 << l);
    float fx;
    const int x0 = safe_floor(coords[(b * 2 + 0) * HW + p] * inv, fx);
    int h, w;
    const float* base;
    pick_level(lv, l, base, h, w);
    const float* row = base + (b * HW + p) * (int64_t)w;
    float v[T];
    load_row<R>(row, x0 - R, w, true, v);
    float* o = out + b * out_bs + l * (int64_t)WIN * HW + p;
    const float wx1 = fx, wx0 = 1.0f - fx;
#pragma unroll
    for (int a = 0; a < WIN; ++a) o[(int64_t)a * HW] = wx0 * v[a] + wx1 * v[a + 1];
}

int check_common(const char* fn, const float* const* lp, const float* coords, float* out, int batch, int H, int W,
                 int levels, int radius) {
    if (!coords || !out) return vo_fail(VO_ERR_INVALID_ARG, "%s: null pointer", fn);
    if (batch <= 0 || H <= 0 || W <= 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: bad sizes batch=%d H=%d W=%d", fn, batch, H, W);
    if (levels < 1 || levels > VO_MAX_LEVELS) return vo_fail(VO_ERR_INVALID_ARG, "%s: levels=%d outside 1..%d", fn, levels, VO_MAX_LEVELS);
    for (int i = 0; i < levels; ++i)
        if (!lp[i]) return vo_fail(VO_ERR_INVALID_ARG, "%s: pyramid level %d is null", fn, i);
    if (radius != 3 && radius != 4) return vo_fail(VO_ERR_UNSUPPORTED, "%s: radius %d (built for 3 and 4)", fn, radius);
    if ((int64_t)H * W > 0x3fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "%s: image too large", fn);
    return VO_OK;
}

}  // namespace

extern "C" int vo_corr_lookup_2d(const float* l0, const float* l1, const float* l2, const float* l3,
                                 const float* coords, float* out, int64_t out_batch_stride, int batch, int H, int W, int H2,
                                 int W2, int levels, int radius, void* stream) {
    const float* lp[VO_MAX_LEVELS] = {l0, l1, l2, l3};
    int rc = check_common("vo_corr_lookup_2d", lp, coords, out, batch, H, W, levels, radius);
    if (rc) return rc;
    VO_REQUIRE(H2 > 0 && W2 > 0, "vo_corr_lookup_2d: bad target size H2=%d W2=%d", H2, W2);
    Levels lv;
    int h = H2, w = W2;
    for (int i = 0; i < VO_MAX_LEVELS; ++i) {
        lv.p[i] = (i < levels) ? lp[i] : nullptr; lv.h[i] = h; lv.w[i] = w;
        h /= 2; w /= 2;
    }
    VO_REQUIRE(lv.h[levels - 1] > 0 && lv.w[levels - 1] > 0, "vo_corr_lookup_2d: level %d is empty", levels - 1);
    const int HW = H * W;
    const int64_t out_bs = out_batch_stride ? out_batch_stride : (int64_t)levels * (2 * radius + 1) * (2 * radius + 1) * HW;
    VO_REQUIRE(out_bs >= (int64_t)levels * (2 * radius + 1) * (2 * radius + 1) * HW, "vo_corr_lookup_2d: out_batch_stride smaller than one item");
    // One lane per (pair, level, pixel).  Splitting a window over 3 or 9 lanes was
    // measured slower (58 / 70 us against 52 us at 8 pairs of 640x480): the kernel is
    // bound by the cache lines it pulls in, and split lanes pull shared rows twice.
    const int64_t total = (int64_t)batch * levels * HW;
    const int64_t blocks = vo_ceil_div(total, 256);
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_corr_lookup_2d: grid too large");
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = vo_stream(stream);
    if (radius == 4)
        lookup2d_kernel<4, 9><<<grid, block, 0, st>>>(lv, coords, out, out_bs, total, HW, levels);
    else
        lookup2d_kernel<3, 7><<<grid, block, 0, st>>>(lv, coords, out, out_bs, total, HW, levels);
    VO_CHECK_LAUNCH("vo_corr_lookup_2d");
    return VO_OK;
}

extern "C" int vo_corr_lookup_1d(const float* l0, const float* l1, const float* l2, const float* l3,
                                 const float* coords, float* out, int64_t out_batch_stride, int batch, int H, int W, int W2,
                                 int levels, int radius, void* stream) {
    const float* lp[VO_MAX_LEVELS] = {l0, l1, l2, l3};
    int rc = check_common("vo_corr_lookup_1d", lp, coords, out, batch, H, W, levels, radius);
    if (rc) return rc;
    VO_REQUIRE(W2 > 0, "vo_corr_lookup_1d: bad target width W2=%d", W2);
    Levels lv;
    int w = W2;
    for (int i = 0; i < VO_MAX_LEVELS; ++i) {
        lv.p[i] = (i < levels) ? lp[i] : nullptr; lv.h[i] = 1; lv.w[i] = w;
        w /= 2;
    }
    VO_REQUIRE(lv.w[levels - 1] > 0, "vo_corr_lookup_1d: level %d is empty", levels - 1);
    const int HW = H * W;
    const int64_t out_bs = out_batch_stride ? out_batch_stride : (int64_t)levels * (2 * radius + 1) * HW;
    VO_REQUIRE(out_bs >= (int64_t)levels * (2 * radius + 1) * HW, "vo_corr_lookup_1d: out_batch_stride smaller than one item");
    const int64_t total = (int64_t)batch * levels * HW;
    const int64_t blocks = vo_ceil_div(total, 256);
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_corr_lookup_1d: grid too large");
    if (radius == 4)
        lookup1d_kernel<4><<<dim3((unsigned)blocks), dim3(256), 0, vo_stream(stream)>>>(lv, coords, out, out_bs, total, HW, levels);
    else
        lookup1d_kernel<3><<<dim3((unsigned)blocks), dim3(256), 0, vo_stream(stream)>>>(lv, coords, out, out_bs, total, HW, levels);
    VO_CHECK_LAUNCH("vo_corr_lookup_1d");
    return VO_OK;
}
