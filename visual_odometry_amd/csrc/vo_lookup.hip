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
// All addressing is 32-bit buffer addressing relative to the workgroup's slices (no
// 64-bit address arithmetic per element, no branch per lane at the slice's edges).
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

// Level selection by compare/select on constant indices: indexing the kernel-argument arrays with a variable would
// copy the struct to scratch memory.  `l` is uniform, so these are scalar selects.
__device__ __forceinline__ void pick_level(const Levels& lv, int l, const float*& p, int& h, int& w) {
    p = lv.p[0]; h = lv.h[0]; w = lv.w[0];
    if (l == 1) { p = lv.p[1]; h = lv.h[1]; w = lv.w[1]; }
    if (l == 2) { p = lv.p[2]; h = lv.h[2]; w = lv.w[2]; }
    if (l == 3) { p = lv.p[3]; h = lv.h[3]; w = lv.w[3]; }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int PB = 256;                                   // pixels (lanes) per workgroup
constexpr int OUTSIDE = (int)0x80000000;                  // a buffer offset past every extent: loads give 0, stores are dropped

// floor() of a coordinate as an int that is safe for any input (NaN, inf, huge)
__device__ __forceinline__ int safe_floor(float v, float& frac) {
    const float c = fminf(fmaxf(v, -1048576.0f), 1048576.0f);   // NaN -> -2^20: far outside
    const float f = floorf(c);
    frac = c - f;
    return (int)f;
}

// One patch row of 2R+2 floats starting at column xs of the row at byte offset `row_off` (relative to the lane's slice
// at `pix_off`) of the buffer `rs`; zero outside [0, w) and for rows outside the slice.  No branch per lane: a segment
// that lies inside its row is three vector loads (row_fast), everything else gets an offset outside the buffer; only when
// some lane of the wave has a segment that crosses a row end does the wave add ten single loads for those lanes
// (row_edges; the two results are OR-ed: one of them is all zero bits).  Two steps, so that a caller can request the
// next row's vector loads before it finishes the current row.
template <int R>
__device__ __forceinline__ void row_fast(const __amdgpu_buffer_rsrc_t& rs, int seg, bool whole, unsigned (&u)[2 * R + 2]) {
    constexpr int T = 2 * R + 2;
    if constexpr (T == 10) {
        const int off = whole ? seg : OUTSIDE;
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
        const u32x2 c = __builtin_amdgcn_raw_buffer_load_b64(rs, off + 32, 0, 0);
        u[0] = a[0]; u[1] = a[1]; u[2] = a[2]; u[3] = a[3]; u[4] = b[0]; u[5] = b[1]; u[6] = b[2]; u[7] = b[3]; u[8] = c[0]; u[9] = c[1];
    } else {
#pragma unroll
        for (int i = 0; i < T; ++i) u[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, whole ? seg + 4 * i : OUTSIDE, 0, 0);
    }
}
template <int R>
__device__ __forceinline__ void row_edges(const __amdgpu_buffer_rsrc_t& rs, int seg, int xs, int w, bool partial, unsigned (&u)[2 * R + 2]) {
    constexpr int T = 2 * R + 2;
    if (__builtin_amdgcn_ballot_w64(partial) != 0) {
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int x = xs + i;
            u[i] |= __builtin_amdgcn_raw_buffer_load_b32(rs, (partial && x >= 0 && x < w) ? seg + 4 * i : OUTSIDE, 0, 0);
        }
    }
}

// One lane per (pair, level, pixel); a workgroup = PB consecutive pixels of one (pair, level), so the level's size and
// the base of the workgroup's slices are uniform: reads go through one buffer over those PB slices (32-bit offsets, the
// range check supplies the zeros), the (2r+1)^2 output planes through one buffer over the pair's output with the plane
// part of the address in the instruction's scalar offset.
template <int R>
__global__ __launch_bounds__(PB) void lookup2d_kernel(Levels lv, const float* __restrict__ coords, float* __restrict__ out,
                                                      int64_t out_bs, int HW, int levels, int blocks_per_image) {
    constexpr int WIN = 2 * R + 1, T = 2 * R + 2;
    const int bl = blockIdx.x / blocks_per_image, l = bl % levels, b = bl / levels;
    const int p0 = (blockIdx.x - bl * blocks_per_image) * PB, lane_p = threadIdx.x, p = p0 + lane_p;
    const bool live = p < HW;
    int h, w;
    const float* lbase;
    pick_level(lv, l, lbase, h, w);
    const int hw4 = h * w * 4;
    const float* base = lbase + ((int64_t)b * HW + p0) * (int64_t)h * w;
    const int npix = min(PB, HW - p0);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, npix * hw4, 0x00020000);
    const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(out + (int64_t)b * out_bs, 0, levels * WIN * WIN * HW * 4, 0x00020000);

    const float inv = 1.0f / (float)(1 << l);                           // exact power of two
    const int pc = live ? p : HW - 1;
    float fx, fy;
    const int x0 = safe_floor(coords[((int64_t)b * 2 + 0) * HW + pc] * inv, fx);
    const int y0 = safe_floor(coords[((int64_t)b * 2 + 1) * HW + pc] * inv, fy);
    const float wx1 = fx, wx0 = 1.0f - fx, wy1 = fy, wy0 = 1.0f - fy;
    const int pix_off = lane_p * hw4;
    const int ovo = live ? p * 4 : OUTSIDE;
    const int plane0 = l * WIN * WIN * HW * 4;                            // this level's first output plane

    const int xs = x0 - R;
    const bool inside = xs >= 0 && xs + T <= w;
    const int seg0 = pix_off + ((y0 - R) * w + xs) * 4, row4 = w * 4;
    auto row_ok = [&](int j) { const int y = y0 - R + j; return live && y >= 0 && y < h; };
    float prev[WIN], cur[WIN];
    unsigned u[T], un[T];
    row_fast<R>(rs, seg0, row_ok(0) && inside, u);
#pragma unroll
    for (int j = 0; j <= WIN; ++j) {
        if (j < WIN) row_fast<R>(rs, seg0 + (j + 1) * row4, row_ok(j + 1) && inside, un);     // in flight while row j is finished
        row_edges<R>(rs, seg0 + j * row4, xs, w, row_ok(j) && !inside, u);
#pragma unroll
        for (int a = 0; a < WIN; ++a) cur[a] = wx0 * __uint_as_float(u[a]) + wx1 * __uint_as_float(u[a + 1]);
        if (j > 0) {
            // window row j-1 is the y-offset index; the x-offset index a is the slow channel index: channel = a*(2r+1) + (j-1)
#pragma unroll
            for (int a = 0; a < WIN; ++a)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(wy0 * prev[a] + wy1 * cur[a]), os, ovo, plane0 + (a * WIN + (j - 1)) * HW * 4, 0);
        }
#pragma unroll
        for (int a = 0; a < WIN; ++a) prev[a] = cur[a];
#pragma unroll
        for (int i = 0; i < T; ++i) u[i] = un[i];
    }
}

template <int R>
__global__ __launch_bounds__(PB) void lookup1d_kernel(Levels lv, const float* __restrict__ coords, float* __restrict__ out,
                                                      int64_t out_bs, int HW, int levels, int blocks_per_image) {
    constexpr int WIN = 2 * R + 1, T = 2 * R + 2;
    const int bl = blockIdx.x / blocks_per_image, l = bl % levels, b = bl / levels;
    const int p0 = (blockIdx.x - bl * blocks_per_image) * PB, lane_p = threadIdx.x, p = p0 + lane_p;
    const bool live = p < HW;
    int h, w;
    const float* lbase;
    pick_level(lv, l, lbase, h, w);
    const float* base = lbase + ((int64_t)b * HW + p0) * (int64_t)w;
    const int npix = min(PB, HW - p0);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, npix * w * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(out + (int64_t)b * out_bs, 0, levels * WIN * HW * 4, 0x00020000);
    const float inv = 1.0f / (float)(1 << l);
    float fx;
    const int x0 = safe_floor(coords[((int64_t)b * 2 + 0) * HW + (live ? p : HW - 1)] * inv, fx);
    const int xs = x0 - R, seg = lane_p * w * 4 + xs * 4;
    const bool inside = xs >= 0 && xs + T <= w;
    unsigned u[T];
    row_fast<R>(rs, seg, live && inside, u);
    row_edges<R>(rs, seg, xs, w, live && !inside, u);
    float v[T];
#pragma unroll
    for (int i = 0; i < T; ++i) v[i] = __uint_as_float(u[i]);
    const float wx1 = fx, wx0 = 1.0f - fx;
    const int ovo = live ? p * 4 : OUTSIDE;
#pragma unroll
    for (int a = 0; a < WIN; ++a)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(wx0 * v[a] + wx1 * v[a + 1]), os, ovo, (l * WIN + a) * HW * 4, 0);
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
    VO_REQUIRE((int64_t)PB * H2 * W2 * 4 < 0x7fffffffLL && (int64_t)levels * (2 * radius + 1) * (2 * radius + 1) * HW * 4 < 0x7fffffffLL,
               "vo_corr_lookup_2d: a workgroup's slices or one pair's output exceed 2 GiB");
    const int per_image = (int)vo_ceil_div(HW, PB);
    const int64_t blocks = (int64_t)per_image * batch * levels;
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_corr_lookup_2d: grid too large");
    const dim3 grid((unsigned)blocks), block(PB);
    hipStream_t st = vo_stream(stream);
    if (radius == 4)
        lookup2d_kernel<4><<<grid, block, 0, st>>>(lv, coords, out, out_bs, HW, levels, per_image);
    else
        lookup2d_kernel<3><<<grid, block, 0, st>>>(lv, coords, out, out_bs, HW, levels, per_image);
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
    VO_REQUIRE((int64_t)PB * W2 * 4 < 0x7fffffffLL && (int64_t)levels * (2 * radius + 1) * HW * 4 < 0x7fffffffLL,
               "vo_corr_lookup_1d: a workgroup's rows or one pair's output exceed 2 GiB");
    const int per_image = (int)vo_ceil_div(HW, PB);
    const int64_t blocks = (int64_t)per_image * batch * levels;
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_corr_lookup_1d: grid too large");
    const dim3 grid((unsigned)blocks), block(PB);
    if (radius == 4)
        lookup1d_kernel<4><<<grid, block, 0, vo_stream(stream)>>>(lv, coords, out, out_bs, HW, levels, per_image);
    else
        lookup1d_kernel<3><<<grid, block, 0, vo_stream(stream)>>>(lv, coords, out, out_bs, HW, levels, per_image);
    VO_CHECK_LAUNCH("vo_corr_lookup_1d");
    return VO_OK;
}
