// The 2-D correlation pyramid in a TILED target order, and the windowed lookup that reads it.
//
// A lookup needs, per (source pixel, level), a 10x10 patch of that pixel's own target image.  In the
// row-major volume each of the ten 40-byte row segments costs one or two 128-byte lines (about 13 lines
// per patch); with the target image cut into tiles of 8 columns x 4 rows -- 32 floats, exactly one
// line -- a patch touches 2-3 tiles across and 3-4 down (about 7 lines).  Lanes cannot share lines:
// neighbouring source pixels own different target images.
//
// Layout of one source pixel's slice at a level of true size h x w:
//   tiles_x = ceil(w / 8), tiles_y = ceil(h / 4), slice = tiles_y * tiles_x * 32 floats,
//   element (y, x) at ((y >> 2) * tiles_x + (x >> 3)) * 32 + (y & 3) * 8 + (x & 7).
// Level 0 comes straight out of the correlation GEMM: its columns are target pixels, so giving the GEMM
// the second feature map with its rows in tiled order (vo_tile_feature_rows; zero rows where a tile
// reaches past the image) makes the volume tiled at no cost.  Entries of a tile that lie outside the true
// size are never read as values: the lookup masks by (h, w).
#include "vo_common.h"

namespace {

struct TiledLevels {
    const float* p[VO_MAX_LEVELS];
    int h[VO_MAX_LEVELS], w[VO_MAX_LEVELS], tx[VO_MAX_LEVELS];
    int64_t slice[VO_MAX_LEVELS];
};

__host__ __device__ inline int tiles_x(int w) { return (w + 7) >> 3; }
__host__ __device__ inline int tiles_y(int h) { return (h + 3) >> 2; }

// fmap (B, H*W, D) -> out (B, tiles*32, D): row n' of the output is the feature vector of the target pixel
// that sits at tiled position n', or zeros.  One lane per float4.
__global__ __launch_bounds__(256) void tile_rows_kernel(const float* __restrict__ fmap, float* __restrict__ out, int64_t total4,
                                                        int H, int W, int D4, int tx) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    const int d = (int)(t % D4);
    const int64_t r = t / D4;
    const int64_t per = (int64_t)tiles_y(H) * tx * 32;
    const int np = (int)(r % per);
    const int64_t b = r / per;
    const int tile = np >> 5, y = (tile / tx) * 4 + ((np >> 3) & 3), x = (tile % tx) * 8 + (np & 7);
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (y < H && x < W) v = reinterpret_cast<const float4*>(fmap)[((b * H + y) * W + x) * D4 + d];
    reinterpret_cast<float4*>(out)[t] = v;
}

// Levels 1..3 from level 0 in one pass, all tiled.  One lane per 8x8 block of level 0 = two tiles, one above
// the other; each tile is 128 contiguous bytes.
struct PyrDims { int h[4], w[4], tx[4]; int64_t slice[4]; };

__global__ __launch_bounds__(256) void pyramid2d_tiled_kernel(const float* __restrict__ l0, float* __restrict__ l1,
                                                              float* __restrict__ l2, float* __restrict__ l3, int64_t total,
                                                              PyrDims d, int hb, int wb) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int bx = (int)(t % wb);
    const int by = (int)((t / wb) % hb);
    const int64_t img = t / ((int64_t)wb * hb);
    const float* src = l0 + img * d.slice[0];
    const int ty0 = tiles_y(d.h[0]);

    // level-1 block a[r][c] = mean of level-0 (8by + 2r .. +1, 8bx + 2c .. +1): rows 2r, 2r+1 of tile (2by + r/2, bx)
    float a[4][4];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int ty = 2 * by + half;
        float4 q[8];
        if (ty < ty0) {
            const float4* p = reinterpret_cast<const float4*>(src + ((int64_t)ty * d.tx[0] + bx) * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = p[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {                         // tile rows 2rr, 2rr+1 -> level-1 row 2*half + rr
            const float4 t0 = q[4 * rr], t1 = q[4 * rr + 1], u0 = q[4 * rr + 2], u1 = q[4 * rr + 3];
            float* o = a[2 * half + rr];
            o[0] = (((t0.x + t0.y) + u0.x) + u0.y) * 0.25f;
            o[1] = (((t0.z + t0.w) + u0.z) + u0.w) * 0.25f;
            o[2] = (((t1.x + t1.y) + u1.x) + u1.y) * 0.25f;
            o[3] = (((t1.z + t1.w) + u1.z) + u1.w) * 0.25f;
        }
    }
    if (l1) {
        // level-1 rows 4by + r, columns 4bx .. 4bx+3: tile (by, bx >> 1), tile row r, tile columns 4*(bx & 1) ..
        if (by < tiles_y(d.h[1]) && (bx >> 1) < d.tx[1]) {
            float* dst = l1 + img * d.slice[1] + ((int64_t)by * d.tx[1] + (bx >> 1)) * 32 + 4 * (bx & 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<float4*>(dst + r * 8) = make_float4(a[r][0], a[r][1], a[r][2], a[r][3]);
        }
    }
    if (!l2) return;
    float b[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
            b[r][c] = (((a[2 * r][2 * c] + a[2 * r][2 * c + 1]) + a[2 * r + 1][2 * c]) + a[2 * r + 1][2 * c + 1]) * 0.25f;
    // level-2 rows 2by + r, columns 2bx + c: tile (by >> 1, bx >> 2), tile row 2*(by & 1) + r, tile column 2*(bx & 3) + c
    if ((by >> 1) < tiles_y(d.h[2]) && (bx >> 2) < d.tx[2]) {
        float* dst = l2 + img * d.slice[2] + ((int64_t)(by >> 1) * d.tx[2] + (bx >> 2)) * 32 + (2 * (by & 1)) * 8 + 2 * (bx & 3);
#pragma unroll
        for (int r = 0; r < 2; ++r) *reinterpret_cast<float2*>(dst + r * 8) = make_float2(b[r][0], b[r][1]);
    }
    if (!l3) return;
    if ((by >> 2) < tiles_y(d.h[3]) && (bx >> 3) < d.tx[3])
        l3[img * d.slice[3] + ((int64_t)(by >> 2) * d.tx[3] + (bx >> 3)) * 32 + (by & 3) * 8 + (bx & 7)] =
            (((b[0][0] + b[0][1]) + b[1][0]) + b[1][1]) * 0.25f;
}

__device__ __forceinline__ void pick(const TiledLevels& lv, int l, const float*& p, int& h, int& w, int& tx, int64_t& slice) {
    p = lv.p[0]; h = lv.h[0]; w = lv.w[0]; tx = lv.tx[0]; slice = lv.slice[0];
    if (l == 1) { p = lv.p[1]; h = lv.h[1]; w = lv.w[1]; tx = lv.tx[1]; slice = lv.slice[1]; }
    if (l == 2) { p = lv.p[2]; h = lv.h[2]; w = lv.w[2]; tx = lv.tx[2]; slice = lv.slice[2]; }
    if (l == 3) { p = lv.p[3]; h = lv.h[3]; w = lv.w[3]; tx = lv.tx[3]; slice = lv.slice[3]; }
}

__device__ __forceinline__ int safe_floor(float v, float& frac) {
    const float c = fminf(fmaxf(v, -1048576.0f), 1048576.0f);   // NaN -> -2^20: far outside
    const float f = floorf(c);
    frac = c - f;
    return (int)f;
}

// One lane per (pair, level, pixel), as the row-major lookup: the (2R+2)^2 integer patch, blended once
// horizontally per row and once vertically per output.  A patch row is fetched as up to four ALIGNED
// groups of four floats (each inside one tile row); the 2R+2 values start 0..3 floats into the first group.
template <int R>
__global__ __launch_bounds__(256) void lookup2d_tiled_kernel(TiledLevels lv, const float* __restrict__ coords,
                                                             float* __restrict__ out, int64_t out_bs, int64_t total, int HW,
                                                             int levels) {
    constexpr int WIN = 2 * R + 1, T = 2 * R + 2;
    static_assert(T <= 13, "a patch row must fit four aligned groups");
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int p = (int)(t % HW);
    const int l = (int)((t / HW) % levels);
    const int64_t b = t / ((int64_t)HW * levels);

    const float inv = 1.0f / (float)(1 << l);
    float fx, fy;
    const int x0 = safe_floor(coords[(b * 2 + 0) * HW + p] * inv, fx);
    const int y0 = safe_floor(coords[(b * 2 + 1) * HW + p] * inv, fy);
    int h, w, tx;
    int64_t slice;
    const float* base;
    pick(lv, l, base, h, w, tx, slice);
    const float* img = base + (b * HW + p) * slice;
    float* o = out + b * out_bs + l * (int64_t)(WIN * WIN) * HW + p;

    const float wx1 = fx, wx0 = 1.0f - fx, wy1 = fy, wy0 = 1.0f - fy;
    const int xs = x0 - R;                                       // first patch column
    const int g0 = xs >> 2, off = xs & 3;                        // arithmetic shift: floor for negative columns too
    float prev[WIN], cur[WIN];
#pragma unroll
    for (int j = 0; j <= WIN; ++j) {
        const int y = y0 - R + j;
        const bool row_ok = y >= 0 && y < h;
        float q[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int gg = g0 + g;
            // the fourth group is needed only when the patch starts three floats into the first
            const bool ok = row_ok && gg >= 0 && 4 * gg < w && (g < 3 || off == 3);
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (ok) v = *reinterpret_cast<const float4*>(img + ((int64_t)(y >> 2) * tx + (gg >> 1)) * 32 + (y & 3) * 8 + (gg & 1) * 4);
            const int xb = 4 * gg;
            q[4 * g + 0] = xb + 0 < w ? v.x : 0.0f;              // tile columns past the image are not values
            q[4 * g + 1] = xb + 1 < w ? v.y : 0.0f;
            q[4 * g + 2] = xb + 2 < w ? v.z : 0.0f;
            q[4 * g + 3] = xb + 3 < w ? v.w : 0.0f;
        }
        float v[T];
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const float s01 = (off & 1) ? q[i + 1] : q[i], s23 = (off & 1) ? q[i + 3] : q[i + 2];
            v[i] = (off & 2) ? s23 : s01;
        }
#pragma unroll
        for (int a = 0; a < WIN; ++a) cur[a] = wx0 * v[a] + wx1 * v[a + 1];
        if (j > 0) {
#pragma unroll
            for (int a = 0; a < WIN; ++a) o[(int64_t)(a * WIN + (j - 1)) * HW] = wy0 * prev[a] + wy1 * cur[a];
        }
#pragma unroll
        for (int a = 0; a < WIN; ++a) prev[a] = cur[a];
    }
}

}  // namespace

extern "C" int64_t vo_tiled_slice_elements(int H2, int W2) {
    if (H2 <= 0 || W2 <= 0) return 0;
    return (int64_t)tiles_y(H2) * tiles_x(W2) * 32;
}

extern "C" int vo_tile_feature_rows(const float* fmap, float* out, int batch, int H, int W, int D, void* stream) {
    VO_REQUIRE(fmap && out, "vo_tile_feature_rows: null pointer");
    VO_REQUIRE(batch > 0 && H > 0 && W > 0 && D > 0, "vo_tile_feature_rows: bad sizes");
    if (D % 4 != 0 || ((reinterpret_cast<uintptr_t>(fmap) | reinterpret_cast<uintptr_t>(out)) & 15) != 0)
        return vo_fail(VO_ERR_UNSUPPORTED, "vo_tile_feature_rows: needs D %% 4 == 0 and 16-byte aligned tensors");
    const int64_t total4 = (int64_t)batch * vo_tiled_slice_elements(H, W) * (D / 4);
    const int64_t blocks = vo_ceil_div(total4, 256);
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_tile_feature_rows: grid too large");
    tile_rows_kernel<<<dim3((unsigned)blocks), dim3(256), 0, vo_stream(stream)>>>(fmap, out, total4, H, W, D / 4, tiles_x(W));
    VO_CHECK_LAUNCH("vo_tile_feature_rows");
    return VO_OK;
}

extern "C" int vo_corr_pyramid_2d_tiled(const float* l0, float* l1, float* l2, float* l3, int64_t images, int H2, int W2,
                                        void* stream) {
    VO_REQUIRE(l0, "vo_corr_pyramid_2d_tiled: level 0 is null");
    VO_REQUIRE(images > 0 && H2 > 0 && W2 > 0, "vo_corr_pyramid_2d_tiled: bad sizes images=%lld H2=%d W2=%d", (long long)images, H2, W2);
    VO_REQUIRE(!(l2 && !l1) && !(l3 && !l2), "vo_corr_pyramid_2d_tiled: levels must be given without gaps");
    if (!l1) return VO_OK;
    if (((reinterpret_cast<uintptr_t>(l0) | reinterpret_cast<uintptr_t>(l1) | reinterpret_cast<uintptr_t>(l2)) & 15) != 0)
        return vo_fail(VO_ERR_UNSUPPORTED, "vo_corr_pyramid_2d_tiled: levels must be 16-byte aligned");
    PyrDims d;
    d.h[0] = H2; d.w[0] = W2;
    for (int i = 1; i < 4; ++i) { d.h[i] = d.h[i - 1] / 2; d.w[i] = d.w[i - 1] / 2; }
    for (int i = 0; i < 4; ++i) { d.tx[i] = tiles_x(d.w[i]); d.slice[i] = vo_tiled_slice_elements(d.h[i], d.w[i]); }
    VO_REQUIRE(d.h[1] > 0 && d.w[1] > 0, "vo_corr_pyramid_2d_tiled: level 1 is empty");
    VO_REQUIRE(!(l2 && (d.h[2] <= 0 || d.w[2] <= 0)) && !(l3 && (d.h[3] <= 0 || d.w[3] <= 0)), "vo_corr_pyramid_2d_tiled: a requested level is empty");
    const int hb = (int)vo_ceil_div(H2, 8), wb = tiles_x(W2);
    const int64_t total = images * hb * wb;
    const int64_t blocks = vo_ceil_div(total, 256);
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_corr_pyramid_2d_tiled: grid too large");
    pyramid2d_tiled_kernel<<<dim3((unsigned)blocks), dim3(256), 0, vo_stream(stream)>>>(l0, l1, l2, l3, total, d, hb, wb);
    VO_CHECK_LAUNCH("vo_corr_pyramid_2d_tiled");
    return VO_OK;
}

extern "C" int vo_corr_lookup_2d_tiled(const float* l0, const float* l1, const float* l2, const float* l3, const float* coords,
                                       float* out, int64_t out_batch_stride, int batch, int H, int W, int H2, int W2, int levels,
                                       int radius, void* stream) {
    const float* lp[VO_MAX_LEVELS] = {l0, l1, l2, l3};
    VO_REQUIRE(coords && out, "vo_corr_lookup_2d_tiled: null pointer");
    VO_REQUIRE(batch > 0 && H > 0 && W > 0 && H2 > 0 && W2 > 0, "vo_corr_lookup_2d_tiled: bad sizes");
    VO_REQUIRE(levels >= 1 && levels <= VO_MAX_LEVELS, "vo_corr_lookup_2d_tiled: levels=%d outside 1..%d", levels, VO_MAX_LEVELS);
    if (radius != 3 && radius != 4) return vo_fail(VO_ERR_UNSUPPORTED, "vo_corr_lookup_2d_tiled: radius %d (built for 3 and 4)", radius);
    if ((int64_t)H * W > 0x3fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_corr_lookup_2d_tiled: image too large");
    TiledLevels lv;
    int h = H2, w = W2;
    for (int i = 0; i < VO_MAX_LEVELS; ++i) {
        lv.p[i] = (i < levels) ? lp[i] : nullptr; lv.h[i] = h; lv.w[i] = w; lv.tx[i] = tiles_x(w > 0 ? w : 1);
        lv.slice[i] = vo_tiled_slice_elements(h, w);
        if (i < levels) {
            VO_REQUIRE(lp[i], "vo_corr_lookup_2d_tiled: pyramid level %d is null", i);
            VO_REQUIRE(h > 0 && w > 0, "vo_corr_lookup_2d_tiled: level %d is empty", i);
            if ((reinterpret_cast<uintptr_t>(lp[i]) & 15) != 0) return vo_fail(VO_ERR_UNSUPPORTED, "vo_corr_lookup_2d_tiled: levels must be 16-byte aligned");
        }
        h /= 2; w /= 2;
    }
    const int HW = H * W, win = 2 * radius + 1;
    const int64_t out_bs = out_batch_stride ? out_batch_stride : (int64_t)levels * win * win * HW;
    VO_REQUIRE(out_bs >= (int64_t)levels * win * win * HW, "vo_corr_lookup_2d_tiled: out_batch_stride smaller than one item");
    const int64_t total = (int64_t)batch * levels * HW;
    const int64_t blocks = vo_ceil_div(total, 256);
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_corr_lookup_2d_tiled: grid too large");
    const dim3 grid((unsigned)blocks), block(256);
    if (radius == 4) lookup2d_tiled_kernel<4><<<grid, block, 0, vo_stream(stream)>>>(lv, coords, out, out_bs, total, HW, levels);
    else lookup2d_tiled_kernel<3><<<grid, block, 0, vo_stream(stream)>>>(lv, coords, out, out_bs, total, HW, levels);
    VO_CHECK_LAUNCH("vo_corr_lookup_2d_tiled");
    return VO_OK;
}
