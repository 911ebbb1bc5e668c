// Fused normalisation / activation passes for the conv encoders (inference only).
//
// The encoders of both networks run norm -> ReLU after every convolution and
// ReLU(x + y) after every residual block, on tensors of up to 1.3 GB (RAFT-Stereo's
// stem works at full resolution).  As separate framework kernels that is 5-7 HBM
// passes per convolution output (statistics, transform, clamp, add, clamp); fused
// here it is 3 for instance norm (one read for the statistics, one read+write to
// apply) and 2 for everything else; the residual join relu(skip + y) rides along in
// the same pass for one more read.  All kernels work IN PLACE on the convolution
// output, planes are (n, c) slices of an NCHW tensor stored as float32 or float16,
// four elements per lane and access, arithmetic in float32.
#include "vo_common.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double* scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, VO_WAVE);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += scratch[w];
    return t;
}

__device__ __forceinline__ float4 relu4(float4 v) {
    return make_float4(fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f), fmaxf(v.z, 0.0f), fmaxf(v.w, 0.0f));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <typename T> __device__ __forceinline__ bool aligned4(const T* a, const T* b) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & (4 * sizeof(T) - 1)) == 0;
}

// InstanceNorm2d (no affine, biased variance) + optional ReLU; one workgroup per plane.
// With ``skip`` the residual join is folded in: x = relu(skip + act(norm(x))).
template <typename T>
__global__ __launch_bounds__(256) void instnorm_kernel(T* __restrict__ x, const T* __restrict__ skip, int64_t hw,
                                                       float eps, int relu) {
    __shared__ double scratch[4];
    T* p = x + (int64_t)blockIdx.x * hw;
    const T* q = skip ? skip + (int64_t)blockIdx.x * hw : nullptr;
    const int64_t n4 = aligned4<T>(p, q) ? (hw >> 2) : 0;
    float s = 0.0f, ss = 0.0f;
    double ds = 0.0, dss = 0.0;
    int since = 0;
    for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
        const float4 v = vo_ld4<T>(p, i);
        s += (v.x + v.y) + (v.z + v.w);
        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        if (++since == 64) { ds += s; dss += ss; s = 0.0f; ss = 0.0f; since = 0; }   // bound the float32 run length
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < hw; i += blockDim.x) { const float v = vo_ld1(p, i); s += v; ss += v * v; }
    ds += s; dss += ss;
    const double tot = block_sum(ds, scratch);
    const double tot2 = block_sum(dss, scratch);
    const double mean = tot / (double)hw;
    double var = tot2 / (double)hw - mean * mean;
    if (var < 0.0) var = 0.0;
    const float m = (float)mean, r = (float)(1.0 / sqrt(var + (double)eps));
    for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
        float4 v = vo_ld4<T>(p, i);
        v.x = (v.x - m) * r; v.y = (v.y - m) * r; v.z = (v.z - m) * r; v.w = (v.w - m) * r;
        if (relu) v = relu4(v);
        if (q) v = relu4(add4(v, vo_ld4<T>(q, i)));
        vo_st4<T>(p, i, v);
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < hw; i += blockDim.x) {
        float v = (vo_ld1(p, i) - m) * r;
        if (relu) v = fmaxf(v, 0.0f);
        if (q) v = fmaxf(v + vo_ld1(q, i), 0.0f);
        vo_st1(p, i, v);
    }
}

// Instance norm with known statistics: grid.y = plane, grid.x strides the plane.  With ``skip_stats`` the
// residual branch too is a raw convolution output with its plane sums: it is normalised (and, with
// ``skip_relu``, rectified) on the fly, so the branch never exists in normalised form.
// STREAM: the tensor is far larger than the caches -- the raw values and the skip branch are read once, the result is
// not read again before it has been evicted -- so loads and stores are marked nontemporal (float32 path).
template <typename T, bool STREAM = false>
__global__ __launch_bounds__(256) void instnorm_apply_kernel(T* __restrict__ x, const T* __restrict__ skip,
                                                             const double* __restrict__ stats,
                                                             const double* __restrict__ skip_stats, int64_t hw, float eps,
                                                             float skip_eps, int relu, int skip_relu, int64_t plane0) {
    const int64_t plane = plane0 + blockIdx.y;
    const double mean = stats[2 * plane] / (double)hw;
    double var = stats[2 * plane + 1] / (double)hw - mean * mean;
    if (var < 0.0) var = 0.0;
    const float m = (float)mean, r = (float)(1.0 / sqrt(var + (double)eps));
    float sm = 0.0f, sr = 1.0f;
    if (skip_stats) {
        const double smean = skip_stats[2 * plane] / (double)hw;
        double svar = skip_stats[2 * plane + 1] / (double)hw - smean * smean;
        if (svar < 0.0) svar = 0.0;
        sm = (float)smean; sr = (float)(1.0 / sqrt(svar + (double)skip_eps));
    }
    const float slo = skip_relu ? 0.0f : -INFINITY;
    T* p = x + plane * hw;
    const T* q = skip ? skip + plane * hw : nullptr;
    const int64_t n4 = aligned4<T>(p, q) ? (hw >> 2) : 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    for (int64_t i = t; i < n4; i += stride) {
        float4 v;
        if constexpr (STREAM) {
            const f32x4_ t4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4_*>(p) + i);
            v = make_float4(t4[0], t4[1], t4[2], t4[3]);
        } else {
            v = vo_ld4<T>(p, i);
        }
        v.x = (v.x - m) * r; v.y = (v.y - m) * r; v.z = (v.z - m) * r; v.w = (v.w - m) * r;
        if (relu) v = relu4(v);
        if (q) {
            float4 s;
            if constexpr (STREAM) {
                const f32x4_ t4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4_*>(q) + i);
                s = make_float4(t4[0], t4[1], t4[2], t4[3]);
            } else {
                s = vo_ld4<T>(q, i);
            }
            if (skip_stats) {
                s.x = fmaxf((s.x - sm) * sr, slo); s.y = fmaxf((s.y - sm) * sr, slo);
                s.z = fmaxf((s.z - sm) * sr, slo); s.w = fmaxf((s.w - sm) * sr, slo);
            }
            v = relu4(add4(v, s));
        }
        if constexpr (STREAM) {
            const f32x4_ o4 = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(o4, reinterpret_cast<f32x4_*>(p) + i);
        } else {
            vo_st4<T>(p, i, v);
        }
    }
    for (int64_t i = (n4 << 2) + t; i < hw; i += stride) {
        float v = (vo_ld1(p, i) - m) * r;
        if (relu) v = fmaxf(v, 0.0f);
        if (q) {
            float s = vo_ld1(q, i);
            if (skip_stats) s = fmaxf((s - sm) * sr, slo);
            v = fmaxf(v + s, 0.0f);
        }
        vo_st1(p, i, v);
    }
}

// y = x * scale[c] + shift[c] (+ ReLU) (+ residual join): BatchNorm2d in inference with
// the running statistics folded into scale / shift, or a convolution's bias + ReLU.
// grid.y = plane, grid.x strides the plane.
template <typename T>
__global__ __launch_bounds__(256) void affine_kernel(T* __restrict__ x, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const T* __restrict__ skip,
                                                     int C, int64_t hw, int relu, int64_t plane0) {
    const int64_t plane = plane0 + blockIdx.y;
    const float a = scale[plane % C], b = shift[plane % C];
    T* p = x + plane * hw;
    const T* q = skip ? skip + plane * hw : nullptr;
    const int64_t n4 = aligned4<T>(p, q) ? (hw >> 2) : 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t; i < n4; i += stride) {
        float4 v = vo_ld4<T>(p, i);
        v.x = v.x * a + b; v.y = v.y * a + b; v.z = v.z * a + b; v.w = v.w * a + b;
        if (relu) v = relu4(v);
        if (q) v = relu4(add4(v, vo_ld4<T>(q, i)));
        vo_st4<T>(p, i, v);
    }
    for (int64_t i = (n4 << 2) + t; i < hw; i += stride) {
        float v = vo_ld1(p, i) * a + b;
        if (relu) v = fmaxf(v, 0.0f);
        if (q) v = fmaxf(v + vo_ld1(q, i), 0.0f);
        vo_st1(p, i, v);
    }
}

// a = relu(a + b)
template <typename T>
__global__ __launch_bounds__(256) void add_relu_kernel(T* __restrict__ a, const T* __restrict__ b, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n4 = aligned4<T>(a, b) ? (n >> 2) : 0;
    for (int64_t i = t; i < n4; i += stride) vo_st4<T>(a, i, relu4(add4(vo_ld4<T>(a, i), vo_ld4<T>(b, i))));
    for (int64_t i = (n4 << 2) + t; i < n; i += stride) vo_st1(a, i, fmaxf(vo_ld1(a, i) + vo_ld1(b, i), 0.0f));
}

// Bilinear resize with corner-aligned sampling (F.interpolate(mode="bilinear",
// align_corners=True)): the inter-level exchange of RAFT-Stereo's multi-level GRU.
// One lane per 4 output pixels of a row; source rows are re-read from L2.
template <typename T>
__global__ __launch_bounds__(256) void resize_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t total,
                                                     int channels, int64_t out_batch_stride, int h, int w, int H, int W,
                                                     float sy, float sx, bool vec) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int wq = (W + 3) >> 2;
    const int xq = (int)(t % wq);
    const int y = (int)((t / wq) % H);
    const int64_t plane = t / ((int64_t)wq * H);
    const int64_t b = plane / channels, c = plane - b * channels;
    const float fy = sy * (float)y;
    const int y0 = min((int)fy, h - 1), y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
    const T* r0 = in + (plane * h + y0) * (int64_t)w;
    const T* r1 = in + (plane * h + y1) * (int64_t)w;
    T* o = out + b * out_batch_stride + (c * H + y) * (int64_t)W + 4 * xq;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = min(4 * xq + i, W - 1);
        const float fx = sx * (float)x;
        const int x0 = min((int)fx, w - 1), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
        v[i] = ly0 * (lx0 * vo_ld1(r0, x0) + lx1 * vo_ld1(r0, x1)) + ly1 * (lx0 * vo_ld1(r1, x0) + lx1 * vo_ld1(r1, x1));
    }
    if (vec) {                                                   // W a multiple of 4, rows 16-byte (8-byte for halves) aligned
        vo_st4<T>(o, 0, make_float4(v[0], v[1], v[2], v[3]));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * xq + i < W) vo_st1(o, i, v[i]);
    }
}

// The same resize for the shapes the networks have (an enlargement by 1.5 or more whose output rows are whole groups
// of four pixels): a lane's four outputs read source columns inside one window of four, so it fetches the window of
// both source rows with two 16-byte (8-byte for halves) loads instead of sixteen scalar ones (the scalar form is bound
// by the address path: 1.8 TB/s) and picks its taps out of registers.  Same arithmetic as resize_kernel.
// grid: x over the (row, group of four) pairs of one plane, y over the planes.
template <typename T> struct alignas(sizeof(T)) Window4 { T v[4]; };

template <typename T>
__global__ __launch_bounds__(256) void resize_window_kernel(const T* __restrict__ in, T* __restrict__ out, int planes,
                                                            int channels, int64_t out_batch_stride, int h, int w, int H,
                                                            int W, float sy, float sx) {
    const int wq = W >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * wq) return;
    const int y = idx / wq, xq = idx - y * wq;
    const float fy = sy * (float)y;
    const int y0 = min((int)fy, h - 1), y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
    const int base = min(min((int)(sx * (float)(4 * xq)), w - 1), w - 4);       // first column of the window
    int k0[4], k1[4];
    float lx0[4], lx1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float fx = sx * (float)(4 * xq + i);
        const int x0 = min((int)fx, w - 1), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        lx1[i] = fx - (float)x0;
        lx0[i] = 1.0f - lx1[i];
        k0[i] = x0 - base;
        k1[i] = x1 - base;
    }
    // (a chain of selects on registers: no indexed access, no branches)
    auto pick = [](float q0, float q1, float q2, float q3, int k) {
        float r = q0;
        r = k == 1 ? q1 : r;
        r = k == 2 ? q2 : r;
        r = k == 3 ? q3 : r;
        return r;
    };
    for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
        const int b = plane / channels, c = plane - b * channels;
        const Window4<T> w0 = *reinterpret_cast<const Window4<T>*>(in + ((int64_t)plane * h + y0) * w + base);
        const Window4<T> w1 = *reinterpret_cast<const Window4<T>*>(in + ((int64_t)plane * h + y1) * w + base);
        const float a0 = vo_ld1(w0.v, 0), a1 = vo_ld1(w0.v, 1), a2 = vo_ld1(w0.v, 2), a3 = vo_ld1(w0.v, 3);
        const float d0 = vo_ld1(w1.v, 0), d1 = vo_ld1(w1.v, 1), d2 = vo_ld1(w1.v, 2), d3 = vo_ld1(w1.v, 3);
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[i] = ly0 * (lx0[i] * pick(a0, a1, a2, a3, k0[i]) + lx1[i] * pick(a0, a1, a2, a3, k1[i])) +
                   ly1 * (lx0[i] * pick(d0, d1, d2, d3, k0[i]) + lx1[i] * pick(d0, d1, d2, d3, k1[i]));
        vo_st4<T>(out + b * out_batch_stride + ((int64_t)c * H + y) * W + 4 * xq, 0, make_float4(v[0], v[1], v[2], v[3]));
    }
}

// 3x3 average pooling, stride 2, padding 1, the padding counted in the divisor (torch's
// avg_pool2d(x, 3, stride=2, padding=1)): RAFT-Stereo's pool2x between GRU levels.
// One lane per 4 output pixels of a row: 9 input columns of 3 rows.
template <typename T>
__global__ __launch_bounds__(256) void pool_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t total, int h, int w,
                                                   int H, int W, bool vec, int64_t channels, int64_t out_batch_stride) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int wq = (W + 3) >> 2;
    const int xq = (int)(t % wq);
    const int y = (int)((t / wq) % H);
    const int64_t plane = t / ((int64_t)wq * H);
    const T* p = in + plane * (int64_t)h * w;
    float col[9];                                                // column sums over the three rows, columns 8xq-1 .. 8xq+7
#pragma unroll
    for (int j = 0; j < 9; ++j) col[j] = 0.0f;
    const int xb = 8 * xq;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = 2 * y + dy;
        if (yy < 0 || yy >= h) continue;
        const T* r = p + (int64_t)yy * w;
        if (xb >= 1) col[0] += vo_ld1(r, xb - 1);
        if (vec && xb + 8 <= w) {
            const float4 a = vo_ld4<T>(r + xb, 0), b = vo_ld4<T>(r + xb + 4, 0);
            col[1] += a.x; col[2] += a.y; col[3] += a.z; col[4] += a.w;
            col[5] += b.x; col[6] += b.y; col[7] += b.z; col[8] += b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (xb + j < w) col[1 + j] += vo_ld1(r, xb + j);
        }
    }
    // the output may be a channel slice of a wider tensor: items out_batch_stride elements apart
    const int64_t item = plane / channels;
    T* o = out + item * out_batch_stride + ((plane - item * channels) * H + y) * (int64_t)W + 4 * xq;
    const float k = 1.0f / 9.0f;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (col[2 * i] + col[2 * i + 1] + col[2 * i + 2]) * k;
    if (vec && (W & 3) == 0) {
        vo_st4<T>(o, 0, make_float4(v[0], v[1], v[2], v[3]));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * xq + i < W) vo_st1(o, i, v[i]);
    }
}

int check_dtype(const char* fn, int dtype) {
    if (dtype != VO_F32 && dtype != VO_F16) return vo_fail(VO_ERR_INVALID_ARG, "%s: dtype=%d is neither VO_F32 nor VO_F16", fn, dtype);
    return VO_OK;
}

}  // namespace

extern "C" int vo_resize_bilinear(const void* in, void* out, int64_t out_batch_stride, int batch, int channels, int h,
                                  int w, int H, int W, int dtype, void* stream) {
    VO_REQUIRE(in && out, "vo_resize_bilinear: null pointer");
    VO_REQUIRE(batch > 0 && channels > 0 && h > 0 && w > 0 && H > 0 && W > 0, "vo_resize_bilinear: bad sizes");
    if (int rc = check_dtype("vo_resize_bilinear", dtype)) return rc;
    if (out_batch_stride == 0) out_batch_stride = (int64_t)channels * H * W;
    VO_REQUIRE(out_batch_stride >= (int64_t)channels * H * W, "vo_resize_bilinear: out_batch_stride smaller than one item");
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    const int64_t total = (int64_t)batch * channels * H * ((W + 3) / 4);
    const int64_t blocks = vo_ceil_div(total, 256);
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_resize_bilinear: grid too large");
    const dim3 grid((unsigned)blocks);
    const bool vec = (W & 3) == 0 && (out_batch_stride & 3) == 0 && vo_vec4_ok(out, dtype == VO_F16 ? 2 : 4);
    if (vec && w >= 4 && 3.0f * sx <= 2.0f && (int64_t)H * (W >> 2) < 0x7fffff00LL && (int64_t)batch * channels < 0x7fffffffLL) {
        // four consecutive outputs span at most three source columns (floor(a + 3 sx) - floor(a) <= 2) plus the right tap
        const int planes = batch * channels;
        // eight planes per lane: the tap positions and weights of a lane are the same in every plane
        const int64_t ny = vo_ceil_div((int64_t)planes, 8);
        const dim3 wgrid((unsigned)vo_ceil_div((int64_t)H * (W >> 2), 256), (unsigned)(ny < 65535 ? ny : 65535));
        if (dtype == VO_F16)
            resize_window_kernel<__half><<<wgrid, dim3(256), 0, vo_stream(stream)>>>((const __half*)in, (__half*)out, planes, channels,
                                                                                    out_batch_stride, h, w, H, W, sy, sx);
        else
            resize_window_kernel<float><<<wgrid, dim3(256), 0, vo_stream(stream)>>>((const float*)in, (float*)out, planes, channels,
                                                                                   out_batch_stride, h, w, H, W, sy, sx);
        VO_CHECK_LAUNCH("vo_resize_bilinear");
        return VO_OK;
    }
    if (dtype == VO_F16)
        resize_kernel<__half><<<grid, dim3(256), 0, vo_stream(stream)>>>((const __half*)in, (__half*)out, total, channels,
                                                                        out_batch_stride, h, w, H, W, sy, sx, vec);
    else
        resize_kernel<float><<<grid, dim3(256), 0, vo_stream(stream)>>>((const float*)in, (float*)out, total, channels,
                                                                       out_batch_stride, h, w, H, W, sy, sx, vec);
    VO_CHECK_LAUNCH("vo_resize_bilinear");
    return VO_OK;
}

namespace {
int launch_pool(const char* name, const void* in, void* out, int64_t out_batch_stride, int64_t batch, int64_t channels, int h, int w,
                int dtype, void* stream) {
    if (!in || !out) return vo_fail(VO_ERR_INVALID_ARG, "%s: null pointer", name);
    if (batch <= 0 || channels <= 0 || h <= 0 || w <= 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: bad sizes", name);
    if (int rc = check_dtype(name, dtype)) return rc;
    const int H = (h - 1) / 2 + 1, W = (w - 1) / 2 + 1;
    if (out_batch_stride == 0) out_batch_stride = channels * H * W;
    if (out_batch_stride < channels * H * W) return vo_fail(VO_ERR_INVALID_ARG, "%s: out_batch_stride smaller than one item", name);
    const int64_t total = batch * channels * H * ((W + 3) / 4);
    const int64_t blocks = vo_ceil_div(total, 256);
    if (blocks > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "%s: grid too large", name);
    const size_t es = dtype == VO_F16 ? 2 : 4;
    // rows of both tensors 4-element aligned
    const bool vec = (w & 3) == 0 && (out_batch_stride & 3) == 0 && vo_vec4_ok(in, es) && vo_vec4_ok(out, es);
    const dim3 grid((unsigned)blocks);
    if (dtype == VO_F16)
        pool_kernel<__half><<<grid, dim3(256), 0, vo_stream(stream)>>>((const __half*)in, (__half*)out, total, h, w, H, W, vec, channels,
                                                                      out_batch_stride);
    else
        pool_kernel<float><<<grid, dim3(256), 0, vo_stream(stream)>>>((const float*)in, (float*)out, total, h, w, H, W, vec, channels,
                                                                     out_batch_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e));
    return VO_OK;
}
}  // namespace

extern "C" int vo_avg_pool3x3s2(const void* in, void* out, int64_t planes, int h, int w, int dtype, void* stream) {
    return launch_pool("vo_avg_pool3x3s2", in, out, 0, 1, planes, h, w, dtype, stream);
}

extern "C" int vo_avg_pool3x3s2_into(const void* in, void* out, int64_t out_batch_stride, int batch, int channels, int h, int w,
                                     int dtype, void* stream) {
    return launch_pool("vo_avg_pool3x3s2_into", in, out, out_batch_stride, batch, channels, h, w, dtype, stream);
}

extern "C" int vo_instance_norm_relu(void* x, const void* skip, int64_t planes, int64_t hw, float eps, int relu,
                                     int dtype, void* stream) {
    VO_REQUIRE(x, "vo_instance_norm_relu: null pointer");
    VO_REQUIRE(planes > 0 && hw > 0, "vo_instance_norm_relu: bad sizes planes=%lld hw=%lld", (long long)planes, (long long)hw);
    if (int rc = check_dtype("vo_instance_norm_relu", dtype)) return rc;
    if (planes > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_instance_norm_relu: too many planes");
    const dim3 grid((unsigned)planes);
    if (dtype == VO_F16)
        instnorm_kernel<__half><<<grid, dim3(256), 0, vo_stream(stream)>>>((__half*)x, (const __half*)skip, hw, eps, relu);
    else
        instnorm_kernel<float><<<grid, dim3(256), 0, vo_stream(stream)>>>((float*)x, (const float*)skip, hw, eps, relu);
    VO_CHECK_LAUNCH("vo_instance_norm_relu");
    return VO_OK;
}

extern "C" int vo_instance_norm_apply(void* x, const void* skip, const double* stats, const double* skip_stats, int64_t planes,
                                      int64_t hw, float eps, float skip_eps, int relu, int skip_relu, int dtype, void* stream) {
    VO_REQUIRE(x && stats, "vo_instance_norm_apply: null pointer");
    VO_REQUIRE(planes > 0 && hw > 0, "vo_instance_norm_apply: bad sizes");
    VO_REQUIRE(!(skip_stats && !skip), "vo_instance_norm_apply: skip statistics without a skip tensor");
    VO_REQUIRE(skip != x, "vo_instance_norm_apply: skip must not alias the tensor normalised in place");
    if (int rc = check_dtype("vo_instance_norm_apply", dtype)) return rc;
    int64_t bx = vo_ceil_div(vo_ceil_div(hw, 4), 256);
    if (bx > 64) bx = 64;
    // planes ride on grid.y, which holds at most 65535: larger tensors go out in slices
    for (int64_t p0 = 0; p0 < planes; p0 += 65535) {
        const dim3 grid((unsigned)bx, (unsigned)(planes - p0 < 65535 ? planes - p0 : 65535));
        if (dtype == VO_F16)
            instnorm_apply_kernel<__half><<<grid, dim3(256), 0, vo_stream(stream)>>>((__half*)x, (const __half*)skip, stats, skip_stats, hw,
                                                                                  eps, skip_eps, relu, skip_relu, p0);
        else if (planes * hw * 4 >= (512LL << 20))
            instnorm_apply_kernel<float, true><<<grid, dim3(256), 0, vo_stream(stream)>>>((float*)x, (const float*)skip, stats, skip_stats, hw,
                                                                                       eps, skip_eps, relu, skip_relu, p0);
        else
            instnorm_apply_kernel<float><<<grid, dim3(256), 0, vo_stream(stream)>>>((float*)x, (const float*)skip, stats, skip_stats, hw,
                                                                                 eps, skip_eps, relu, skip_relu, p0);
        VO_CHECK_LAUNCH("vo_instance_norm_apply");
    }
    return VO_OK;
}

extern "C" int vo_channel_affine_relu(void* x, const float* scale, const float* shift, const void* skip,
                                      int64_t batch, int C, int64_t hw, int relu, int dtype, void* stream) {
    VO_REQUIRE(x && scale && shift, "vo_channel_affine_relu: null pointer");
    VO_REQUIRE(batch > 0 && C > 0 && hw > 0, "vo_channel_affine_relu: bad sizes");
    if (int rc = check_dtype("vo_channel_affine_relu", dtype)) return rc;
    const int64_t planes = batch * C;
    int64_t bx = vo_ceil_div(vo_ceil_div(hw, 4), 256);
    if (bx > 64) bx = 64;
    // planes ride on grid.y, which holds at most 65535: larger tensors go out in slices
    for (int64_t p0 = 0; p0 < planes; p0 += 65535) {
        const dim3 grid((unsigned)bx, (unsigned)(planes - p0 < 65535 ? planes - p0 : 65535));
        if (dtype == VO_F16)
            affine_kernel<__half><<<grid, dim3(256), 0, vo_stream(stream)>>>((__half*)x, scale, shift, (const __half*)skip, C, hw, relu, p0);
        else
            affine_kernel<float><<<grid, dim3(256), 0, vo_stream(stream)>>>((float*)x, scale, shift, (const float*)skip, C, hw, relu, p0);
        VO_CHECK_LAUNCH("vo_channel_affine_relu");
    }
    return VO_OK;
}

extern "C" int vo_add_relu(void* a, const void* b, int64_t n, int dtype, void* stream) {
    VO_REQUIRE(a && b, "vo_add_relu: null pointer");
    VO_REQUIRE(n > 0, "vo_add_relu: n=%lld", (long long)n);
    if (int rc = check_dtype("vo_add_relu", dtype)) return rc;
    int64_t blocks = vo_ceil_div(vo_ceil_div(n, 4), 256);
    if (blocks > 4096) blocks = 4096;
    const dim3 grid((unsigned)blocks);
    if (dtype == VO_F16)
        add_relu_kernel<__half><<<grid, dim3(256), 0, vo_stream(stream)>>>((__half*)a, (const __half*)b, n);
    else
        add_relu_kernel<float><<<grid, dim3(256), 0, vo_stream(stream)>>>((float*)a, (const float*)b, n);
    VO_CHECK_LAUNCH("vo_add_relu");
    return VO_OK;
}

namespace {

// out[n,k,y,x] = bias[k] + sum_{r,s} P[n, (k*3+r)*3+s, y+r-1, x+s-1]  (zero outside the image):
// the second half of a 3x3 convolution with very few output channels, whose first half is
// a 1x1 convolution with 9K output channels (one plane per filter tap).  One lane per pixel.
__global__ __launch_bounds__(256) void shift_add_kernel(const float* __restrict__ P, const float* __restrict__ bias,
                                                        float* __restrict__ out, int K, int H, int W, int z0) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int n = (z0 + blockIdx.z) / K, k = (z0 + blockIdx.z) % K;
    if (x >= W) return;
    const float* p = P + ((int64_t)n * K + k) * 9 * H * W;
    float acc = bias ? bias[k] : 0.0f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = y + r - 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int xx = x + s - 1;
            if (xx >= 0 && xx < W) acc += p[((int64_t)(r * 3 + s) * H + yy) * W + xx];
        }
    }
    out[(((int64_t)n * K + k) * H + y) * W + x] = acc;
}

}  // namespace

extern "C" int vo_shift_add_3x3(const float* planes, const float* bias, float* out, int N, int K, int H, int W,
                                void* stream) {
    VO_REQUIRE(planes && out, "vo_shift_add_3x3: null pointer");
    VO_REQUIRE(N > 0 && K > 0 && H > 0 && W > 0, "vo_shift_add_3x3: bad sizes");
    if (H > 65535 || (int64_t)N * K > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "vo_shift_add_3x3: grid too large");
    // (image, channel) rides on grid.z, which holds at most 65535: larger batches go out in slices
    for (int64_t z0 = 0; z0 < (int64_t)N * K; z0 += 65535) {
        const int64_t nz = (int64_t)N * K - z0 < 65535 ? (int64_t)N * K - z0 : 65535;
        const dim3 grid((unsigned)vo_ceil_div(W, 256), (unsigned)H, (unsigned)nz);
        shift_add_kernel<<<grid, dim3(256), 0, vo_stream(stream)>>>(planes, bias, out, K, H, W, (int)z0);
        VO_CHECK_LAUNCH("vo_shift_add_3x3");
    }
    return VO_OK;
}
