// Fused gate arithmetic of the convolutional GRUs (RAFT's SepConvGRU, RAFT-Stereo's
// ConvGRU).  The convolutions stay in MIOpen and run WITHOUT bias; these two passes
// apply bias, the precomputed context terms, the non-linearities and the blends:
//
//   gates : z = sigmoid(conv_z + b_z + c_z);  r = sigmoid(conv_r + b_r + c_r)
//           z -> z_out;  r*h -> rh_out (written straight into the first channels of the
//           next convolution's input buffer, so no concatenation copy is needed)
//   update: q = tanh(conv_q + b_q + c_q);  h' = (1-z)*h + z*q  -> h_out (+ optional copy
//           into the next GRU half's input buffer)
//
// Stock PyTorch runs 9-12 pointwise kernels and 2-3 torch.cat copies per GRU step.
// Tensors are NCHW, stored as float32 or float16 (biases always float32, arithmetic
// float32); "strided" operands are channel slices of a larger tensor (plane-contiguous,
// with their own batch stride in elements).  Four elements per lane and access.
#include "vo_common.h"

namespace {

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

template <typename T> struct Slice { const T* p; int64_t bs; };       // read-only channel slice
template <typename T> struct OutSlice { T* p; int64_t bs; };

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void gates_kernel(const T* __restrict__ zr, const float* __restrict__ bz,
                                                    const float* __restrict__ br, Slice<T> cz, Slice<T> cr, Slice<T> h,
                                                    T* __restrict__ z_out, OutSlice<T> rh, int C, int64_t hw) {
    const int plane = blockIdx.y, b = plane / C, c = plane - b * C;
    const T* pz = zr + ((int64_t)b * 2 * C + c) * hw;
    const T* pr = pz + (int64_t)C * hw;
    const T* ph = h.p + b * h.bs + (int64_t)c * hw;
    const T* pcz = cz.p ? cz.p + b * cz.bs + (int64_t)c * hw : nullptr;
    const T* pcr = cr.p ? cr.p + b * cr.bs + (int64_t)c * hw : nullptr;
    T* oz = z_out + (int64_t)plane * hw;
    T* orh = rh.p + b * rh.bs + (int64_t)c * hw;
    const float biz = bz ? bz[c] : 0.0f, bir = br ? br[c] : 0.0f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (VEC) {
        for (int64_t i = t; i < (hw >> 2); i += stride) {
            float4 a = vo_ld4<T>(pz, i), d = vo_ld4<T>(pr, i);
            const float4 hh = vo_ld4<T>(ph, i);
            if (pcz) { const float4 v = vo_ld4<T>(pcz, i); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
            if (pcr) { const float4 v = vo_ld4<T>(pcr, i); d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w; }
            float4 z, q;
            z.x = sigmoidf(a.x + biz); z.y = sigmoidf(a.y + biz); z.z = sigmoidf(a.z + biz); z.w = sigmoidf(a.w + biz);
            q.x = sigmoidf(d.x + bir) * hh.x; q.y = sigmoidf(d.y + bir) * hh.y;
            q.z = sigmoidf(d.z + bir) * hh.z; q.w = sigmoidf(d.w + bir) * hh.w;
            vo_st4<T>(oz, i, z);
            vo_st4<T>(orh, i, q);
        }
    } else {
        for (int64_t i = t; i < hw; i += stride) {
            const float a = vo_ld1(pz, i) + (pcz ? vo_ld1(pcz, i) : 0.0f) + biz;
            const float d = vo_ld1(pr, i) + (pcr ? vo_ld1(pcr, i) : 0.0f) + bir;
            vo_st1(oz, i, sigmoidf(a));
            vo_st1(orh, i, sigmoidf(d) * vo_ld1(ph, i));
        }
    }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void update_kernel(const T* __restrict__ qc, const float* __restrict__ bq, Slice<T> cq,
                                                     const T* __restrict__ z, Slice<T> h, OutSlice<T> out,
                                                     OutSlice<T> copy, int C, int64_t hw) {
    const int plane = blockIdx.y, b = plane / C, c = plane - b * C;
    const T* pq = qc + (int64_t)plane * hw;
    const T* pz = z + (int64_t)plane * hw;
    const T* ph = h.p + b * h.bs + (int64_t)c * hw;
    const T* pcq = cq.p ? cq.p + b * cq.bs + (int64_t)c * hw : nullptr;
    T* po = out.p + b * out.bs + (int64_t)c * hw;
    T* pc = copy.p ? copy.p + b * copy.bs + (int64_t)c * hw : nullptr;
    const float biq = bq ? bq[c] : 0.0f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (VEC) {
        for (int64_t i = t; i < (hw >> 2); i += stride) {
            float4 a = vo_ld4<T>(pq, i);
            const float4 zz = vo_ld4<T>(pz, i), hh = vo_ld4<T>(ph, i);
            if (pcq) { const float4 v = vo_ld4<T>(pcq, i); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
            float4 o;
            o.x = (1.0f - zz.x) * hh.x + zz.x * tanhf(a.x + biq);
            o.y = (1.0f - zz.y) * hh.y + zz.y * tanhf(a.y + biq);
            o.z = (1.0f - zz.z) * hh.z + zz.z * tanhf(a.z + biq);
            o.w = (1.0f - zz.w) * hh.w + zz.w * tanhf(a.w + biq);
            vo_st4<T>(po, i, o);
            if (pc) vo_st4<T>(pc, i, o);
        }
    } else {
        for (int64_t i = t; i < hw; i += stride) {
            const float q = tanhf(vo_ld1(pq, i) + (pcq ? vo_ld1(pcq, i) : 0.0f) + biq);
            const float zz = vo_ld1(pz, i);
            const float o = (1.0f - zz) * vo_ld1(ph, i) + zz * q;
            vo_st1(po, i, o);
            if (pc) vo_st1(pc, i, o);
        }
    }
}

bool stride_ok(int64_t bs) { return (bs & 3) == 0; }

int check_dims(const char* fn, int batch, int C, int64_t hw, int dtype) {
    if (batch <= 0 || C <= 0 || hw <= 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: bad sizes batch=%d C=%d hw=%lld", fn, batch, C, (long long)hw);
    if (dtype != VO_F32 && dtype != VO_F16) return vo_fail(VO_ERR_INVALID_ARG, "%s: dtype=%d is neither VO_F32 nor VO_F16", fn, dtype);
    if ((int64_t)batch * C > 65535) return vo_fail(VO_ERR_UNSUPPORTED, "%s: batch*C=%lld exceeds 65535", fn, (long long)batch * C);
    return VO_OK;
}

template <typename T>
void launch_gates(bool vec, dim3 grid, hipStream_t st, const void* zr, const float* bz, const float* br, const void* cz,
                  int64_t czs, const void* cr, int64_t crs, const void* h, int64_t hs, void* z_out, void* rh, int64_t rhs,
                  int C, int64_t hw) {
    const Slice<T> scz{(const T*)cz, czs}, scr{(const T*)cr, crs}, sh{(const T*)h, hs};
    const OutSlice<T> srh{(T*)rh, rhs};
    if (vec)
        gates_kernel<T, true><<<grid, dim3(256), 0, st>>>((const T*)zr, bz, br, scz, scr, sh, (T*)z_out, srh, C, hw);
    else
        gates_kernel<T, false><<<grid, dim3(256), 0, st>>>((const T*)zr, bz, br, scz, scr, sh, (T*)z_out, srh, C, hw);
}

template <typename T>
void launch_update(bool vec, dim3 grid, hipStream_t st, const void* q, const float* bq, const void* cq, int64_t cqs,
                   const void* z, const void* h, int64_t hs, void* out, int64_t os, void* copy, int64_t cs, int C,
                   int64_t hw) {
    const Slice<T> scq{(const T*)cq, cqs}, sh{(const T*)h, hs};
    const OutSlice<T> so{(T*)out, os}, sc{(T*)copy, cs};
    if (vec)
        update_kernel<T, true><<<grid, dim3(256), 0, st>>>((const T*)q, bq, scq, (const T*)z, sh, so, sc, C, hw);
    else
        update_kernel<T, false><<<grid, dim3(256), 0, st>>>((const T*)q, bq, scq, (const T*)z, sh, so, sc, C, hw);
}

}  // namespace

extern "C" int vo_gru_gates(const void* zr_conv, const float* bias_z, const float* bias_r, const void* cz,
                            int64_t cz_batch_stride, const void* cr, int64_t cr_batch_stride, const void* h,
                            int64_t h_batch_stride, void* z_out, void* rh_out, int64_t rh_batch_stride, int batch,
                            int C, int64_t hw, int dtype, void* stream) {
    VO_REQUIRE(zr_conv && h && z_out && rh_out, "vo_gru_gates: null pointer");
    int rc = check_dims("vo_gru_gates", batch, C, hw, dtype);
    if (rc) return rc;
    const size_t es = dtype == VO_F16 ? 2 : 4;
    const bool vec = (hw % 4 == 0) && vo_vec4_ok(zr_conv, es) && vo_vec4_ok(h, es) && vo_vec4_ok(z_out, es) &&
                     vo_vec4_ok(rh_out, es) && (!cz || vo_vec4_ok(cz, es)) && (!cr || vo_vec4_ok(cr, es)) &&
                     stride_ok(cz_batch_stride) && stride_ok(cr_batch_stride) && stride_ok(h_batch_stride) &&
                     stride_ok(rh_batch_stride);
    int64_t bx = vo_ceil_div(vo_ceil_div(hw, 4), 256);
    if (bx > 32) bx = 32;
    const dim3 grid((unsigned)bx, (unsigned)(batch * C));
    if (dtype == VO_F16)
        launch_gates<__half>(vec, grid, vo_stream(stream), zr_conv, bias_z, bias_r, cz, cz_batch_stride, cr, cr_batch_stride,
                             h, h_batch_stride, z_out, rh_out, rh_batch_stride, C, hw);
    else
        launch_gates<float>(vec, grid, vo_stream(stream), zr_conv, bias_z, bias_r, cz, cz_batch_stride, cr, cr_batch_stride,
                            h, h_batch_stride, z_out, rh_out, rh_batch_stride, C, hw);
    VO_CHECK_LAUNCH("vo_gru_gates");
    return VO_OK;
}

extern "C" int vo_gru_update(const void* q_conv, const float* bias_q, const void* cq, int64_t cq_batch_stride,
                             const void* z, const void* h, int64_t h_batch_stride, void* h_out,
                             int64_t out_batch_stride, void* h_copy, int64_t copy_batch_stride, int batch, int C,
                             int64_t hw, int dtype, void* stream) {
    VO_REQUIRE(q_conv && z && h && h_out, "vo_gru_update: null pointer");
    int rc = check_dims("vo_gru_update", batch, C, hw, dtype);
    if (rc) return rc;
    const size_t es = dtype == VO_F16 ? 2 : 4;
    const bool vec = (hw % 4 == 0) && vo_vec4_ok(q_conv, es) && vo_vec4_ok(z, es) && vo_vec4_ok(h, es) &&
                     vo_vec4_ok(h_out, es) && (!cq || vo_vec4_ok(cq, es)) && (!h_copy || vo_vec4_ok(h_copy, es)) &&
                     stride_ok(cq_batch_stride) && stride_ok(h_batch_stride) && stride_ok(out_batch_stride) &&
                     stride_ok(copy_batch_stride);
    int64_t bx = vo_ceil_div(vo_ceil_div(hw, 4), 256);
    if (bx > 32) bx = 32;
    const dim3 grid((unsigned)bx, (unsigned)(batch * C));
    if (dtype == VO_F16)
        launch_update<__half>(vec, grid, vo_stream(stream), q_conv, bias_q, cq, cq_batch_stride, z, h, h_batch_stride, h_out,
                              out_batch_stride, h_copy, copy_batch_stride, C, hw);
    else
        launch_update<float>(vec, grid, vo_stream(stream), q_conv, bias_q, cq, cq_batch_stride, z, h, h_batch_stride, h_out,
                             out_batch_stride, h_copy, copy_batch_stride, C, hw);
    VO_CHECK_LAUNCH("vo_gru_update");
    return VO_OK;
}
