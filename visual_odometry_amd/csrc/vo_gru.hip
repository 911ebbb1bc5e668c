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
// Tensors are NCHW float32; "strided" operands are channel slices of a larger tensor
// (plane-contiguous, with their own batch stride in elements).  16 bytes per lane.
#include "vo_common.h"

namespace {

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

struct Slice { const float* p; int64_t bs; };       // read-only channel slice
struct OutSlice { float* p; int64_t bs; };

template <bool VEC>
__global__ __launch_bounds__(256) void gates_kernel(const float* __restrict__ zr, const float* __restrict__ bz,
                                                    const float* __restrict__ br, Slice cz, Slice cr, Slice h,
                                                    float* __restrict__ z_out, OutSlice rh, int C, int64_t hw) {
    const int plane = blockIdx.y, b = plane / C, c = plane - b * C;
    const float* pz = zr + ((int64_t)b * 2 * C + c) * hw;
    const float* pr = pz + (int64_t)C * hw;
    const float* ph = h.p + b * h.bs + (int64_t)c * hw;
    const float* pcz = cz.p ? cz.p + b * cz.bs + (int64_t)c * hw : nullptr;
    const float* pcr = cr.p ? cr.p + b * cr.bs + (int64_t)c * hw : nullptr;
    float* oz = z_out + (int64_t)plane * hw;
    float* orh = rh.p + b * rh.bs + (int64_t)c * hw;
    const float biz = bz ? bz[c] : 0.0f, bir = br ? br[c] : 0.0f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (VEC) {
        for (int64_t i = t; i < (hw >> 2); i += stride) {
            float4 a = reinterpret_cast<const float4*>(pz)[i], d = reinterpret_cast<const float4*>(pr)[i];
            const float4 hh = reinterpret_cast<const float4*>(ph)[i];
            if (pcz) { const float4 v = reinterpret_cast<const float4*>(pcz)[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
            if (pcr) { const float4 v = reinterpret_cast<const float4*>(pcr)[i]; d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w; }
            float4 z, q;
            z.x = sigmoidf(a.x + biz); z.y = sigmoidf(a.y + biz); z.z = sigmoidf(a.z + biz); z.w = sigmoidf(a.w + biz);
            q.x = sigmoidf(d.x + bir) * hh.x; q.y = sigmoidf(d.y + bir) * hh.y;
            q.z = sigmoidf(d.z + bir) * hh.z; q.w = sigmoidf(d.w + bir) * hh.w;
            reinterpret_cast<float4*>(oz)[i] = z;
            reinterpret_cast<float4*>(orh)[i] = q;
        }
    } else {
        for (int64_t i = t; i < hw; i += stride) {
            const float a = pz[i] + (pcz ? pcz[i] : 0.0f) + biz, d = pr[i] + (pcr ? pcr[i] : 0.0f) + bir;
            oz[i] = sigmoidf(a);
            orh[i] = sigmoidf(d) * ph[i];
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void update_kernel(const float* __restrict__ qc, const float* __restrict__ bq, Slice cq,
                                                     const float* __restrict__ z, Slice h, OutSlice out, OutSlice copy,
                                                     int C, int64_t hw) {
    const int plane = blockIdx.y, b = plane / C, c = plane - b * C;
    const float* pq = qc + (int64_t)plane * hw;
    const float* pz = z + (int64_t)plane * hw;
    const float* ph = h.p + b * h.bs + (int64_t)c * hw;
    const float* pcq = cq.p ? cq.p + b * cq.bs + (int64_t)c * hw : nullptr;
    float* po = out.p + b * out.bs + (int64_t)c * hw;
    float* pc = copy.p ? copy.p + b * copy.bs + (int64_t)c * hw : nullptr;
    const float biq = bq ? bq[c] : 0.0f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (VEC) {
        for (int64_t i = t; i < (hw >> 2); i += stride) {
            float4 a = reinterpret_cast<const float4*>(pq)[i];
            const float4 zz = reinterpret_cast<const float4*>(pz)[i], hh = reinterpret_cast<const float4*>(ph)[i];
            if (pcq) { const float4 v = reinterpret_cast<const float4*>(pcq)[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
            float4 o;
            o.x = (1.0f - zz.x) * hh.x + zz.x * tanhf(a.x + biq);
            o.y = (1.0f - zz.y) * hh.y + zz.y * tanhf(a.y + biq);
            o.z = (1.0f - zz.z) * hh.z + zz.z * tanhf(a.z + biq);
            o.w = (1.0f - zz.w) * hh.w + zz.w * tanhf(a.w + biq);
            reinterpret_cast<float4*>(po)[i] = o;
            if (pc) reinterpret_cast<float4*>(pc)[i] = o;
        }
    } else {
        for (int64_t i = t; i < hw; i += stride) {
            const float q = tanhf(pq[i] + (pcq ? pcq[i] : 0.0f) + biq);
            const float o = (1.0f - pz[i]) * ph[i] + pz[i] * q;
            po[i] = o;
            if (pc) pc[i] = o;
        }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool stride_ok(int64_t bs) { return (bs & 3) == 0; }

int check_dims(const char* fn, int batch, int C, int64_t hw) {
    if (batch <= 0 || C <= 0 || hw <= 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: bad sizes batch=%d C=%d hw=%lld", fn, batch, C, (long long)hw);
    if ((int64_t)batch * C > 65535) return vo_fail(VO_ERR_UNSUPPORTED, "%s: batch*C=%lld exceeds 65535", fn, (long long)batch * C);
    return VO_OK;
}

}  // namespace

extern "C" int vo_gru_gates(const float* zr_conv, const float* bias_z, const float* bias_r, const float* cz,
                            int64_t cz_batch_stride, const float* cr, int64_t cr_batch_stride, const float* h,
                            int64_t h_batch_stride, float* z_out, float* rh_out, int64_t rh_batch_stride, int batch,
                            int C, int64_t hw, void* stream) {
    VO_REQUIRE(zr_conv && h && z_out && rh_out, "vo_gru_gates: null pointer");
    int rc = check_dims("vo_gru_gates", batch, C, hw);
    if (rc) return rc;
    const bool vec = (hw % 4 == 0) && aligned16(zr_conv) && aligned16(h) && aligned16(z_out) && aligned16(rh_out) &&
                     (!cz || aligned16(cz)) && (!cr || aligned16(cr)) && stride_ok(cz_batch_stride) &&
                     stride_ok(cr_batch_stride) && stride_ok(h_batch_stride) && stride_ok(rh_batch_stride);
    int64_t bx = vo_ceil_div(vo_ceil_div(hw, 4), 256);
    if (bx > 32) bx = 32;
    const dim3 grid((unsigned)bx, (unsigned)(batch * C));
    const Slice scz{cz, cz_batch_stride}, scr{cr, cr_batch_stride}, sh{h, h_batch_stride};
    const OutSlice srh{rh_out, rh_batch_stride};
    if (vec)
        gates_kernel<true><<<grid, dim3(256), 0, vo_stream(stream)>>>(zr_conv, bias_z, bias_r, scz, scr, sh, z_out, srh, C, hw);
    else
        gates_kernel<false><<<grid, dim3(256), 0, vo_stream(stream)>>>(zr_conv, bias_z, bias_r, scz, scr, sh, z_out, srh, C, hw);
    VO_CHECK_LAUNCH("vo_gru_gates");
    return VO_OK;
}

extern "C" int vo_gru_update(const float* q_conv, const float* bias_q, const float* cq, int64_t cq_batch_stride,
                             const float* z, const float* h, int64_t h_batch_stride, float* h_out,
                             int64_t out_batch_stride, float* h_copy, int64_t copy_batch_stride, int batch, int C,
                             int64_t hw, void* stream) {
    VO_REQUIRE(q_conv && z && h && h_out, "vo_gru_update: null pointer");
    int rc = check_dims("vo_gru_update", batch, C, hw);
    if (rc) return rc;
    const bool vec = (hw % 4 == 0) && aligned16(q_conv) && aligned16(z) && aligned16(h) && aligned16(h_out) &&
                     (!cq || aligned16(cq)) && (!h_copy || aligned16(h_copy)) && stride_ok(cq_batch_stride) &&
                     stride_ok(h_batch_stride) && stride_ok(out_batch_stride) && stride_ok(copy_batch_stride);
    int64_t bx = vo_ceil_div(vo_ceil_div(hw, 4), 256);
    if (bx > 32) bx = 32;
    const dim3 grid((unsigned)bx, (unsigned)(batch * C));
    const Slice scq{cq, cq_batch_stride}, sh{h, h_batch_stride};
    const OutSlice so{h_out, out_batch_stride}, sc{h_copy, copy_batch_stride};
    if (vec)
        update_kernel<true><<<grid, dim3(256), 0, vo_stream(stream)>>>(q_conv, bias_q, scq, z, sh, so, sc, C, hw);
    else
        update_kernel<false><<<grid, dim3(256), 0, vo_stream(stream)>>>(q_conv, bias_q, scq, z, sh, so, sc, C, hw);
    VO_CHECK_LAUNCH("vo_gru_update");
    return VO_OK;
}
