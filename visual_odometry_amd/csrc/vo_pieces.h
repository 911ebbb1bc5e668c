// The 16-bit piece formats of the split kernels: every float32 operand is cut into a high and a low 16-bit part
// and a product is the sum of three matrix-core terms, x*w ~ xl*wh + xh*wl + xh*wh, accumulated in float32.
#pragma once
#include "vo_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 p = __builtin_convertvector(v, bf16x2);          // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, p);
}

// The two 16-bit number formats a float32 operand can be cut into.  A policy supplies the matrix instruction, the
// cut of a pair of values (two neighbouring channels of one pixel -> one dword of high parts, one of low parts)
// and what the three terms of a product look like.
//
// PiecesBf16: x = xh + xl, both bf16 (8 significant bits each, float32's exponent range):
//     x*w ~ xl*wh + xh*wl + xh*wh, 16 significant bits per operand.
// PiecesF16:  both float16 (11 significant bits each): 22 significant bits per operand, i.e. within a factor 4 of
//     float32's own rounding, which disappears under the float32 accumulation's (measured: the error of a layer is
//     that of the float32 matrix instruction).  float16's narrow exponent range is handled by exact powers of two:
//       weights   w' = w * 2^e, e chosen on the host so that max|w'| lies in [2^13, 2^14); wh = f16(w'),
//                 wl = f16(w' - wh) (normal for every weight down to 2^-17 of the largest); the accumulators are
//                 multiplied by 2^-e once, before the epilogue;
//       inputs    xh = f16(x), xl = f16((x - xh) * 2^11): the low part is stored 2^11 times too large, so it stays a
//                 normal float16 wherever xh is one, and the term that uses it takes wh * 2^-11 (a packed
//                 multiplication of the fragment in registers, exact because wh >= 2^-3 wherever it matters).
//     Inputs keep 22 bits for 6.1e-5 <= |x| <= 65504 and an absolute error below 2^-36 under that; |x| > 65504
//     becomes infinity in xh and NaN in the result (never a silently wrong number).
struct PiecesBf16 {
    static constexpr bool F16 = false;
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ u32x4 for_low_input(u32x4 a_high) { return a_high; }
    static __device__ __forceinline__ void cut(float a, float b, unsigned& hi, unsigned& lo) {
        hi = pack_bf16(a, b);
        lo = pack_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
    }
    // one weight -> its two 16-bit parts (wscale is ignored: bf16 has float32's exponent range)
    static __device__ __forceinline__ void cut_weight(float w, float, unsigned& hi, unsigned& lo) {
        hi = pack_bf16(w, 0.0f) & 0xffffu;
        lo = pack_bf16(w - __uint_as_float(hi << 16), 0.0f) & 0xffffu;
    }
};
struct PiecesF16 {
    static constexpr bool F16 = true;
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    // the high weights for the term whose input fragment is the (2^11 times too large) low part
    static __device__ __forceinline__ u32x4 for_low_input(u32x4 a_high) {
        const f16x8 s = __builtin_bit_cast(f16x8, a_high) * (_Float16)0x1p-11f;
        return __builtin_bit_cast(u32x4, s);
    }
    // two values -> (dword of high parts, dword of low parts)
    static __device__ __forceinline__ void cut(float a, float b, unsigned& hi, unsigned& lo) {
        const f32x2 v = {a, b};
        const f16x2 h = __builtin_convertvector(v, f16x2);                      // round to nearest even; overflow -> infinity
        const f32x2 r = (v - __builtin_convertvector(h, f32x2)) * 2048.0f;
        const f16x2 l = __builtin_convertvector(r, f16x2);
        hi = __builtin_bit_cast(unsigned, h);
        lo = __builtin_bit_cast(unsigned, l);
    }
    // one weight, multiplied by the power of two wscale first -> its two float16 parts (the low one unscaled)
    static __device__ __forceinline__ void cut_weight(float w, float wscale, unsigned& hi, unsigned& lo) {
        const float ws = w * wscale;
        const _Float16 h = (_Float16)ws;
        const _Float16 l = (_Float16)(ws - (float)h);
        hi = __builtin_bit_cast(unsigned short, h);
        lo = __builtin_bit_cast(unsigned short, l);
    }
};


}  // namespace
