// Sustained chip-wide rate of v_mfma_f32_32x32x16_f16 against v_mfma_f32_32x32x16_bf16 on random operands held in
// registers, each measured over about a second (long enough for the clock to settle under the power limit), one after
// the other and then again: is the float16 instruction itself slower than the bf16 one at the clock the chip holds?
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f16_vs_bf16.hip -o tools/bin/mfma_f16_vs_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// 16-bit patterns with random sign and mantissa and an exponent field near the format's bias: values of magnitude ~1
template <bool F16> __device__ u32x4 operand(unsigned seed) {
    u32x4 r;
    for (int i = 0; i < 4; ++i) {
        const unsigned h = mix(seed * 4 + i);
        unsigned lo = h & 0xffffu, hi = h >> 16;
        if (F16) { lo = (lo & 0x83ffu) | (((13 + ((h >> 3) & 3)) & 31) << 10); hi = (hi & 0x83ffu) | (((13 + ((h >> 19) & 3)) & 31) << 10); }
        else     { lo = (lo & 0x807fu) | ((125 + ((h >> 3) & 3)) << 7);        hi = (hi & 0x807fu) | ((125 + ((h >> 19) & 3)) << 7); }
        r[i] = lo | (hi << 16);
    }
    return r;
}

template <bool F16, bool ZERO>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
    f32x16 acc[8];
    for (int c = 0; c < 8; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.0f;
    u32x4 a[2], b[4];
    for (int i = 0; i < 2; ++i) a[i] = ZERO ? u32x4{0, 0, 0, 0} : operand<F16>(threadIdx.x * 8 + i + blockIdx.x * 4096);
    for (int i = 0; i < 4; ++i) b[i] = ZERO ? u32x4{0, 0, 0, 0} : operand<F16>(threadIdx.x * 8 + 2 + i + blockIdx.x * 4096);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (F16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[c & 1]), __builtin_bit_cast(f16x8, b[c >> 1]), acc[c], 0, 0, 0);
            else     acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[c & 1]), __builtin_bit_cast(bf16x8, b[c >> 1]), acc[c], 0, 0, 0);
        }
    }
    float s = 0.0f;
    for (int c = 0; c < 8; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool F16, bool ZERO>
void run(float* out, const char* what) {
    const int iters = 1 << 20;                                        // 8.4 M instructions per wave: about a second
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<F16, ZERO><<<256, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)iters * 8 * 32768.0 * 1024;
    printf("%-28s %.1f ms, %.0f TFLOP/s\n", what, ms, flops / ms / 1e9);
    fflush(stdout);
}

int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<false, false>(out, "bf16, random operands");
        run<true, false>(out, "f16, random operands");
        run<false, true>(out, "bf16, zero operands");
        run<true, true>(out, "f16, zero operands");
    }
    return 0;
}
