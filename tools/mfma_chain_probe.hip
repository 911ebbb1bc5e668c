// Issue rate of v_mfma_f32_32x32x16_f16 as a function of how many INDEPENDENT accumulator tiles a wave cycles through,
// with three consecutive instructions per tile (the three terms of a split product) as the convolution template issues
// them: does a wave with few accumulator tiles (the stride-2 layers: 2, plus 2 of the fused shortcut) wait for the
// previous instruction's result?  Operands in registers, one or two waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_chain_probe.hip -o tools/bin/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, int RUN>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.0f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f / (1 + i)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
#pragma unroll
            for (int r = 0; r < RUN; ++r) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.0f;
    for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS, int RUN>
void run(int wgs_per_cu, float* out) {
    const int iters = 4096 / (CHAINS * RUN) * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<CHAINS, RUN><<<256 * wgs_per_cu, 256>>>(out, iters);
    hipEventRecord(e0);
    probe<CHAINS, RUN><<<256 * wgs_per_cu, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * CHAINS * RUN;                 // instructions per wave
    const double flops = n * 32768.0 * 1024 * wgs_per_cu;          // 1024 SIMDs x waves per SIMD
    printf("%d tiles x %d in a row, %d wave(s) per SIMD: %.3f ms, %.0f TFLOP/s, %.1f ns per instruction and wave\n", CHAINS, RUN, wgs_per_cu, ms,
           flops / ms / 1e9, ms * 1e6 / n);
}

int main() {
    float* out; hipMalloc(&out, 512 * 256 * 4);
    for (int w = 1; w <= 2; ++w) {
        run<1, 1>(w, out); run<1, 3>(w, out); run<2, 3>(w, out); run<4, 3>(w, out); run<8, 3>(w, out); run<2, 1>(w, out); run<4, 1>(w, out); run<8, 1>(w, out);
    }
    return 0;
}
