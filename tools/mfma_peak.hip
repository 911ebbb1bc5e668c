// Sustained float32 MFMA rate of the whole chip: every wave issues independent
// v_mfma_f32_32x32x2f32 back to back.  Prints TFLOP/s for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
    if (s == 123.456f) out[threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;                    // 256 threads = one wave per SIMD
        spin<<<blocks, 256>>>(out, 100);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            spin<<<blocks, 256>>>(out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * iters * 4 * 4096.0;
            printf("waves/SIMD %d: %.2f ms  %.1f TFLOP/s\n", wps, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
