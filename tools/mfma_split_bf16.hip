// Evidence for a split-bf16 multiplication path: (1) the sustained chip-wide rate of
// v_mfma_f32_32x32x16_bf16 next to v_mfma_f32_32x32x2_f32, (2) the error of a 32x32xK product
// computed with float32 MFMA, with three bf16 MFMAs per k-block (hi*hi + hi*lo + lo*hi, float32
// accumulation) and with plain bf16, each against float64 on the host.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_split_bf16.hip -o tools/bin/mfma_split_bf16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void spin_bf16(float* out, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    bf16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(threadIdx.x * 1e-3f + j); y[j] = (__bf16)(blockIdx.x * 1e-3f - j); }
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
    if (s == 123.456f) out[threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void spin_f32(float* out, int iters) {
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

// One wave: C[32][32] = A[32][K] * B[K][32] three ways.  A row-major, B row-major, K a multiple of 16.
// out: [3][32][32] = float32 MFMA, split bf16, plain bf16.
__global__ __launch_bounds__(64) void product(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 c32 = {0}, cs = {0}, cb = {0};
    for (int k0 = 0; k0 < K; k0 += 2)
        c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k0 + h], B[(k0 + h) * 32 + r], c32, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 ah, al, bh, bl;
        for (int j = 0; j < 8; ++j) {
            const float a = A[r * K + k0 + 8 * h + j], b = B[(k0 + 8 * h + j) * 32 + r];
            ah[j] = (__bf16)a; al[j] = (__bf16)(a - (float)ah[j]);
            bh[j] = (__bf16)b; bl[j] = (__bf16)(b - (float)bh[j]);
        }
        cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, cb, 0, 0, 0);
        // small terms first, so that they are not rounded away against the large one
        cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, cs, 0, 0, 0);
        cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, cs, 0, 0, 0);
        cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, cs, 0, 0, 0);
    }
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        out[0 * 1024 + row * 32 + r] = c32[e];
        out[1 * 1024 + row * 32 + r] = cs[e];
        out[2 * 1024 + row * 32 + r] = cb[e];
    }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    float* out; CHECK(hipMalloc(&out, 3 * 1024 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int kind = 0; kind < 2; ++kind)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int blocks = 256 * wps;                    // 256 threads = one wave per SIMD
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                if (kind == 0) spin_f32<<<blocks, 256>>>(out, iters); else spin_bf16<<<blocks, 256>>>(out, iters);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            const double flop = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * (kind == 0 ? 2 : 16);
            printf("%s, waves/SIMD %d: %.2f ms  %.1f TFLOP/s%s\n", kind == 0 ? "f32 32x32x2 " : "bf16 32x32x16", wps, best,
                   flop / best / 1e9, kind == 1 ? "  (a three-term split product: a third of that)" : "");
        }
    // accuracy at the reduction lengths of the networks' layers (channels x taps)
    for (int K : {576, 2304, 3456}) {
        std::vector<float> A(32 * K), B(K * 32), C(3 * 1024);
        srand(K);
        auto rnd = [] { float s = 0; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.0f; };
        for (auto& v : A) v = rnd();
        for (auto& v : B) v = rnd() / sqrtf((float)K);
        float *dA, *dB;
        CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dB, B.size() * 4));
        CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        product<<<1, 64>>>(dA, dB, out, K);
        CHECK(hipMemcpy(C.data(), out, C.size() * 4, hipMemcpyDeviceToHost));
        double err[3] = {0, 0, 0}, big = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += (double)A[i * K + k] * (double)B[k * 32 + j];
                big = fmax(big, fabs(ref));
                for (int v = 0; v < 3; ++v) err[v] = fmax(err[v], fabs(C[v * 1024 + i * 32 + j] - ref));
            }
        printf("K = %4d: max error / max |C|: f32 MFMA %.1e   split bf16 (3 terms) %.1e   plain bf16 %.1e\n", K, err[0] / big,
               err[1] / big, err[2] / big);
        CHECK(hipFree(dA)); CHECK(hipFree(dB));
    }
    return 0;
}
