// Does the ORDER of matrix instructions matter at the power limit?  v_mfma_f32_32x32x16_f16 on random operands from
// registers, 8 accumulator tiles in rotation, 4 A and 4 B fragments per wave; the same 16 (A, B) products per pass issued
//   0: so that neither operand repeats from one instruction to the next,
//   1: B held for four consecutive instructions (A changes),
//   2: A held for four consecutive instructions (B changes),
//   3: one fixed (A, B) pair for every instruction (the floor of operand switching).
// Time = energy here: the chip runs at its power limit, so fewer switched operand bits show up as a higher clock.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_operand_order.hip -o tools/bin/mfma_operand_order
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ u32x4 operand(unsigned seed) {
    u32x4 r;
    for (int i = 0; i < 4; ++i) {
        const unsigned h = mix(seed * 4 + i);
        unsigned lo = h & 0xffffu, hi = h >> 16;
        lo = (lo & 0x83ffu) | (((13 + ((h >> 3) & 3)) & 31) << 10); hi = (hi & 0x83ffu) | (((13 + ((h >> 19) & 3)) & 31) << 10);
        r[i] = lo | (hi << 16);
    }
    return r;
}

template <int ORDER>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
    f32x16 acc[8];
    for (int c = 0; c < 8; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.0f;
    u32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = operand(threadIdx.x * 8 + i + blockIdx.x * 4096); b[i] = operand(threadIdx.x * 8 + 4 + i + blockIdx.x * 4096); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            int ia, ib;
            if (ORDER == 0) { ia = s & 3; ib = (s + (s >> 2)) & 3; }          // both change every instruction
            else if (ORDER == 1) { ib = s >> 2; ia = s & 3; }                  // B held for four
            else if (ORDER == 2) { ia = s >> 2; ib = s & 3; }                  // A held for four
            else { ia = 0; ib = 0; }
            const int t = s & 7;                                               // eight accumulator tiles in rotation (the sums are not the point)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ia]), __builtin_bit_cast(f16x8, b[ib]), acc[t], 0, 0, 0);
        }
    }
    float sum = 0.0f;
    for (int c = 0; c < 8; ++c) for (int e = 0; e < 16; ++e) sum += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int ORDER>
void run(float* out, const char* what) {
    const int iters = 1 << 19;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    probe<ORDER><<<256, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.1f ms, %.0f TFLOP/s\n", what, ms, (double)iters * 16 * 32768.0 * 1024 / ms / 1e9);
    fflush(stdout);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(out, "A and B change every instruction");
        run<1>(out, "B held for four instructions");
        run<2>(out, "A held for four instructions");
        run<3>(out, "one (A, B) pair throughout");
    }
    return 0;
}
