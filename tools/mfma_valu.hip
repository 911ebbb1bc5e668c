// Cost of other instruction classes in an MFMA-saturated SIMD.  Each loop trip issues 4 independent
// v_mfma_f32_32x32x2f32 (256 matrix-pipe cycles) plus N instructions of one class; 4 waves per SIMD.
// Cases 10 and up: the same next to v_mfma_f32_32x32x16_bf16 (4 per trip = 128 matrix-pipe cycles).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
enum { NONE, VALU, PKADD, DSW, DSW2, DSR64, BARRIER, VMOV, SALU };
template <int KIND, int N, bool BF = false> __global__ __launch_bounds__(256) void spin(float* out, int iters) {
    __shared__ float sm[8192];
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    float v[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[4];
    for (int j = 0; j < 8; ++j) v[j] = x + j;
    for (int j = 0; j < 4; ++j) p[j] = f2{x + j, y};
    sm[threadIdx.x] = x; sm[threadIdx.x + 256] = y;
    __syncthreads();
    const unsigned la = (threadIdx.x & 255) * 4, la8 = (threadIdx.x & 255) * 8;
    int sacc = 0;
    bf16x8 xb, yb;
    for (int j = 0; j < 8; ++j) { xb[j] = (__bf16)(x + j); yb[j] = (__bf16)(y - j); }
    for (int i = 0; i < iters; ++i) {
        if (BF) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a3, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (KIND == VALU) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(x));
            if (KIND == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j & 3]) : "v"(p[(j + 1) & 3]));
            if (KIND == VMOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j & 7]) : "v"(x));
            if (KIND == DSW) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(la), "v"(x), "n"((j & 7) * 1024));
            if (KIND == DSW2) asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(la), "v"(x), "v"(y), "n"((j & 7) * 4), "n"((j & 7) * 4 + 32));
            if (KIND == DSR64) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(p[j & 3]) : "v"(la8), "n"((j & 7) * 2048));
            if (KIND == BARRIER) __builtin_amdgcn_s_barrier();
            if (KIND == SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
        }
        if (KIND == DSW || KIND == DSW2 || KIND == DSR64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = sacc;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
    for (int j = 0; j < 8; ++j) s += v[j];
    for (int j = 0; j < 4; ++j) s += p[j].x + p[j].y;
    if (s == 123.456f) out[threadIdx.x] = s;
}
static float* g_out;
template <int KIND, int N, bool BF = false> void run(const char* name) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 4;
    spin<KIND, N, BF><<<blocks, 256>>>(g_out, 2000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    spin<KIND, N, BF><<<blocks, 256>>>(g_out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 4 * 4096.0 * (BF ? 8 : 1);
    const double cyc = ms * 1e-3 * 2.4e9 / (4.0 * iters);   // SIMD cycles per wave-trip at 2.4 GHz
    const double base = BF ? 128 : 256;
    printf("%-28s %7.2f ms %6.1f TFLOP/s  %6.0f cycles/trip (%.0f = MFMA only)  %+.1f cycles per extra instr\n", name, ms,
           flop / ms / 1e9, cyc, base, N ? (cyc - base) / N : 0.0);
    fflush(stdout);
}
int main(int argc, char** argv) {
    (void)hipMalloc(&g_out, 4096);
    const int which = argc > 1 ? atoi(argv[1]) : 0;
    run<NONE, 0>("warm");
    if (which >= 10) {
        run<NONE, 0, true>("bf16 mfma only");
        switch (which) {
            case 10: run<VALU, 16, true>("bf16 + 16 v_add_f32"); run<VALU, 48, true>("bf16 + 48 v_add_f32"); break;
            case 11: run<PKADD, 16, true>("bf16 + 16 v_pk_add_f32"); break;
            case 14: run<DSW, 16, true>("bf16 + 16 ds_write_b32"); break;
            case 16: run<DSR64, 16, true>("bf16 + 16 ds_read_b64"); break;
        }
        return 0;
    }
    run<NONE, 0>("mfma only");
    switch (which) {
        case 0: run<VALU, 16>("16 v_add_f32"); run<VALU, 48>("48 v_add_f32"); break;
        case 1: run<PKADD, 16>("16 v_pk_add_f32"); run<PKADD, 48>("48 v_pk_add_f32"); break;
        case 2: run<VMOV, 32>("32 v_mov_b32"); break;
        case 3: run<SALU, 32>("32 s_add_u32"); break;
        case 4: run<DSW, 16>("16 ds_write_b32"); break;
        case 5: run<DSW2, 16>("16 ds_write2st64_b32"); break;
        case 6: run<DSR64, 16>("16 ds_read_b64"); break;
        case 7: run<BARRIER, 2>("2 s_barrier"); break;
    }
    return 0;
}
