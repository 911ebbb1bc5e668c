// The multiply phase of the split-bf16 3x3 convolution with the two bf16 matrix-instruction shapes, on random
// operands resident in LDS (no global traffic inside the loop): the same 64 channels x 128 pixels per wave and
// the same float32 products per chunk of 16 input channels,
//   32x32x16: 9 taps x 4 pixel tiles x (2 B reads, 1 A read, 6 instructions)            = 216 instructions of 32 cycles
//   16x16x32: 14 items (9 cross-term pairs, 5 pairs of high*high taps) x (4 A reads, 8 x (1 B read, 4 instructions))
//                                                                                        = 448 instructions of 16 cycles
// Two workgroups of four waves per CU, as the convolution kernel runs.  Prints wall time and the rate of USEFUL
// bf16 multiply-adds (27 per float32 product tap-channel, i.e. the 216-instruction count for both).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_probe.hip -o tools/bin/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PATCH = 4 * 18 * 36 * 16;      // [part][k-half][row][column][8 channels]
constexpr int WEIGHTS = 9 * 4 * 1024;        // [tap][part][m-tile][lane][16 bytes]
constexpr int LDS = PATCH + WEIGHTS;

__device__ void fill_lds(unsigned char* lds, const unsigned* src) {
    for (int i = threadIdx.x; i < LDS / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = src[i];
    __syncthreads();
}

__global__ __launch_bounds__(256, 2) void shape32(const unsigned* src, float* out, int chunks) {
    __shared__ __align__(16) unsigned char lds[LDS];
    fill_lds(lds, src);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* a_lds = lds + PATCH + lane * 16;
    const unsigned char* b_lds = lds + (lane >> 5) * (18 * 36 * 16) + ((4 * wave) * 36 + (lane & 31)) * 16;
    f32x16 acc[2][4];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 4; ++n) for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;
    for (int c = 0; c < chunks; ++c) {
        bf16x8 fa[2][2][2], fb[2][2];
        for (int part = 0; part < 2; ++part) {
            for (int mt = 0; mt < 2; ++mt) fa[0][part][mt] = *reinterpret_cast<const bf16x8*>(a_lds + (part * 2 + mt) * 1024);
            fb[0][part] = *reinterpret_cast<const bf16x8*>(b_lds + part * 2 * 18 * 36 * 16);
        }
#pragma unroll
        for (int st = 0; st < 36; ++st) {
            const int tap = st / 4, nt = st % 4, ca = tap & 1, cb = st & 1;
            if (st + 1 < 36) {
                const int t2 = (st + 1) / 4, n2 = (st + 1) % 4;
                for (int part = 0; part < 2; ++part)
                    fb[cb ^ 1][part] = *reinterpret_cast<const bf16x8*>(b_lds + part * 2 * 18 * 36 * 16 + ((n2 + t2 / 3) * 36 + t2 % 3) * 16);
            }
            if (tap + 1 < 9) { const int f = nt; fa[ca ^ 1][f / 2][f % 2] = *reinterpret_cast<const bf16x8*>(a_lds + (((tap + 1) * 2 + f / 2) * 2 + f % 2) * 1024); }
            for (int mt = 0; mt < 2; ++mt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ca][1][mt], fb[cb][0], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ca][0][mt], fb[cb][1], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ca][0][mt], fb[cb][0], acc[mt][nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 4; ++n) for (int e = 0; e < 16; ++e) s += acc[m][n][e];
    if (s == 123.456f) out[threadIdx.x] = s;
}

// 16x16x32: lane (row or column = lane & 15, k-group = lane >> 4): k-groups 0,1 and 2,3 take their eight channels
// from two different (tap, part) images, so one instruction adds two of the three terms, or the high*high term of two taps
__global__ __launch_bounds__(256, 2) void shape16(const unsigned* src, float* out, int chunks) {
    __shared__ __align__(16) unsigned char lds[LDS];
    fill_lds(lds, src);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, kg = lane >> 4, pair = kg >> 1, kh = kg & 1;
    constexpr int PL = 18 * 36 * 16;
    f32x4 acc[4][8];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 8; ++n) for (int e = 0; e < 4; ++e) acc[m][n][e] = 0.0f;
    // weights image as the 32x32 kernel keeps it: ((tap*2 + part)*2 + mt32) KiB + (k-half*32 + channel)*16.
    // Every address is a per-lane base + a per-item lane term (which of the item's two images) + a constant.
    const unsigned char* a_base = lds + PATCH + (kh * 32 + l16) * 16;
    const unsigned char* b_base = lds + kh * PL + (4 * wave * 36 + l16) * 16;
    auto a_off = [](int tap, int part, int m16) { return ((tap * 2 + part) * 2 + (m16 >> 1)) * 1024 + (m16 & 1) * 256; };
    auto b_off = [](int tap, int part, int pt) { return part * 2 * PL + (((pt >> 1) + tap / 3) * 36 + (pt & 1) * 16 + tap % 3) * 16; };
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int item = 0; item < 14; ++item) {
            // items 0..8: cross terms of tap `item` (k-groups 0,1: low x * high w; 2,3: high x * low w);
            // items 9..13: high*high of taps 2(item-9) and 2(item-9)+1 (the last one pairs tap 8 with nothing)
            const bool cross = item < 9;
            const int t0 = cross ? item : 2 * (item - 9), t1 = cross ? item : (t0 + 1 < 9 ? t0 + 1 : t0);
            const int da = a_off(t1, cross ? 1 : 0, 0) - a_off(t0, 0, 0);            // second image - first image
            const int db = b_off(t1, 0, 0) - b_off(t0, cross ? 1 : 0, 0);
            const unsigned char* pa = a_base + pair * da + a_off(t0, 0, 0);
            const unsigned char* pb = b_base + pair * db + b_off(t0, cross ? 1 : 0, 0);
            bf16x8 fa[4], fb[2];
            for (int m = 0; m < 4; ++m) {
                fa[m] = *reinterpret_cast<const bf16x8*>(pa + a_off(0, 0, m));
                if (!cross && t1 == t0 && pair) for (int j = 0; j < 8; ++j) fa[m][j] = (__bf16)0.0f;
            }
            fb[0] = *reinterpret_cast<const bf16x8*>(pb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pt = 0; pt < 8; ++pt) {
                if (pt + 1 < 8) fb[(pt + 1) & 1] = *reinterpret_cast<const bf16x8*>(pb + b_off(0, 0, pt + 1));
                for (int m = 0; m < 4; ++m) acc[m][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], fb[pt & 1], acc[m][pt], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 8; ++n) for (int e = 0; e < 4; ++e) s += acc[m][n][e];
    if (s == 123.456f) out[threadIdx.x] = s;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    std::vector<unsigned> h(LDS / 4);
    srand(1);
    for (auto& v : h) {                                   // two random bf16 values in [-1, 1) per word
        auto r = []() { union { float f; unsigned u; } c; c.f = (rand() / (float)RAND_MAX) * 2.0f - 1.0f; return c.u >> 16; };
        v = r() | (r() << 16);
    }
    unsigned* src; float* out;
    CHECK(hipMalloc(&src, LDS)); CHECK(hipMalloc(&out, 1024));
    CHECK(hipMemcpy(src, h.data(), LDS, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int chunks = 400; const int blocks = argc > 1 ? atoi(argv[1]) : 512;
    for (int round = 0; round < 3; ++round)
        for (int kind = 0; kind < 2; ++kind) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0));
                if (kind == 0) shape32<<<blocks, 256>>>(src, out, chunks); else shape16<<<blocks, 256>>>(src, out, chunks);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double useful = 2.0 * 216 * 32 * 32 * 16 * 4.0 * blocks * chunks;      // bf16 flops of the 216-instruction count
            printf("%s: %.3f ms  %.0f TFLOP/s of useful bf16 multiply-adds (%.0f float32-equivalent)\n", kind ? "16x16x32" : "32x32x16", best,
                   useful / best / 1e9, useful / best / 3e9);
        }
    return 0;
}
