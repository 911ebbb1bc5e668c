// 1x1 stride-1 convolution with many channels on both sides, float32 in and out (NCHW), every product from three
// 16-bit matrix-core terms (bf16 or float16 pieces, see vo_pieces.h and vo_conv_split.h for the arithmetic; same
// weights image, same order of terms, the same bits in the result as the template's kernel): the motion encoder's first layer (correlation window -> 256 channels, every refinement
// iteration) and the mask head's last layer.
//
// A 1x1 layer has one tap: 24 matrix instructions per chunk of 16 input channels and wave, against 216 for a 3x3
// layer, so a kernel that stages patch and weights through LDS chunk by chunk spends its time at the two barriers
// of a chunk.  Here the WEIGHTS stay: a workgroup of 8 waves copies the whole image of its 64 output channels
// (C x 256 bytes: 84 KiB for 336 input channels) into LDS once and then walks over pixel tiles with no barrier at
// all; every wave owns 64 pixels of a tile, loads its own input (already the B operand's shape: a lane takes
// eight channels of NTL consecutive pixels), splits it in registers, and keeps the next chunk's loads in flight
// while it multiplies.  The workgroups of the channel blocks of one layer walk the same pixel tiles at the same
// time and sit on the same XCD, so the input is read from memory once.
// The stores are 8 bytes wide ON PURPOSE.  A 16-byte buffer store whose offset rides in the scalar operand
// (raw_buffer_store_b128 with a non-constant soffset) may be followed at once by a vector instruction that
// overwrites its data registers: the compiler inserts the wait states this needs only when the scalar offset is a
// constant, and on gfx950 the store then writes the NEW register contents for the lanes it reads last (measured: a
// variant with four pixels per lane returned sporadically wrong values -- the first data register, lanes 12-15 of
// each row of 16 -- with the scalar offset, none with the same offset folded into the lane offset).  Stores of at
// most 8 bytes are not affected; every 16-byte buffer store of this library passes a constant 0 as scalar offset.
// tests/test_raft_ops_gpu.py keeps this kernel under a bit-for-bit comparison with the template's.
#include "vo_common.h"
#include "vo_pieces.h"

namespace {

constexpr int CK = 16;                        // input channels per chunk (the k of one matrix instruction)
constexpr int KB = 64;                        // output channels per workgroup
constexpr int A_CHUNK = 2 * 2 * 64 * 16;      // bytes of weights per chunk: [part][m-tile][lane][16]
constexpr int NTL = 2;                        // pixel tiles of 32 pixels per wave (a lane's NTL pixels are consecutive)
constexpr int PW = 32 * NTL;                  // pixels per wave
constexpr int WAVES = 8;

struct StreamEp { const float* scale; const float* shift; int relu; int64_t y_batch_stride; float unscale; };

template <typename PC>
__global__ __launch_bounds__(64 * WAVES, 2) void conv1x1_stream_kernel(const float* __restrict__ x, const unsigned short* __restrict__ Ag,
                                                                     float* __restrict__ y, StreamEp ep, int C, int P, int K,
                                                                     int kblocks, int ptiles, int items, int groups) {
    extern __shared__ __align__(16) unsigned char lds[];         // the channel block's weights: nchunks x A_CHUNK
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = C / CK;
    // workgroups are dealt round robin over the 8 XCDs: the channel blocks of one group take consecutive slots of ONE XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int kb = slot % kblocks, group = xcd * (per_xcd / kblocks) + slot / kblocks;
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(Ag + (int64_t)kb * nchunks * (A_CHUNK / 2));
        for (int i = tid; i < nchunks * (A_CHUNK / 16); i += 64 * WAVES) reinterpret_cast<u32x4*>(lds)[i] = src[i];
    }
    __syncthreads();                                             // the only barrier

    const int kk = lane >> 5, nn = lane & 31;
    const unsigned char* a_lds = lds + lane * 16;                // + chunk * A_CHUNK + (part * 2 + mt) * 1024
    for (int item = group; item < items; item += groups) {
        const int n = item / ptiles, pt = item - n * ptiles;
        const int pw0 = (pt * WAVES + wave) * PW;
        if (pw0 >= P) continue;                                  // this wave's pixels lie past the plane (waves are independent)
        const int p0 = pw0 + NTL * nn;                           // this lane's first pixel
        // one image of x: channels past C and bytes past the image read as 0.0f
        const __amdgpu_buffer_rsrc_t xrs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (int64_t)n * C * P), 0, C * P * 4, 0x00020000);
        // the lane's part of the address (the part the range check covers: pixels past the plane read as zeros);
        // the channel's part is uniform and rides in the scalar offset, so chunks are never addressed past the end
        const int x_off = p0 < P ? (8 * kk * P + p0) * 4 : (int)0x80000000;      // + (chunk * 16 + j) * P * 4

        f32x16 acc[2][NTL];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < NTL; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][t][e] = 0.0f;

        u32x2 b0[8], b1[8];
        auto load = [&](u32x2 (&b)[8], int chunk) {
            chunk = chunk < nchunks ? chunk : nchunks - 1;       // past the end: reloaded, never used
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = __builtin_amdgcn_raw_buffer_load_b64(xrs, x_off, (chunk * CK + j) * P * 4, 0);
        };
        auto multiply = [&](const u32x2 (&b)[8], int chunk) {
            const unsigned char* a = a_lds + chunk * A_CHUNK;
            u32x4 fa[2][2], fs[2];
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) fa[part][mt] = *reinterpret_cast<const u32x4*>(a + (part * 2 + mt) * 1024);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) fs[mt] = PC::for_low_input(fa[0][mt]);
#pragma unroll
            for (int t = 0; t < NTL; ++t) {
                u32x4 hi, lo;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    unsigned h, l;
                    PC::cut(__uint_as_float(b[2 * jj][t]), __uint_as_float(b[2 * jj + 1][t]), h, l);
                    hi[jj] = h;
                    lo[jj] = l;
                }
                const u32x4 xh = hi, xl = lo;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    // the small terms first, so that they are not rounded away against the large one
                    acc[mt][t] = PC::mfma(fa[1][mt], xh, acc[mt][t]);
                    acc[mt][t] = PC::mfma(fs[mt], xl, acc[mt][t]);
                    acc[mt][t] = PC::mfma(fa[0][mt], xh, acc[mt][t]);
                }
            }
        };
        load(b0, 0);
        for (int chunk = 0; chunk < nchunks; chunk += 2) {
            load(b1, chunk + 1);
            multiply(b0, chunk);
            load(b0, chunk + 2);
            if (chunk + 1 < nchunks) multiply(b1, chunk + 1);
        }

        if constexpr (PC::F16) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int t = 0; t < NTL; ++t) acc[mt][t] *= ep.unscale;      // back to the weights' own scale (a power of two)
        }

        // C/D layout of a 32x32 tile: pixel = lane & 31 (pixel p0 + t of this lane in tile t), channel row =
        // (e & 3) + 8*(e >> 2) + 4*(lane >> 5).  A lane stores NTL consecutive pixels of a channel at once (the host
        // admits only planes of whole groups of four pixels, 16-byte aligned outputs and channel counts in multiples of 8).
        // The lane's part of the address is the same for every channel row (pixels past the plane carry an offset
        // past the extent, which the range check drops); the row's part is uniform and rides in the scalar offset.
        const __amdgpu_buffer_rsrc_t yrs =
            __builtin_amdgcn_make_buffer_rsrc(y + (int64_t)n * ep.y_batch_stride, 0, K * P * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t scr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ep.scale ? ep.scale : ep.shift), 0, ep.scale ? K * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t shr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ep.shift), 0, ep.shift ? K * 4 : 0, 0x00020000);
        const int y_off = p0 < P ? (4 * kk * P + p0) * 4 : (int)0x80000000;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            // a tile's scale and shift values are loaded before its first store: the wait counter of vector memory
            // operations counts loads and stores in one order, a load between stores would wait for the stores before it
            float scv[16], shv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ku = kb * KB + mt * 32 + (e & 3) + 8 * (e >> 2);           // + 4 * kk
                // both loads always go out (an empty descriptor returns zeros): no load sits behind a branch.  The range
                // check covers the lane's part of the address only, so the scalar part is kept inside the K entries by
                // hand: rows past K (a partial last channel block; K is a multiple of 8, so ku < K means ku + 4 < K too)
                // read row K - 8 + 4*kk instead, and are never stored
                const int kc = ku < K ? ku : K - 8;
                const float sv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(scr, kk * 16, kc * 4, 0));
                scv[e] = ep.scale ? sv : 1.0f;
                shv[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(shr, kk * 16, kc * 4, 0));      // 0 without a shift
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ku = kb * KB + mt * 32 + (e & 3) + 8 * (e >> 2);
                u32x2 o;
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    float v = ep.shift ? fmaf(acc[mt][t][e], scv[e], shv[e]) : acc[mt][t][e];
                    if (ep.relu) v = fmaxf(v, 0.0f);
                    o[t] = __float_as_uint(v);
                }
                // K is a multiple of 8: the rows ku and ku + 4 of a lane pair are both inside or both outside
                if (ku < K) __builtin_amdgcn_raw_buffer_store_b64(o, yrs, y_off, ku * P * 4, 0);
            }
        }
    }
}

}  // namespace

// Whether the layer fits this kernel: the weights of a channel block in LDS, 32-bit byte offsets inside one image
// (two chunks past its end included).
bool vo_conv1x1_stream_fits(int C, int64_t P, int K) {
    return C % CK == 0 && C >= 64 && K >= 64 && K % 8 == 0 && P % 4 == 0 && (int64_t)(C / CK) * A_CHUNK <= 156 * 1024 && ((int64_t)C + 2 * CK + 8) * P * 4 < 0x7fffffffLL &&
           (int64_t)K * P * 4 < 0x7fffffffLL;
}

int vo_conv1x1_stream_launch(const char* name, const float* x, const void* arranged, float* y, int64_t y_batch_stride, const float* scale,
                             const float* shift, int relu, int N, int C, int64_t P, int K, int pieces, int weight_exp, hipStream_t st) {
    if (!x || !arranged || !y) return vo_fail(VO_ERR_INVALID_ARG, "%s: null pointer", name);
    if (N <= 0 || C <= 0 || P <= 0 || K <= 0) return vo_fail(VO_ERR_INVALID_ARG, "%s: bad sizes", name);
    if (scale && !shift) return vo_fail(VO_ERR_INVALID_ARG, "%s: scale without shift", name);
    if (x == y) return vo_fail(VO_ERR_INVALID_ARG, "%s: the output must not alias the input", name);
    if (y_batch_stride == 0) y_batch_stride = (int64_t)K * P;
    if (y_batch_stride < (int64_t)K * P) return vo_fail(VO_ERR_INVALID_ARG, "%s: y_batch_stride smaller than one item", name);
    if ((reinterpret_cast<uintptr_t>(arranged) & 15) != 0) return vo_fail(VO_ERR_UNSUPPORTED, "%s: the weights image must be 16-byte aligned", name);
    if ((reinterpret_cast<uintptr_t>(y) & 15) != 0 || (y_batch_stride & 3) != 0)
        return vo_fail(VO_ERR_UNSUPPORTED, "%s: the output must be 16-byte aligned", name);
    if (!vo_conv1x1_stream_fits(C, P, K)) return vo_fail(VO_ERR_UNSUPPORTED, "%s: the layer does not fit the streaming kernel", name);
    const int kblocks = (K + KB - 1) / KB;
    const int64_t ptiles = vo_ceil_div(P, (int64_t)WAVES * PW);
    const int64_t items = ptiles * N;
    if (items > 0x7fffffffLL) return vo_fail(VO_ERR_UNSUPPORTED, "%s: too many pixel tiles", name);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
        cus = 256;
    // as many workgroups per CU as their weights leave room for in LDS (at most 3); per XCD a whole number of groups of
    // kblocks workgroups, no more groups than items
    const int lds_bytes = (C / CK) * A_CHUNK;
    int wgs_per_cu = (160 * 1024) / lds_bytes;
    if (wgs_per_cu > 3) wgs_per_cu = 3;
    if (wgs_per_cu < 1) wgs_per_cu = 1;
    int per_xcd = cus * wgs_per_cu / 8 / kblocks;
    if (per_xcd < 1) per_xcd = 1;
    const int64_t need = vo_ceil_div(items, 8);
    if (per_xcd > need) per_xcd = (int)need;
    const int groups = 8 * per_xcd;
    const int lds = (C / CK) * A_CHUNK;
    const bool f16 = pieces == VO_PIECES_F16;
    hipError_t e = hipFuncSetAttribute(f16 ? reinterpret_cast<const void*>(conv1x1_stream_kernel<PiecesF16>)
                                           : reinterpret_cast<const void*>(conv1x1_stream_kernel<PiecesBf16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "%s: LDS attribute: %s", name, hipGetErrorString(e));
    const StreamEp ep{scale, shift, relu, y_batch_stride, f16 ? ldexpf(1.0f, -weight_exp) : 1.0f};
    if (f16)
        conv1x1_stream_kernel<PiecesF16><<<dim3((unsigned)(groups * kblocks)), dim3(64 * WAVES), lds, st>>>(
            x, static_cast<const unsigned short*>(arranged), y, ep, C, (int)P, K, kblocks, (int)ptiles, (int)items, groups);
    else
        conv1x1_stream_kernel<PiecesBf16><<<dim3((unsigned)(groups * kblocks)), dim3(64 * WAVES), lds, st>>>(
            x, static_cast<const unsigned short*>(arranged), y, ep, C, (int)P, K, kblocks, (int)ptiles, (int)items, groups);
    e = hipGetLastError();
    if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e));
    return VO_OK;
}
