// Batched 3-point RANSAC + inlier refit (camera/camera_svd_xform.py:52-149): one
// 256-lane workgroup per frame pair, one hypothesis per lane per round.
//
//   1. wave 0 compacts the indices of the weighted points into LDS (ballot prefix);
//   2. every lane draws its hypotheses' three distinct points from a counter-based
//      generator, fits the 3-point rotation (same 3x3 fit as the Kabsch kernel) and
//      counts inliers over all points, which stream through LDS in tiles that the
//      whole workgroup loads together (all lanes read the same point: broadcast);
//   3. block-wide arg-max with the reference's tie rule (the FIRST hypothesis with
//      the strictly largest count wins), the winner's transform is shared via LDS;
//   4. wave 0 refits on the winner's inliers with the two-pass centred covariance.
//
// The reference draws from NumPy's unseeded global stream, so its result is not
// reproducible run to run; here the triples are a pure function of
// (seed, pair index in the whole sequence, hypothesis index), so a trajectory does
// not depend on how the sequence is cut into chunks or shards.
#include "vo_common.h"
#include "vo_svd3.h"

namespace {

constexpr int THREADS = 256;
constexpr int TILE = 256;          // points per LDS tile

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// three distinct indices in [0, n), uniform: i0 in [0,n), i1 in [0,n-1) skipping i0,
// i2 in [0,n-2) skipping both
__device__ __forceinline__ void draw_triple(uint64_t seed, uint32_t pair, uint32_t hyp, uint32_t n, uint32_t idx[3]) {
    uint64_t s = splitmix64(seed ^ (((uint64_t)pair << 32) | hyp));
    uint32_t i0 = (uint32_t)(s % n);
    s = splitmix64(s);
    uint32_t i1 = (uint32_t)(s % (n - 1));
    if (i1 >= i0) ++i1;
    s = splitmix64(s);
    uint32_t i2 = (uint32_t)(s % (n - 2));
    const uint32_t lo = i0 < i1 ? i0 : i1, hi = i0 < i1 ? i1 : i0;
    if (i2 >= lo) ++i2;
    if (i2 >= hi) ++i2;
    idx[0] = i0; idx[1] = i1; idx[2] = i2;
}

struct Rigid { double R[3][3]; double t[3]; };

// 3-point Kabsch (camera_svd_xform.py:103-114)
__device__ __forceinline__ void fit3(const double* p, const double* q, const uint32_t g[3], Rigid& x) {
    double cp[3], cq[3], h[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        cp[k] = (p[3 * g[0] + k] + p[3 * g[1] + k] + p[3 * g[2] + k]) / 3.0;
        cq[k] = (q[3 * g[0] + k] + q[3 * g[1] + k] + q[3 * g[2] + k]) / 3.0;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        double dp[3], dq[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { dp[k] = p[3 * g[s] + k] - cp[k]; dq[k] = q[3 * g[s] + k] - cq[k]; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) h[r][c] += dp[r] * dq[c];
    }
    rotation_from_covariance(h, x.R);
#pragma unroll
    for (int r = 0; r < 3; ++r) x.t[r] = cq[r] - (x.R[r][0] * cp[0] + x.R[r][1] * cp[1] + x.R[r][2] * cp[2]);
}

__device__ __forceinline__ bool is_inlier(const Rigid& x, const double* p, const double* q, double thr) {
#pragma clang fp contract(off)
    const double ex = ((x.R[0][0] * p[0] + x.R[0][1] * p[1]) + x.R[0][2] * p[2]) + x.t[0] - q[0];
    const double ey = ((x.R[1][0] * p[0] + x.R[1][1] * p[1]) + x.R[1][2] * p[2]) + x.t[1] - q[1];
    const double ez = ((x.R[2][0] * p[0] + x.R[2][1] * p[1]) + x.R[2][2] * p[2]) + x.t[2] - q[2];
    return sqrt((ex * ex + ey * ey) + ez * ez) < thr;            // errors < threshold (:118-119)
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, VO_WAVE);
    return x;
}

__global__ __launch_bounds__(THREADS) void ransac_kernel(const double* __restrict__ P, const double* __restrict__ Q,
                                                        const uint8_t* __restrict__ wgt, double thr, int iterations,
                                                        uint64_t seed, uint32_t first_pair, double* __restrict__ T,
                                                        int32_t* __restrict__ n_inl_out,
                                                        uint8_t* __restrict__ mask_out, int N) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tile = reinterpret_cast<double*>(smem);                           // [TILE][6]
    double* bestx = tile + TILE * 6;                                          // 12 doubles
    int* red_cnt = reinterpret_cast<int*>(bestx + 12);                        // [THREADS]
    int* red_hyp = red_cnt + THREADS;                                         // [THREADS]
    int* shared_n = red_hyp + THREADS;                                        // 4 ints
    uint16_t* list = reinterpret_cast<uint16_t*>(shared_n + 4);               // [N]

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const uint32_t pair = first_pair + (uint32_t)b;      // the pair's index in the whole sequence keys its triples
    const double* p = P + (int64_t)b * N * 3;
    const double* q = Q + (int64_t)b * N * 3;
    const uint8_t* w = wgt ? wgt + (int64_t)b * N : nullptr;

    // 1. compact the weighted points' indices
    if (tid < VO_WAVE) {
        int count = 0;
        for (int base = 0; base < N; base += VO_WAVE) {
            const int i = base + lane;
            const bool on = i < N && (!w || w[i]);
            const unsigned long long m = __ballot(on);
            if (on) list[count + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
            count += __popcll(m);
        }
        if (lane == 0) shared_n[0] = count;
    }
    __syncthreads();
    const int n = shared_n[0];

    // 2. hypotheses: rounds of THREADS, the points of each round streamed through LDS
    int best_cnt = 0, best_hyp = 0x7fffffff;
    if (n >= 3) {
        for (int h0 = 0; h0 < iterations; h0 += THREADS) {
            const int hyp = h0 + tid;
            const bool live = hyp < iterations;
            Rigid x;
            if (live) {
                uint32_t r[3], g[3];
                draw_triple(seed, pair, (uint32_t)hyp, (uint32_t)n, r);
                g[0] = list[r[0]]; g[1] = list[r[1]]; g[2] = list[r[2]];
                fit3(p, q, g, x);
            }
            int cnt = 0;
            for (int t0 = 0; t0 < n; t0 += TILE) {
                __syncthreads();
                const int m = min(TILE, n - t0);
                for (int e = tid; e < m * 6; e += THREADS) {
                    const int pt = e / 6, c = e - pt * 6;
                    const int gi = list[t0 + pt];
                    tile[e] = c < 3 ? p[3 * gi + c] : q[3 * gi + c - 3];
                }
                __syncthreads();
                if (live)
                    for (int pt = 0; pt < m; ++pt) cnt += is_inlier(x, tile + 6 * pt, tile + 6 * pt + 3, thr) ? 1 : 0;
            }
            if (live && cnt > best_cnt) { best_cnt = cnt; best_hyp = hyp; }      // strictly greater: first wins
        }
    }

    // 3. block arg-max: largest count, then smallest hypothesis index
    red_cnt[tid] = best_cnt; red_hyp[tid] = best_hyp;
    __syncthreads();
    for (int s = THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) {
            const int c2 = red_cnt[tid + s], h2 = red_hyp[tid + s];
            if (c2 > red_cnt[tid] || (c2 == red_cnt[tid] && h2 < red_hyp[tid])) { red_cnt[tid] = c2; red_hyp[tid] = h2; }
        }
        __syncthreads();
    }
    const int win_cnt = red_cnt[0], win_hyp = red_hyp[0];
    if (win_cnt >= 3 && best_hyp == win_hyp && best_cnt == win_cnt) {          // exactly one lane owns it
        uint32_t r[3], g[3];
        Rigid x;
        draw_triple(seed, pair, (uint32_t)win_hyp, (uint32_t)n, r);
        g[0] = list[r[0]]; g[1] = list[r[1]]; g[2] = list[r[2]];
        fit3(p, q, g, x);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) bestx[3 * i + j] = x.R[i][j];
            bestx[9 + i] = x.t[i];
        }
    }
    __syncthreads();

    // optional inlier mask of the winning hypothesis over ALL N points (0 where unweighted)
    Rigid bx;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) bx.R[i][j] = bestx[3 * i + j];
        bx.t[i] = bestx[9 + i];
    }
    if (mask_out) {
        for (int i = tid; i < N; i += THREADS) {
            const bool on = (!w || w[i]) && win_cnt >= 3 && is_inlier(bx, p + 3 * i, q + 3 * i, thr);
            mask_out[(int64_t)b * N + i] = on ? 1 : 0;
        }
    }
    if (tid >= VO_WAVE) return;

    // 4. wave 0: refit on the inliers (camera_svd_xform.py:130-147), identity otherwise
    double* o = T + (int64_t)b * 16;
    if (win_cnt < 3) {
        if (lane < 16) o[lane] = (lane % 5 == 0) ? 1.0 : 0.0;
        if (lane == 0) n_inl_out[b] = (n >= 3) ? win_cnt : -1;               // -1: fewer than 3 points to sample
        return;
    }
    double sw = 0.0, sp[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
    for (int k = lane; k < n; k += VO_WAVE) {
        const int i = list[k];
        if (!is_inlier(bx, p + 3 * i, q + 3 * i, thr)) continue;
        sw += 1.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) { sp[c] += p[3 * i + c]; sq[c] += q[3 * i + c]; }
    }
    sw = wave_sum(sw);
    double cp[3], cq[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { cp[c] = wave_sum(sp[c]) / sw; cq[c] = wave_sum(sq[c]) / sw; }
    double h[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int k = lane; k < n; k += VO_WAVE) {
        const int i = list[k];
        if (!is_inlier(bx, p + 3 * i, q + 3 * i, thr)) continue;
        double dp[3], dq[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { dp[c] = p[3 * i + c] - cp[c]; dq[c] = q[3 * i + c] - cq[c]; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) h[r][c] += dp[r] * dq[c];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) h[r][c] = wave_sum(h[r][c]);
    double R[3][3];
    rotation_from_covariance(h, R);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            o[4 * r + 0] = R[r][0]; o[4 * r + 1] = R[r][1]; o[4 * r + 2] = R[r][2];
            o[4 * r + 3] = cq[r] - (R[r][0] * cp[0] + R[r][1] * cp[1] + R[r][2] * cp[2]);
        }
        o[12] = 0.0; o[13] = 0.0; o[14] = 0.0; o[15] = 1.0;
        n_inl_out[b] = (int32_t)sw;
    }
}

}  // namespace

extern "C" int vo_ransac_kabsch_batch(const double* P, const double* Q, const uint8_t* weight, double threshold,
                                      int iterations, uint64_t seed, double* T_out, int32_t* n_inliers_out,
                                      uint8_t* inlier_mask_out, int batch, int N, int64_t first_pair, void* stream) {
    VO_REQUIRE(P && Q && T_out && n_inliers_out, "vo_ransac_kabsch_batch: null pointer");
    VO_REQUIRE(batch > 0 && iterations > 0, "vo_ransac_kabsch_batch: bad sizes batch=%d iterations=%d", batch, iterations);
    VO_REQUIRE(first_pair >= 0, "vo_ransac_kabsch_batch: first_pair=%lld is negative", (long long)first_pair);
    VO_REQUIRE(N >= 3, "vo_ransac_kabsch_batch: at least 3 points are required for RANSAC (N=%d)", N);
    if (N > 65535) return vo_fail(VO_ERR_UNSUPPORTED, "vo_ransac_kabsch_batch: N=%d exceeds the 65535-point index list", N);
    const size_t lds = (size_t)(TILE * 6 + 12) * sizeof(double) + (2 * THREADS + 4) * sizeof(int) +
                       (((size_t)N * sizeof(uint16_t) + 15) & ~(size_t)15);
    {   // per-device attribute, cheap to set: no process-wide flag
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ransac_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return vo_fail(VO_ERR_LAUNCH, "vo_ransac_kabsch_batch: LDS attribute: %s", hipGetErrorString(e));
    }
    ransac_kernel<<<dim3((unsigned)batch), dim3(THREADS), lds, vo_stream(stream)>>>(
        P, Q, weight, threshold, iterations, seed, (uint32_t)((uint64_t)first_pair & 0xffffffffu), T_out, n_inliers_out,
        inlier_mask_out, N);
    VO_CHECK_LAUNCH("vo_ransac_kabsch_batch");
    return VO_OK;
}
