// Timeline of the split 3x3 convolution kernel's workgroups in a long launch (steady state, not the first generation):
// every workgroup stamps the chip-wide 100 MHz counter at entry, before its first multiplication, after its main
// loop, after issuing its last store and after its stores have completed, and reports the CU it ran on.
// Printed: the mean length of each phase per generation band, the time a CU slot stays empty between two workgroups,
// and how many of the chip's workgroups are in which phase at sampled instants (are the phases in step chip-wide?).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/conv_timeline.hip -o tools/bin/conv_timeline
// Run:   tools/bin/conv_timeline [N C H W K]      (default 48 64 480 640 64)
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>

#define VO_CONV_TIMELINE
__device__ unsigned long long vo_conv_timeline[10 * (1 << 20)];
#include "../visual_odometry_amd/csrc/vo_conv_split.h"

int vo_fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
    return code;
}

__global__ void fill_random(float* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
}

int main(int argc, char** argv) {
    int N = 48, C = 64, H = 480, W = 640, K = 64;
    if (argc >= 6) { N = atoi(argv[1]); C = atoi(argv[2]); H = atoi(argv[3]); W = atoi(argv[4]); K = atoi(argv[5]); }
    const size_t nx = (size_t)N * C * H * W, ny = (size_t)N * K * H * W, nw = (size_t)K * C * 9;
    float *x, *y, *w; void* arranged;
    hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&w, nw * 4);
    const int Kpad = (K + 63) / 64 * 64;
    hipMalloc(&arranged, (size_t)Kpad * C * 9 * 4);
    fill_random<<<(unsigned)((nx + 255) / 256), 256>>>(x, nx, 1u, 1.0f);
    fill_random<<<(unsigned)((nw + 255) / 256), 256>>>(w, nw, 2u, 1.0f / sqrtf(9.0f * C));
    using Api = SplitApi<PiecesF16>;
    const int wx = 13;                                                         // weights below 1: any exponent keeps them in range
    if (Api::filter("filter", w, arranged, K, C, 9, wx, nullptr)) return 1;
    for (int rep = 0; rep < 2; ++rep)
        if (Api::conv3x3(x, arranged, y, (int64_t)K * H * W, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.0f, N, C, H, W, K, wx, nullptr)) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (Api::conv3x3(x, arranged, y, (int64_t)K * H * W, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0.0f, N, C, H, W, K, wx, nullptr)) return 1;
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int bx = (W + 31) / 32, by = (H + 15) / 16;                         // either geometry has 512-pixel tiles
    const size_t nwg = (size_t)((W + 31) / 32) * ((H + 15) / 16) * N * ((K + 63) / 64);
    (void)bx; (void)by;
    std::vector<unsigned long long> tl10(nwg * 10), tl(nwg * 6);
    hipMemcpyFromSymbol(tl10.data(), HIP_SYMBOL(vo_conv_timeline), nwg * 10 * 8);
    double cseg[3] = {0, 0, 0};
    for (size_t i = 0; i < nwg; ++i) {
        for (int j = 0; j < 6; ++j) tl[i * 6 + j] = tl10[i * 10 + j];
        for (int j = 0; j < 3; ++j) cseg[j] += (double)(tl10[i * 10 + 7 + j] - tl10[i * 10 + 6 + j]);
    }
    printf("second chunk, core clocks (mean): staging %.0f, multiply phase %.0f (216 matrix instructions = 6912), weights request + closing barrier %.0f\n",
           cseg[0] / nwg, cseg[1] / nwg, cseg[2] / nwg);
    unsigned long long t_min = ~0ull, t_max = 0;
    for (size_t i = 0; i < nwg; ++i) { t_min = std::min(t_min, tl[i * 6]); t_max = std::max(t_max, tl[i * 6 + 4]); }
    const double tick_us = 0.01;
    printf("N%d %d->%d @%dx%d: %.3f ms by events, %.3f ms by stamps, %zu workgroups (a workgroup's stamps are its thread 0's)\n", N, C, K, H, W, ms,
           (t_max - t_min) * tick_us * 1e-3, nwg);
    // phases by start-time band
    std::vector<size_t> order(nwg);
    for (size_t i = 0; i < nwg; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return tl[a * 6] < tl[b * 6]; });
    const int bands = 8;
    printf("band (by start time)   prologue   main loop   store issue   store drain   total   [us, means]\n");
    for (int b = 0; b < bands; ++b) {
        double s[4] = {0, 0, 0, 0}; size_t n = 0;
        for (size_t j = nwg * b / bands; j < nwg * (b + 1) / bands; ++j, ++n) {
            const unsigned long long* t = &tl[order[j] * 6];
            s[0] += (t[1] - t[0]) * tick_us; s[1] += (t[2] - t[1]) * tick_us; s[2] += (t[3] - t[2]) * tick_us; s[3] += (t[4] - t[3]) * tick_us;
        }
        printf("  %d                    %7.2f    %7.2f     %7.2f       %7.2f    %7.2f\n", b, s[0] / n, s[1] / n, s[2] / n, s[3] / n, (s[0] + s[1] + s[2] + s[3]) / n);
    }
    // gaps per CU: sort each CU's workgroups by start; with two slots per CU, the gap of a workgroup = its start minus
    // the latest end among earlier workgroups that ended before it started
    std::map<unsigned long long, std::vector<size_t>> per_cu;
    for (size_t i = 0; i < nwg; ++i) {
        const unsigned long long id = tl[i * 6 + 5];
        const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(i);
    }
    double gap_sum = 0; size_t gap_n = 0; double conc_sum = 0; size_t conc_n = 0;
    for (auto& kv : per_cu) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end(), [&](size_t a, size_t b) { return tl[a * 6] < tl[b * 6]; });
        for (size_t j = 2; j < v.size(); ++j) {
            // the slot this workgroup took over: the one of the latest-ending earlier workgroup that ended before its start
            unsigned long long best = 0;
            for (size_t k = (j > 6 ? j - 6 : 0); k < j; ++k) {
                const unsigned long long end = tl[v[k] * 6 + 4];
                if (end <= tl[v[j] * 6] && end > best) best = end;
            }
            if (best) { gap_sum += (tl[v[j] * 6] - best) * tick_us; ++gap_n; }
            int conc = 0;
            for (size_t k = (j > 6 ? j - 6 : 0); k < j; ++k) if (tl[v[k] * 6 + 4] > tl[v[j] * 6]) ++conc;
            conc_sum += conc; ++conc_n;
        }
    }
    printf("%zu CUs seen; a slot stays empty %.2f us between two workgroups (mean over %zu hand-overs); a starting workgroup finds %.2f others on its CU\n",
           per_cu.size(), gap_sum / gap_n, gap_n, conc_sum / conc_n);
    // chip-wide phase census at 16 instants in the middle half of the launch
    printf("instant   in prologue   in main loop   storing   (workgroups chip-wide)\n");
    for (int sidx = 0; sidx < 16; ++sidx) {
        const unsigned long long t = t_min + (t_max - t_min) / 4 + (t_max - t_min) / 2 * sidx / 16;
        size_t a = 0, b = 0, c = 0;
        for (size_t i = 0; i < nwg; ++i) {
            const unsigned long long* q = &tl[i * 6];
            if (t >= q[0] && t < q[1]) ++a; else if (t >= q[1] && t < q[2]) ++b; else if (t >= q[2] && t < q[4]) ++c;
        }
        printf("  %2d        %5zu         %5zu        %5zu\n", sidx, a, b, c);
    }
    return 0;
}
