// 3x3 rotation fit used by the batched Kabsch kernel.  Host+device so that the
// arithmetic can also be exercised on the CPU (tests/host_checks.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define VO_HD __host__ __device__ __forceinline__

VO_HD void cross(const double a[3], const double b[3], double c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// One-sided (Hestenes) Jacobi: rotate column pairs of A until they are mutually
// orthogonal, accumulating the rotations in V, so that A_in * V = A_out = U*Sigma.
// Column-major storage: a[c][r] is row r of column c.
VO_HD void jacobi_columns(double a[3][3], double v[3][3]) {
    for (int sweep = 0; sweep < 24; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = (pq == 2) ? 1 : 0;
            const int q = (pq == 0) ? 1 : 2;
            double alpha = a[p][0] * a[p][0] + a[p][1] * a[p][1] + a[p][2] * a[p][2];
            double beta  = a[q][0] * a[q][0] + a[q][1] * a[q][1] + a[q][2] * a[q][2];
            double gamma = a[p][0] * a[q][0] + a[p][1] * a[q][1] + a[p][2] * a[q][2];
            if (gamma != 0.0 && fabs(gamma) > 1e-16 * sqrt(alpha * beta)) {
                rotated = true;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t);
                double s = c * t;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double ap = a[p][r], aq = a[q][r];
                    a[p][r] = c * ap - s * aq;
                    a[q][r] = s * ap + c * aq;
                    double vp = v[p][r], vq = v[q][r];
                    v[p][r] = c * vp - s * vq;
                    v[q][r] = s * vp + c * vq;
                }
            }
        }
        if (!rotated) break;
    }
}

VO_HD void swap_cols(double a[3][3], double v[3][3], double n[3], int i, int j) {
    if (n[i] < n[j]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double t = a[i][r]; a[i][r] = a[j][r]; a[j][r] = t;
            t = v[i][r]; v[i][r] = v[j][r]; v[j][r] = t;
        }
        double t = n[i]; n[i] = n[j]; n[j] = t;
    }
}

// R = V diag(1,1,det(V U^T)) U^T for H = U S V^T, written as the product of the
// two PROPER bases whose third columns are cross products of the leading pair:
// negating the smallest-sigma row of Vt when det < 0 (camera_svd_xform.py:37-39)
// and completing both bases right-handed give the same matrix.
VO_HD void rotation_from_covariance(const double H[3][3], double R[3][3]) {
    double a[3][3], v[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) { a[c][r] = H[r][c]; v[c][r] = (r == c) ? 1.0 : 0.0; }
    jacobi_columns(a, v);
    double n[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) n[c] = sqrt(a[c][0] * a[c][0] + a[c][1] * a[c][1] + a[c][2] * a[c][2]);
    swap_cols(a, v, n, 0, 1);
    swap_cols(a, v, n, 0, 2);
    swap_cols(a, v, n, 1, 2);

    double u1[3], u2[3], u3[3], v3[3];
    if (!(n[0] > 0.0)) {                       // H == 0 (or NaN): LAPACK returns identity bases
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) R[i][j] = (n[0] == 0.0) ? ((i == j) ? 1.0 : 0.0) : n[0];
        return;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) u1[r] = a[0][r] / n[0];
    double d = u1[0] * a[1][0] + u1[1] * a[1][1] + u1[2] * a[1][2];
#pragma unroll
    for (int r = 0; r < 3; ++r) u2[r] = a[1][r] - d * u1[r];
    double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    if (n2 > 1e-300 && n[1] > 1e-15 * n[0]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) u2[r] /= n2;
    } else {
        // rank-1 covariance (collinear points): any unit vector orthogonal to u1
        int k = (fabs(u1[0]) <= fabs(u1[1]) && fabs(u1[0]) <= fabs(u1[2])) ? 0 : (fabs(u1[1]) <= fabs(u1[2]) ? 1 : 2);
        double e[3] = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0};
        cross(u1, e, u2);
        n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
#pragma unroll
        for (int r = 0; r < 3; ++r) u2[r] /= n2;
    }
    cross(u1, u2, u3);
    cross(v[0], v[1], v3);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = v[0][i] * u1[j] + v[1][i] * u2[j] + v3[i] * u3[j];
}

