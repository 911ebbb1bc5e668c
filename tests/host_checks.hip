// CPU-side entry points for the host+device math headers, so that the 3x3
// rotation fit can be checked against LAPACK without a GPU.  Built by
// visual_odometry_amd.build.build_host_checks(); only the host half is ever run.
#include "../visual_odometry_amd/csrc/vo_svd3.h"

extern "C" void host_rotation_from_covariance(const double* H9, double* R9) {
    double H[3][3], R[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) H[i][j] = H9[3 * i + j];
    rotation_from_covariance(H, R);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R9[3 * i + j] = R[i][j];
}
