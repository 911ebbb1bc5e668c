"""CPU restatement of the stereo rectification stage -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference delegates this stage to OpenCV
(stereo/stereo_rectification.py:34-39 ``cv2.stereoRectify`` +
``cv2.initUndistortRectifyMap``, :67-68 ``cv2.remap(..., INTER_LINEAR)``, :110-123 the
validity masks and ``cv2.erode``); OpenCV is not installed here or on the GPU box, its
version is not pinned by the reference, and the reference holds no rectified image or
map to compare with.  What follows restates OpenCV's published algorithms (Bouguet's
rectification, the plumb-bob distortion model with (k1, k2, p1, p2[, k3]), fixed-point
bilinear remapping with 5 fractional bits and a constant zero border, 3x3 erosion that
ignores pixels outside the image).  Nothing under ``visual_odometry_amd/`` imports this
module; tests use it to check the HIP kernels and the product's host-side set-up.
"""
from __future__ import annotations

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def rodrigues_to_matrix(r):
    r = np.asarray(r, dtype=np.float64).reshape(3)
    theta = np.linalg.norm(r)
    if theta < 1e-15:
        return np.eye(3)
    k = r / theta
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(theta) * np.eye(3) + (1 - np.cos(theta)) * np.outer(k, k) + np.sin(theta) * Kx


def matrix_to_rodrigues(R):
    R = np.asarray(R, dtype=np.float64)
    c = (np.trace(R) - 1.0) / 2.0
    c = min(1.0, max(-1.0, c))
    theta = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(v) / 2.0
    if s < 1e-12:
        if c > 0:
            return np.zeros(3)
        # rotation by pi: axis from the diagonal
        axis = np.sqrt(np.maximum((np.diag(R) + 1.0) / 2.0, 0.0))
        if R[0, 1] < 0: axis[1] = -axis[1]
        if R[0, 2] < 0: axis[2] = -axis[2]
        return theta * axis / np.linalg.norm(axis)
    return v / (2.0 * s) * theta


def _dist(D):
    d = np.zeros(8)
    D = np.asarray(D, dtype=np.float64).reshape(-1)
    d[:min(D.size, 8)] = D[:8]
    return d        # k1 k2 p1 p2 k3 k4 k5 k6


def undistort_points(pts, K, D, R=None, P=None, iterations=5):
    """Pixel coordinates -> ideal (undistorted, optionally rotated / re-projected)
    coordinates: fixed-point iteration of the inverse plumb-bob model."""
    k = _dist(D)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    out = np.empty((len(pts), 2))
    for i, (u, v) in enumerate(np.asarray(pts, dtype=np.float64)):
        x0 = x = (u - cx) / fx
        y0 = y = (v - cy) / fy
        for _ in range(iterations):
            r2 = x * x + y * y
            icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
            dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
            dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
            x = (x0 - dx) * icdist
            y = (y0 - dy) * icdist
        if R is not None:
            xx, yy, ww = R @ np.array([x, y, 1.0])
            x, y = xx / ww, yy / ww
        if P is not None:
            x, y = x * P[0, 0] + P[0, 2], y * P[1, 1] + P[1, 2]
        out[i] = (x, y)
    return out


def _inner_rectangle(K, D, R, P, size, n=9):
    """Largest axis-aligned rectangle inside the rectified footprint of an n x n grid."""
    w, h = size
    pts = [(x * (w - 1) / (n - 1), y * (h - 1) / (n - 1)) for y in range(n) for x in range(n)]
    pts = np.float32(pts).astype(np.float64)
    u = undistort_points(pts, K, D, R, P).astype(np.float32).astype(np.float64).reshape(n, n, 2)
    x0, x1 = u[:, 0, 0].max(), u[:, n - 1, 0].min()
    y0, y1 = u[0, :, 1].max(), u[n - 1, :, 1].min()
    return x0, y0, x1 - x0, y1 - y0


def stereo_rectify(K1, D1, K2, D2, size, R, T):
    """Bouguet's algorithm with CALIB_ZERO_DISPARITY and the default free-scaling
    (alpha = -1, same output size) -> R1, R2, P1, P2, Q, roi1, roi2 (x, y, w, h)."""
    K1, K2, R, T = (np.asarray(a, dtype=np.float64) for a in (K1, K2, R, T))
    T = T.reshape(3)
    r_r = rodrigues_to_matrix(-0.5 * matrix_to_rodrigues(R))        # each camera turns half way
    t = r_r @ T
    idx = 0 if abs(t[0]) > abs(t[1]) else 1
    c, nt = t[idx], np.linalg.norm(t)
    uu = np.zeros(3); uu[idx] = 1.0 if c > 0 else -1.0
    ww = np.cross(t, uu)
    nw = np.linalg.norm(ww)
    if nw > 0.0:
        ww *= np.arccos(abs(c) / nt) / nw
    wR = rodrigues_to_matrix(ww)
    R1 = wR @ r_r.T
    R2 = wR @ r_r
    t = R2 @ T
    nx, ny = size
    fc_new = np.inf
    for K, D in ((K1, D1), (K2, D2)):
        dk1 = _dist(D)[0]
        fc = K[idx ^ 1, idx ^ 1]
        if dk1 < 0:
            fc *= 1 + dk1 * (nx * nx + ny * ny) / (4 * fc * fc)
        fc_new = min(fc_new, fc)
    cc = []
    for K, D, Rk in ((K1, D1, R1), (K2, D2, R2)):
        corners = np.float32([(0, 0), (nx - 1, 0), (0, ny - 1), (nx - 1, ny - 1)]).astype(np.float64)
        p = undistort_points(corners, K, D).astype(np.float32).astype(np.float64)
        p3 = np.column_stack([p, np.ones(4)]) @ Rk.T
        proj = (fc_new * p3[:, :2] / p3[:, 2:3]).astype(np.float32).astype(np.float64)
        cc.append(((nx - 1) / 2 - proj[:, 0].mean(), (ny - 1) / 2 - proj[:, 1].mean()))
    cxn = (cc[0][0] + cc[1][0]) * 0.5
    cyn = (cc[0][1] + cc[1][1]) * 0.5
    P1 = np.array([[fc_new, 0, cxn, 0], [0, fc_new, cyn, 0], [0, 0, 1, 0]], dtype=np.float64)
    P2 = P1.copy()
    P2[idx, 3] = t[idx] * fc_new
    Q = np.array([[1, 0, 0, -cxn], [0, 1, 0, -cyn], [0, 0, 0, fc_new],
                  [0, 0, -1.0 / t[idx], 0.0]], dtype=np.float64)
    if idx == 1:
        Q[3, 2] = -1.0 / t[idx]
    rois = []
    for K, D, Rk, Pk in ((K1, D1, R1, P1), (K2, D2, R2, P2)):
        x, y, w, h = _inner_rectangle(K, D, Rk, Pk, size)
        rx0, ry0 = int(np.ceil(x)), int(np.ceil(y))
        rx1, ry1 = rx0 + int(np.floor(w)), ry0 + int(np.floor(h))
        rx0, ry0, rx1, ry1 = max(rx0, 0), max(ry0, 0), min(rx1, nx), min(ry1, ny)
        rois.append((rx0, ry0, max(rx1 - rx0, 0), max(ry1 - ry0, 0)))
    return R1, R2, P1, P2, Q, rois[0], rois[1]


def init_undistort_rectify_map(K, D, R, P, size):
    """For every pixel of the rectified image, where to sample the raw image:
    (map_x, map_y) float32 of shape (h, w)."""
    w, h = size
    k = _dist(D)
    K = np.asarray(K, dtype=np.float64)
    iR = np.linalg.inv(np.asarray(P, dtype=np.float64)[:3, :3] @ np.asarray(R, dtype=np.float64))
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    X = iR[0, 0] * u + iR[0, 1] * v + iR[0, 2]
    Y = iR[1, 0] * u + iR[1, 1] * v + iR[1, 2]
    W = iR[2, 0] * u + iR[2, 1] * v + iR[2, 2]
    x, y = X / W, Y / W
    x2, y2 = x * x, y * y
    r2 = x2 + y2
    _2xy = 2 * x * y
    kr = (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2) / (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2)
    xd = x * kr + k[2] * _2xy + k[3] * (r2 + 2 * x2)
    yd = y * kr + k[2] * (r2 + 2 * y2) + k[3] * _2xy
    return (K[0, 0] * xd + K[0, 2]).astype(np.float32), (K[1, 1] * yd + K[1, 2]).astype(np.float32)


def remap_linear_u8(img, map_x, map_y):
    """Bilinear remap of a uint8 image with 5-bit fixed-point fractions, weights scaled
    to 2^15, constant zero border: dst = (sum w_k * tap_k + 2^14) >> 15."""
    img = np.asarray(img)
    H, W = img.shape
    sx = np.rint(map_x.astype(np.float32) * np.float32(INTER_TAB_SIZE)).astype(np.int64)
    sy = np.rint(map_y.astype(np.float32) * np.float32(INTER_TAB_SIZE)).astype(np.int64)
    ix = np.clip(sx >> INTER_BITS, -32768, 32767)
    iy = np.clip(sy >> INTER_BITS, -32768, 32767)
    fx = sx & (INTER_TAB_SIZE - 1)
    fy = sy & (INTER_TAB_SIZE - 1)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return np.where(ok, img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0).astype(np.int64)

    acc = ((INTER_TAB_SIZE - fy) * (INTER_TAB_SIZE - fx) * 32 * tap(iy, ix) + (INTER_TAB_SIZE - fy) * fx * 32 * tap(iy, ix + 1)
           + fy * (INTER_TAB_SIZE - fx) * 32 * tap(iy + 1, ix) + fy * fx * 32 * tap(iy + 1, ix + 1))
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def erode3x3(mask):
    """3x3 erosion; pixels outside the image do not erode (OpenCV's default border for
    erosion is +infinity)."""
    m = np.asarray(mask, dtype=bool)
    p = np.pad(m, 1, constant_values=True)
    out = np.ones_like(m)
    for dy in range(3):
        for dx in range(3):
            out &= p[dy:dy + m.shape[0], dx:dx + m.shape[1]]
    return out


def rectification_masks(map_l, map_r, size, roi1, roi2):
    """stereo/stereo_rectification.py:110-133 -> (stereo_mask, left_mask, right_mask, roi_mask)."""
    w, h = size
    white = np.full((h, w), 255, dtype=np.uint8)
    left = remap_linear_u8(white, *map_l) > 0
    right = remap_linear_u8(white, *map_r) > 0
    stereo = erode3x3(left & right)
    roi = np.zeros((h, w), dtype=np.uint8)
    for (x, y, rw, rh) in (roi1, roi2):
        roi[y:y + rh, x:x + rw] = 1
    return stereo, left, right, roi
