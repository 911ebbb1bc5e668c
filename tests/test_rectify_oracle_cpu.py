"""Self-consistency of the rectification restatement (no OpenCV to compare with): the
rectified cameras must see every 3-D point on the same image row with disparity f*B/Z,
and the sampling maps must agree with the forward camera model."""
import numpy as np
import yaml

from oracle import rectify_ref as rr
from visual_odometry_amd import synthetic as syn


def _rig():
    d = yaml.safe_load(syn.RIG_YAML)
    mk = lambda i: np.array([[i[0], 0, i[2]], [0, i[1], i[3]], [0, 0, 1.0]])
    T = np.array(d["cam1"]["T_cn_cnm1"])
    return (mk(d["cam0"]["intrinsics"]), np.array(d["cam0"]["distortion_coeffs"]), mk(d["cam1"]["intrinsics"]),
            np.array(d["cam1"]["distortion_coeffs"]), tuple(d["cam0"]["resolution"]), T[:3, :3], T[:3, 3])


def _project(K, D, X):
    k = np.zeros(8); k[:len(D)] = D
    x, y = X[:, 0] / X[:, 2], X[:, 1] / X[:, 2]
    r2 = x * x + y * y
    kr = 1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2
    xd = x * kr + 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
    yd = y * kr + k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
    return np.stack([K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]], 1)


def test_rectified_pair_is_row_aligned():
    K1, D1, K2, D2, size, R, T = _rig()
    R1, R2, P1, P2, Q, roi1, roi2 = rr.stereo_rectify(K1, D1, K2, D2, size, R, T)
    for M in (R1, R2):
        assert np.abs(M @ M.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(M) - 1) < 1e-12
    rng = np.random.default_rng(0)
    X = np.column_stack([rng.uniform(-2, 2, 200), rng.uniform(-1.5, 1.5, 200), rng.uniform(1, 12, 200)])
    Xl, Xr = X @ R1.T, (X @ R.T + T) @ R2.T
    ul = Xl[:, :2] / Xl[:, 2:] * P1[0, 0] + P1[:2, 2]
    ur = Xr[:, :2] / Xr[:, 2:] * P2[0, 0] + P2[:2, 2]
    assert np.abs(ul[:, 1] - ur[:, 1]).max() < 1e-9                       # same row
    B = np.linalg.norm(T)
    np.testing.assert_allclose(ul[:, 0] - ur[:, 0], P1[0, 0] * B / Xl[:, 2], rtol=1e-9)    # disparity = f B / Z
    np.testing.assert_allclose(abs(P2[0, 3]), P1[0, 0] * B, rtol=1e-12)
    # reprojection matrix: Q [u v d 1] recovers the rectified 3-D point
    d = ul[:, 0] - ur[:, 0]
    h = np.column_stack([ul, d, np.ones(len(d))]) @ Q.T
    np.testing.assert_allclose(h[:, :3] / h[:, 3:], Xl, rtol=1e-8, atol=1e-9)
    for (x, y, w, h_) in (roi1, roi2):
        assert 0 <= x < x + w <= size[0] and 0 <= y < y + h_ <= size[1] and w * h_ > 0.5 * size[0] * size[1]


def test_maps_follow_the_forward_camera_model():
    K1, D1, K2, D2, size, R, T = _rig()
    R1, R2, P1, P2, *_ = rr.stereo_rectify(K1, D1, K2, D2, size, R, T)
    mx, my = rr.init_undistort_rectify_map(K2, D2, R2, P2, size)
    rng = np.random.default_rng(1)
    u, v = rng.integers(0, size[0], 300), rng.integers(0, size[1], 300)
    rays = np.column_stack([(u - P2[0, 2]) / P2[0, 0], (v - P2[1, 2]) / P2[1, 1], np.ones(300)]) @ R2   # R2^T ray
    want = _project(K2, D2, rays)
    assert np.abs(mx[v, u] - want[:, 0]).max() < 1e-3 and np.abs(my[v, u] - want[:, 1]).max() < 1e-3
    # undistort_points inverts the model
    back = rr.undistort_points(want, K2, D2, R2, P2)
    assert np.abs(back - np.column_stack([u, v])).max() < 1e-2            # five fixed-point iterations


def test_remap_identity_shift_and_border():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (20, 32), dtype=np.uint8)
    ys, xs = np.meshgrid(np.arange(20, dtype=np.float32), np.arange(32, dtype=np.float32), indexing="ij")
    assert np.array_equal(rr.remap_linear_u8(img, xs, ys), img)
    half = rr.remap_linear_u8(img, xs + 0.5, ys)
    want = (img[:, :-1].astype(int) + img[:, 1:].astype(int) + 1) >> 1
    assert np.array_equal(half[:, :-1], want)
    assert np.array_equal(half[:, -1], (img[:, -1].astype(int) + 1) >> 1)          # right tap is the zero border
    assert rr.remap_linear_u8(img, xs - 40, ys).max() == 0
    m = np.ones((5, 6), bool); m[2, 3] = False
    e = rr.erode3x3(m)
    assert e[0, 0] and not e[1:4, 2:5].any() and e.sum() == 30 - 9
