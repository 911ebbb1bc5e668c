"""Stereo rectification kernels against the NumPy restatement of OpenCV's published
algorithms (oracle/rectify_ref.py).  OpenCV itself is not available, so this stage is
"parity unpinned": the tests pin the HIP path to the restatement, not to cv2."""
import contextlib
import io

import numpy as np
import pytest
import torch
import yaml

from oracle import rectify_ref as rr
from visual_odometry_amd import synthetic as syn

pytestmark = pytest.mark.gpu


class _Params:
    def __init__(self, distortion_scale=1.0, resolution=None):
        d = yaml.safe_load(syn.RIG_YAML)
        mk = lambda i: np.array([[i[0], 0, i[2]], [0, i[1], i[3]], [0, 0, 1.0]])
        self.K_l, self.K_r = mk(d["cam0"]["intrinsics"]), mk(d["cam1"]["intrinsics"])
        self.D_l = np.array(d["cam0"]["distortion_coeffs"]) * distortion_scale
        self.D_r = np.array(d["cam1"]["distortion_coeffs"]) * distortion_scale
        T = np.array(d["cam1"]["T_cn_cnm1"])
        self.R, self.T = T[:3, :3], T[:3, 3]
        self.resolution = tuple(resolution or d["cam0"]["resolution"])


@pytest.mark.parametrize("scale", [1.0, 8.0, 0.0])
def test_rectification_matches_restatement(vo, scale):
    from visual_odometry_amd.rectification import HipStereoRectification
    p = _Params(scale)
    rect = HipStereoRectification(p)
    R1, R2, P1, P2, Q, roi1, roi2 = rr.stereo_rectify(p.K_l, p.D_l, p.K_r, p.D_r, p.resolution, p.R, p.T)
    for got, want in zip(rect.get_rectification_matrices(), (R1, R2, P1, P2, Q)):
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    assert (rect._roi1, rect._roi2) == (roi1, roi2)
    # rectified epipolar geometry: R2 R R1^T is the identity and the new baseline lies on x
    assert np.abs(R2 @ p.R @ R1.T - np.eye(3)).max() < 1e-12 and abs(P2[0, 3]) > 1 and P2[1, 3] == 0
    maps = {}
    for name, K, D, R, P, gx, gy in (("l", p.K_l, p.D_l, R1, P1, rect.map1_l, rect.map2_l),
                                     ("r", p.K_r, p.D_r, R2, P2, rect.map1_r, rect.map2_r)):
        wx, wy = rr.init_undistort_rectify_map(K, D, R, P, p.resolution)
        assert gx.dtype == torch.float32 and tuple(gx.shape) == (p.resolution[1], p.resolution[0])
        # float64 arithmetic on both sides, rounded once to float32: at most one ulp apart
        assert np.abs(gx.cpu().numpy() - wx).max() < 1e-4 and np.abs(gy.cpu().numpy() - wy).max() < 1e-4
        maps[name] = (gx.cpu().numpy(), gy.cpu().numpy())
    # remap of textured frames, bit-exact given the same maps
    rng = np.random.default_rng(4)
    w, h = p.resolution
    frames = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    gl, gr = rect.rectify_batch(torch.from_numpy(frames).cuda(), torch.from_numpy(frames[::-1].copy()).cuda())
    for f in range(3):
        assert np.array_equal(gl[f].cpu().numpy(), rr.remap_linear_u8(frames[f], *maps["l"]))
        assert np.array_equal(gr[f].cpu().numpy(), rr.remap_linear_u8(frames[2 - f], *maps["r"]))
    # single-image interface and the masks
    with contextlib.redirect_stdout(io.StringIO()):
        a, b = rect.rectify_images(frames[0], frames[1])
        only_left = rect.rectify_images(frames[0], None)
        stereo, left, right, roi = rect.get_rectification_masks()
    assert np.array_equal(a, gl[0].cpu().numpy()) and np.array_equal(b, rr.remap_linear_u8(frames[1], *maps["r"]))
    assert only_left[1] is None and np.array_equal(only_left[0], a)
    ws, wl, wr, wroi = rr.rectification_masks(maps["l"], maps["r"], p.resolution, roi1, roi2)
    assert stereo.dtype == np.bool_ and np.array_equal(left, wl) and np.array_equal(right, wr)
    assert np.array_equal(stereo, ws) and np.array_equal(roi, wroi)
    assert 0.5 < stereo.mean() <= 1.0
    assert rect.rectify_images(None, None) is None
    with pytest.raises(TypeError):
        rect.rectify_images(frames[0].astype(np.float32), None)


def test_remap_edge_cases(vo):
    from visual_odometry_amd import ops
    rng = np.random.default_rng(6)
    H, W, Ho, Wo = 37, 53, 24, 40
    img = rng.integers(0, 256, (2, H, W), dtype=np.uint8)
    mx = rng.uniform(-3, W + 2, (Ho, Wo)).astype(np.float32)
    my = rng.uniform(-3, H + 2, (Ho, Wo)).astype(np.float32)
    mx[0, :8] = [0.0, W - 1, W - 1 + 1 / 64, -1.0, -0.984375, W - 0.5, 1e9, np.nan]     # exact pixels, borders, junk
    my[0, :8] = [0.0, H - 1, 5.0, 4.0, H - 1.015625, -0.5, 3.0, 2.0]
    mx[1, :4] = [2.015625, 2.046875, 7.5, 7.484375]                                     # ties of the 1/32 rounding
    got = ops.remap_bilinear_u8(torch.from_numpy(img).cuda(), torch.from_numpy(mx).cuda(), torch.from_numpy(my).cuda())
    for f in range(2):
        want = rr.remap_linear_u8(img[f], np.nan_to_num(mx, nan=-1e6), my)
        assert np.array_equal(got[f].cpu().numpy(), want)
    # identity maps reproduce the image
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(52, dtype=np.float32), indexing="ij")
    same = ops.remap_bilinear_u8(torch.from_numpy(img).cuda(), torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda())
    assert np.array_equal(same.cpu().numpy(), img[:, :, :52])
    with pytest.raises(Exception):
        ops.remap_bilinear_u8(torch.from_numpy(img).cuda(), torch.from_numpy(mx[:, :39].copy()).cuda(),
                              torch.from_numpy(my[:, :39].copy()).cuda())
    m = (rng.random((H, W)) > 0.2).astype(np.uint8)
    assert np.array_equal(ops.erode3x3(torch.from_numpy(m).cuda()).cpu().numpy().astype(bool), rr.erode3x3(m))
