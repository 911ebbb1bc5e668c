"""Seeded synthetic inputs shaped like the UZH-FPV "snapdragon" stereo rig.

There is no dataset offline, so tests, ``bench.py`` and the smoke run all draw
their inputs from here.  Everything is NumPy on the host and deterministic for a
given seed (``numpy.random.default_rng``).

* geometry inputs: a planar scene seen from two camera poses, which gives exact
  depth maps for both frames, the exact frame-1 -> frame-2 flow and the exact
  relative pose, so the recovered pose can be checked against ground truth as
  well as against the reference implementation;
* image inputs: band-limited noise textures warped by a disparity / flow field,
  so the RAFT iterations operate on image-like content.
"""
from __future__ import annotations

import numpy as np

# Kalibr-style rig description with UZH-FPV-like numbers (640x480, ~278 px focal
# length, ~8 cm baseline).  stereo/stereo_params_YAML.py:78-86 turns such a file
# into K_l, K_r, focal_length_px (mean of the two fx) and baseline (norm of T).
RIG_YAML = """
cam0:
  intrinsics: [278.66723066149086, 278.48991409740296, 319.75221200593535, 241.96858910358173]
  distortion_coeffs: [-0.013721808247486035, 0.020727425669427896, -0.012786476702685545, 0.0025242267320687625]
  resolution: [640, 480]
cam1:
  intrinsics: [277.61640629770613, 277.63749695723294, 314.8944703346039, 236.04310050462587]
  distortion_coeffs: [-0.008456929295619607, 0.011407590938612062, -0.006951788325762078, 0.0015368127092821786]
  resolution: [640, 480]
  T_cn_cnm1:
  - [0.9998053017199788, 0.011197738450911484, 0.01624713224548414, -0.07961594300469246]
  - [-0.011147758116324, 0.9999328574031386, -0.0031635699090552542, 0.0007443452072558462]
  - [-0.016281466199246444, 0.00298183486707869, 0.9998630018753686, 0.0004425529195268342]
  - [0.0, 0.0, 0.0, 1.0]
"""

FX_L, FY_L, CX_L, CY_L = 278.66723066149086, 278.48991409740296, 319.75221200593535, 241.96858910358173
FX_R = 277.61640629770613
BASELINE_T = (-0.07961594300469246, 0.0007443452072558462, 0.0004425529195268342)
CAMERA_OFFSET = (0.02172, -6.61e-05, -0.00049)   # test_camera_tracking.py:80


def left_K(scale: float = 1.0) -> np.ndarray:
    """Left intrinsic matrix (float64); ``scale`` = 2 gives a 1280x960-style rig."""
    return np.array([[FX_L * scale, 0.0, CX_L * scale],
                     [0.0, FY_L * scale, CY_L * scale],
                     [0.0, 0.0, 1.0]])


def focal_length_px() -> np.float64:
    return np.float64((FX_L + FX_R) / 2)


def baseline() -> np.float64:
    return np.float64(np.linalg.norm(np.array(BASELINE_T)))


def random_rotation(rng, max_angle: float) -> np.ndarray:
    axis = rng.standard_normal(3)
    axis /= np.linalg.norm(axis)
    ang = rng.uniform(-max_angle, max_angle)
    kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * kx + (1 - np.cos(ang)) * (kx @ kx)


def ellipse_mask(h: int, w: int, shrink: float = 0.92) -> np.ndarray:
    """Boolean validity mask shaped like an eroded rectification footprint."""
    yy, xx = np.mgrid[0:h, 0:w]
    return (((xx - (w - 1) / 2) / (shrink * w / 2)) ** 2
            + ((yy - (h - 1) / 2) / (shrink * h / 2)) ** 2) <= 1.0


def holed_mask(rng, h: int, w: int, holes: int = 12) -> np.ndarray:
    m = np.ones((h, w), dtype=bool)
    for _ in range(holes):
        y0 = int(rng.integers(0, h - 8)); x0 = int(rng.integers(0, w - 8))
        m[y0:y0 + int(rng.integers(4, h // 6)), x0:x0 + int(rng.integers(4, w // 6))] = False
    return m


def smooth_field(rng, h: int, w: int, cells: int = 6) -> np.ndarray:
    """Bilinear up-sampling of a coarse random grid -> smooth field in [0, 1]."""
    g = rng.random((cells + 1, cells + 1))
    ys = np.linspace(0, cells, h); xs = np.linspace(0, cells, w)
    y0 = np.minimum(ys.astype(int), cells - 1); x0 = np.minimum(xs.astype(int), cells - 1)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def plane_pair(rng, h: int = 480, w: int = 640, K: np.ndarray | None = None,
               max_angle: float = 0.05, max_shift: float = 0.15, noise_px: float = 0.0,
               holes: bool = True):
    """One frame pair of a tilted-plane scene with a random rigid camera motion.

    Returns dict(depth1, depth2 (H,W) f32, flow (2,H,W) f32, R, t) where a scene
    point X1 in frame-1 camera coordinates has frame-2 coordinates R @ X1 + t,
    which is the transform a Kabsch fit of (frame-1 points -> frame-2 points)
    recovers.  Zero-depth patches model invalid disparity.
    """
    K = left_K() if K is None else K
    Kinv = np.linalg.inv(K)
    n = np.array([rng.uniform(-0.25, 0.25), rng.uniform(-0.25, 0.25), -1.0])
    n /= np.linalg.norm(n)
    d = -rng.uniform(2.0, 9.0)                      # plane n.X = d, in front of the camera
    R = random_rotation(rng, max_angle)
    t = rng.uniform(-1, 1, 3); t *= rng.uniform(0, max_shift) / np.linalg.norm(t)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    rays = np.stack([xx, yy, np.ones_like(xx)], -1) @ Kinv.T
    z1 = d / (rays @ n)
    n2 = R @ n; d2 = d + n2 @ t
    z2 = d2 / (rays @ n2)
    X1 = rays * z1[..., None]
    X2 = X1 @ R.T + t
    proj = X2 @ K.T
    u2 = proj[..., 0] / proj[..., 2]; v2 = proj[..., 1] / proj[..., 2]
    flow = np.stack([u2 - xx, v2 - yy]).astype(np.float32)
    if noise_px > 0:
        flow += rng.normal(0, noise_px, flow.shape).astype(np.float32)
    depth1 = np.where(z1 > 0, z1, 0).astype(np.float32)
    depth2 = np.where(z2 > 0, z2, 0).astype(np.float32)
    if holes:
        for dm in (depth1, depth2):
            for _ in range(4):
                y0 = int(rng.integers(0, h - 40)); x0 = int(rng.integers(0, w - 40))
                dm[y0:y0 + 24, x0:x0 + 32] = 0.0
    return dict(depth1=depth1, depth2=depth2, flow=flow, R=R, t=t)


def random_geometry(rng, h: int = 480, w: int = 640, flow_sigma: float = 2.0):
    """Unstructured inputs for the keypoint kernels: disparity = -U(0,60) with zero
    patches, N(0, sigma) flow, depth maps with zeros and negatives."""
    disp = -(rng.random((h, w), dtype=np.float32) * 60).astype(np.float32)
    disp[rng.random((h, w)) < 0.03] = 0.0
    disp[h // 3:h // 3 + 20, w // 4:w // 4 + 50] = 0.0
    flow = rng.normal(0, flow_sigma, (2, h, w)).astype(np.float32)
    depth1 = (0.5 + 14.5 * smooth_field(rng, h, w)).astype(np.float32)
    depth2 = (depth1 + rng.normal(0, 0.3, (h, w))).astype(np.float32)
    depth1[rng.random((h, w)) < 0.05] = 0.0
    depth2[rng.random((h, w)) < 0.05] = -1.0
    depth2[rng.random((h, w)) < 0.02] = 20.0
    return dict(disparity=disp, flow=flow, depth1=depth1, depth2=depth2)


def texture(rng, h: int, w: int, octaves=(4, 8, 16, 32, 64)) -> np.ndarray:
    """Band-limited noise texture in [0, 255] float32 (sum of smooth random fields)."""
    img = np.zeros((h, w))
    for i, c in enumerate(octaves):
        img += smooth_field(rng, h, w, c) / (i + 1)
    img -= img.min(); img /= img.max()
    return (255 * img).astype(np.float32)


def warp_rows(img: np.ndarray, shift: np.ndarray) -> np.ndarray:
    """out[y, x] = img[y, x + shift[y, x]] with linear interpolation along x (border clamp)."""
    h, w = img.shape
    xs = np.clip(np.arange(w)[None, :] + shift, 0, w - 1)
    x0 = np.minimum(xs.astype(int), w - 2); f = xs - x0
    rows = np.arange(h)[:, None]
    return img[rows, x0] * (1 - f) + img[rows, x0 + 1] * f


def stereo_sequence(seed: int, frames: int, h: int = 480, w: int = 640):
    """``frames`` synthetic grayscale stereo frames -> (left, right) uint8 (F,H,W).

    Frame f is one texture panned by a per-frame offset (so consecutive left
    frames are related by a small flow) and the right view is the left view
    shifted along x by a smooth disparity of 2..30 px (rectified geometry).
    """
    rng = np.random.default_rng(seed)
    pad = 4 * frames + 64
    big = texture(rng, h + pad, w + pad)
    disp = 2.0 + 28.0 * smooth_field(rng, h, w, 4)
    left = np.empty((frames, h, w), np.uint8); right = np.empty_like(left)
    oy = ox = 0.0
    for f in range(frames):
        oy += rng.uniform(0, 3); ox += rng.uniform(0, 3)
        y0, x0 = int(oy), int(ox)
        fr = big[y0:y0 + h, x0:x0 + w]
        left[f] = np.clip(fr, 0, 255).astype(np.uint8)
        right[f] = np.clip(warp_rows(fr, disp), 0, 255).astype(np.uint8)
    return left, right
