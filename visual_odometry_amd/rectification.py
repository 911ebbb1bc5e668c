"""Stereo rectification with the per-frame work on the GPU.

Drop-in for the reference's ``StereoRectification`` (stereo/stereo_rectification.py),
which calls OpenCV for everything: ``cv2.stereoRectify`` + ``cv2.initUndistortRectifyMap``
once at construction (:34-39), ``cv2.remap`` per image (:67-68) and a remap of a white
image + ``cv2.erode`` for the validity masks (:110-123).  Here the one-time 3x3 algebra
(Bouguet's rectifying rotations, the new camera matrix, the ROI rectangles) runs in
NumPy at construction; the sampling maps, the remap of whole batches of frames and the
mask erosion are HIP kernels.  OpenCV is not a dependency; its results could not be
compared here (it is not installed), so equality with it is by construction of the
published algorithms only.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .interfaces import StereoRectificationInterface


def _rot(axis_angle: np.ndarray) -> np.ndarray:
    """Rotation vector -> matrix."""
    theta = float(np.linalg.norm(axis_angle))
    if theta < 1e-15:
        return np.eye(3)
    k = axis_angle / theta
    S = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.eye(3) + np.sin(theta) * S + (1.0 - np.cos(theta)) * (S @ S)


def _rot_vec(R: np.ndarray) -> np.ndarray:
    """Rotation matrix -> rotation vector."""
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s, c = np.linalg.norm(w) / 2.0, min(1.0, max(-1.0, (np.trace(R) - 1.0) / 2.0))
    theta = np.arccos(c)
    if s < 1e-12:
        if c > 0:
            return np.zeros(3)
        axis = np.sqrt(np.maximum((np.diag(R) + 1.0) / 2.0, 0.0))
        axis[1] *= -1.0 if R[0, 1] < 0 else 1.0
        axis[2] *= -1.0 if R[0, 2] < 0 else 1.0
        return theta * axis / np.linalg.norm(axis)
    return w / (2.0 * s) * theta


def _coeffs(D) -> np.ndarray:
    k = np.zeros(8)
    d = np.asarray(D, dtype=np.float64).reshape(-1)
    k[:min(d.size, 8)] = d[:8]
    return k


def _undistort(pts: np.ndarray, K: np.ndarray, D, R=None, P=None) -> np.ndarray:
    """Raw pixel positions -> ideal positions (five fixed-point iterations of the inverse
    plumb-bob model), optionally rotated by R and projected by P."""
    k = _coeffs(D)
    x0 = (pts[:, 0] - K[0, 2]) / K[0, 0]
    y0 = (pts[:, 1] - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    for _ in range(5):
        r2 = x * x + y * y
        inv = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        x, y = (x0 - dx) * inv, (y0 - dy) * inv
    if R is not None:
        h = np.stack([x, y, np.ones_like(x)], axis=1) @ np.asarray(R).T
        x, y = h[:, 0] / h[:, 2], h[:, 1] / h[:, 2]
    if P is not None:
        x, y = x * P[0, 0] + P[0, 2], y * P[1, 1] + P[1, 2]
    return np.stack([x, y], axis=1)


def _f32(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


def stereo_rectify(K_l, D_l, K_r, D_r, size, R, T):
    """``cv2.stereoRectify(K_l, D_l, K_r, D_r, size, R, T, flags=CALIB_ZERO_DISPARITY)``
    (stereo/stereo_rectification.py:34-36) -> R1, R2, P1, P2, Q, roi1, roi2."""
    K = [np.asarray(K_l, dtype=np.float64), np.asarray(K_r, dtype=np.float64)]
    D = [D_l, D_r]
    R = np.asarray(R, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64).reshape(3)
    nx, ny = int(size[0]), int(size[1])
    half = _rot(-0.5 * _rot_vec(R))                       # both cameras turn half of the relative rotation
    t = half @ T
    axis = 0 if abs(t[0]) > abs(t[1]) else 1              # horizontal or vertical rig
    e = np.zeros(3)
    e[axis] = 1.0 if t[axis] > 0 else -1.0
    w = np.cross(t, e)
    if np.linalg.norm(w) > 0.0:
        w *= np.arccos(abs(t[axis]) / np.linalg.norm(t)) / np.linalg.norm(w)
    align = _rot(w)                                       # brings the baseline onto the image axis
    Rn = [align @ half.T, align @ half]
    t = Rn[1] @ T
    f = np.inf
    for Kc, Dc in zip(K, D):
        fc, k1 = Kc[axis ^ 1, axis ^ 1], _coeffs(Dc)[0]
        if k1 < 0:
            fc *= 1 + k1 * (nx * nx + ny * ny) / (4 * fc * fc)
        f = min(f, fc)
    corners = np.array([[0, 0], [nx - 1, 0], [0, ny - 1], [nx - 1, ny - 1]], dtype=np.float64)
    centres = []
    for Kc, Dc, Rc in zip(K, D, Rn):
        p = _f32(_undistort(corners, Kc, Dc))
        h = np.column_stack([p, np.ones(4)]) @ Rc.T
        proj = _f32(f * h[:, :2] / h[:, 2:3])
        centres.append([(nx - 1) / 2 - proj[:, 0].mean(), (ny - 1) / 2 - proj[:, 1].mean()])
    cx, cy = np.mean(centres, axis=0)                     # zero disparity: one principal point for both
    P1 = np.array([[f, 0, cx, 0], [0, f, cy, 0], [0, 0, 1, 0]], dtype=np.float64)
    P2 = P1.copy()
    P2[axis, 3] = t[axis] * f
    Q = np.array([[1, 0, 0, -cx], [0, 1, 0, -cy], [0, 0, 0, f], [0, 0, -1.0 / t[axis], 0]], dtype=np.float64)
    rois = []
    n = 9
    grid = _f32([(i * (nx - 1) / (n - 1), j * (ny - 1) / (n - 1)) for j in range(n) for i in range(n)])
    for Kc, Dc, Rc, Pc in zip(K, D, Rn, (P1, P2)):
        u = _f32(_undistort(grid, Kc, Dc, Rc, Pc)).reshape(n, n, 2)
        x0, x1 = u[:, 0, 0].max(), u[:, -1, 0].min()
        y0, y1 = u[0, :, 1].max(), u[-1, :, 1].min()
        a, b = int(np.ceil(x0)), int(np.ceil(y0))
        c, d = a + int(np.floor(x1 - x0)), b + int(np.floor(y1 - y0))
        a, b, c, d = max(a, 0), max(b, 0), min(c, nx), min(d, ny)
        rois.append((a, b, max(c - a, 0), max(d - b, 0)))
    return Rn[0], Rn[1], P1, P2, Q, rois[0], rois[1]


class HipStereoRectification(StereoRectificationInterface):
    """Same constructor and methods as the reference's ``StereoRectification``
    (stereo/stereo_rectification.py:6-133); ``rectify_batch`` is the throughput entry."""

    def __init__(self, params):
        if not torch.cuda.is_available():
            raise RuntimeError("HipStereoRectification needs an AMD GPU: there is no CPU path")
        self.params = params
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.left_mask = self.right_mask = self.stereo_mask = self.roi_mask = None
        p = params
        size = tuple(int(v) for v in p.resolution)                    # (width, height), :30
        self.R1, self.R2, self.P1, self.P2, self.Q, self._roi1, self._roi2 = stereo_rectify(
            p.K_l, p.D_l, p.K_r, p.D_r, size, p.R, p.T)
        w, h = size
        self.map1_l, self.map2_l = ops.rectify_maps(p.K_l, p.D_l, self.R1, self.P1, h, w, self.device)
        self.map1_r, self.map2_r = ops.rectify_maps(p.K_r, p.D_r, self.R2, self.P2, h, w, self.device)

    # -- reference interface ---------------------------------------------------
    def rectify_images(self, img_left, img_right):
        if img_left is None and img_right is None:
            return
        ref = img_right if img_right is not None else img_left
        h, w = ref.shape[:2]
        if self.stereo_mask is None or self.stereo_mask.shape != (h, w):
            self._compute_masks(h, w)
        out = []
        for img, mx, my in ((img_left, self.map1_l, self.map2_l), (img_right, self.map1_r, self.map2_r)):
            if img is None:
                out.append(None)
                continue
            a = np.asarray(img)
            if a.dtype != np.uint8 or a.ndim != 2:
                raise TypeError("rectify_images expects 2-D uint8 (grayscale) images")
            out.append(ops.remap_bilinear_u8(torch.from_numpy(np.ascontiguousarray(a)).to(self.device)[None], mx, my)[0]
                       .cpu().numpy())
        return out[0], out[1]

    def get_rectification_masks(self):
        if self.stereo_mask is None or self.roi_mask is None:
            h, w = tuple(self.params.resolution)[::-1]
            self._compute_masks(h, w)
        return self.stereo_mask, self.left_mask, self.right_mask, self.roi_mask

    def get_rectification_matrices(self):
        return self.R1, self.R2, self.P1, self.P2, self.Q

    # -- batched ---------------------------------------------------------------
    def rectify_batch(self, left: torch.Tensor, right: torch.Tensor):
        """(F,H,W) uint8 GPU frames of each camera -> rectified (F,h,w) uint8 GPU frames."""
        return (ops.remap_bilinear_u8(left, self.map1_l, self.map2_l),
                ops.remap_bilinear_u8(right, self.map1_r, self.map2_r))

    def masks_device(self, h: int, w: int):
        """(left, right, stereo) uint8 0/1 GPU masks for raw images of size (h, w)."""
        white = torch.full((1, h, w), 255, device=self.device, dtype=torch.uint8)
        left = (ops.remap_bilinear_u8(white, self.map1_l, self.map2_l)[0] > 0).to(torch.uint8)
        right = (ops.remap_bilinear_u8(white, self.map1_r, self.map2_r)[0] > 0).to(torch.uint8)
        return left, right, ops.erode3x3((left & right).contiguous())

    def _compute_masks(self, h: int, w: int):
        left, right, stereo = self.masks_device(h, w)
        self.left_mask, self.right_mask = left.bool().cpu().numpy(), right.bool().cpu().numpy()
        self.stereo_mask = stereo.bool().cpu().numpy()
        # the reference recomputes the ROIs for this image size (:127-130)
        p = self.params
        roi1, roi2 = stereo_rectify(p.K_l, p.D_l, p.K_r, p.D_r, (w, h), p.R, p.T)[5:7]
        self.roi_mask = np.zeros((h, w), dtype=np.uint8)
        for (x, y, rw, rh) in (roi1, roi2):
            self.roi_mask[y:y + rh, x:x + rw] = 1
