"""HIP-backed drop-ins for the reference's geometry plugins.

Each class keeps the reference's constructor and method signatures (host NumPy in,
freshly allocated host NumPy out, same exceptions) and adds ``*_batch`` methods
that take and return GPU tensors so a whole chunk of frame pairs stays in HBM.

  HipStereoDepth        stereo/stereo_depth.py:4-49
  UniformKeyPoints      keypoints/keypoints_uniform.py:4-59   (host, once per mask)
  HipKeypoints3DXform   keypoints/keypoints_3d.py:5-70
  HipKeypoints3DFlow    keypoints/keypoints_3d_flow.py:8-88
  HipCameraSvdXform     camera/camera_svd_xform.py:4-48
  HipCameraRansacXform  camera/camera_svd_xform.py:52-149
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .interfaces import (CameraXformInterface, KeyPointsInterface, Keypoints3DFlowInterface,
                         Keypoints3DInterface, StereoParamsInterface)


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("the HIP backend needs a visible AMD GPU (there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _to_gpu(a, dtype) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(_device())


# --------------------------------------------------------------------------- depth

class HipStereoDepth:
    """depth = fx * baseline / |disparity|, zero where the disparity is zero."""

    def __init__(self, params):
        if not isinstance(params, StereoParamsInterface):
            raise ValueError("params must be an instance of StereoParamsInterface")
        self.fx = params.focal_length_px
        self.baseline = params.get_baseline()
        if self.fx <= 0 or self.baseline <= 0:
            raise ValueError("Focal length and baseline must be positive values.")
        fxB = self.fx * self.baseline
        # NumPy evaluates float64-scalar / float32-array in float64 and rounds once
        # on assignment; a Python-float or float32 product divides in float32.
        self._f64 = isinstance(fxB, np.float64)
        self._fxB = float(fxB)

    def compute_depth(self, disparity):
        if disparity is None:
            raise ValueError("Disparity map is None.")
        d = np.asarray(disparity)
        out = self.compute_depth_batch(_to_gpu(d, np.float32))
        return out.cpu().numpy().reshape(d.shape)

    def compute_depth_batch(self, disparity: torch.Tensor) -> torch.Tensor:
        """Any-shape float32 GPU tensor of disparities -> depths, same shape."""
        return ops.disparity_to_depth(disparity, self._fxB, self._f64)


# ----------------------------------------------------------------------- keypoints

class UniformKeyPoints(KeyPointsInterface):
    """Aspect-aware uniform grid inside a validity mask.  The grid depends only on
    the mask, so it is built once on the host and cached (host and device)."""

    def __init__(self, rectification_mask: np.ndarray, margin: int = 0):
        self.rectification_mask = rectification_mask
        self.margin = margin
        self._cache = {}

    def get_keypoints(self, image, max_number: int) -> np.ndarray:
        key = int(max_number)
        if key not in self._cache:
            self._cache[key] = self._build(key)
        return self._cache[key].copy()

    def get_keypoints_device(self, max_number: int) -> torch.Tensor:
        """The same grid as an (N,2) float64 GPU tensor (exact: |index| < 2^53)."""
        return _to_gpu(self.get_keypoints(None, max_number), np.float64)

    def _build(self, max_number: int) -> np.ndarray:
        mask = self.rectification_mask
        rows, cols = mask.shape
        left, right = self.margin, cols - self.margin
        top, bottom = self.margin, rows - self.margin
        wide, tall = right - left, bottom - top
        count_x = int(np.sqrt(max_number * (wide / tall)))
        count_y = int(np.sqrt(max_number * (tall / wide)))
        gx = np.linspace(left, right - 1, count_x).astype(int)
        gy = np.linspace(top, bottom - 1, count_y).astype(int)
        grid = np.stack([np.tile(gx, count_y), np.repeat(gy, count_x)], axis=-1)
        grid = grid[mask[grid[:, 1], grid[:, 0]]]
        if len(grid) > max_number:
            grid = grid[np.linspace(0, len(grid) - 1, max_number, dtype=int)]
        return grid


def _check_indexable(kp: np.ndarray, h: int, w: int):
    """The reference indexes depth[int(v), int(u)]; NumPy raises IndexError outside
    [-n, n).  The kernels never fault on such input, but the drop-in reports it."""
    if kp.size == 0:
        return
    iu = np.trunc(kp[:, 0]); iv = np.trunc(kp[:, 1])
    if not (np.all(np.isfinite(kp)) and iu.min() >= -w and iu.max() < w and iv.min() >= -h and iv.max() < h):
        raise IndexError("keypoint outside the image")


class HipKeypoints3DXform(Keypoints3DInterface):
    def __init__(self, camera_params):
        self.camera_params = camera_params
        self.K_inv = np.linalg.inv(self.camera_params.K)

    def to_3d(self, keypoints: np.ndarray, depth_map: np.ndarray) -> np.ndarray:
        if depth_map.ndim != 2:
            raise ValueError("depth_map must be a 2D array.")
        kp = np.asarray(keypoints)
        if kp.shape[0] == 0:
            return np.array([])
        kp = kp.astype(np.float64).reshape(-1, 2)
        _check_indexable(kp, *depth_map.shape)
        out = self.to_3d_batch(_to_gpu(kp, np.float64), _to_gpu(depth_map, np.float32)[None])
        return out[0].cpu().numpy()

    def to_3d_batch(self, keypoints: torch.Tensor, depth: torch.Tensor) -> torch.Tensor:
        """(N,2)|(B,N,2) float64 keypoints, (B,H,W) float32 depth -> (B,N,3) float64."""
        return ops.keypoints_to_3d(keypoints, depth, self.K_inv)

    def to_2d(self, points_3D: np.ndarray) -> np.ndarray:
        """Pinhole projection; rows with Z <= 0 are dropped, so the result can be
        shorter than the input (keypoints/keypoints_3d.py:61-70)."""
        pts = np.asarray(points_3D, dtype=np.float64).reshape(-1, 3)
        if pts.shape[0] == 0:
            return np.array([])
        uv, keep = self.to_2d_batch(_to_gpu(pts, np.float64))
        return uv[keep].cpu().numpy() if bool(keep.any()) else np.array([])

    def to_2d_batch(self, points: torch.Tensor):
        """(...,3) float64 GPU points -> (uv (...,2) float64, keep (...) bool); nothing is
        dropped, keep marks the rows the reference would return."""
        return ops.keypoints_to_2d(points, self.camera_params.K)


class HipKeypoints3DFlow(Keypoints3DFlowInterface):
    def __init__(self, camera_params, keypoints_xform, rectification_mask: np.ndarray):
        self.camera_params = camera_params
        self.keypoints_xform = keypoints_xform
        self.rectification_mask = rectification_mask
        k_inv = getattr(keypoints_xform, "K_inv", None)
        self.K_inv = np.linalg.inv(camera_params.K) if k_inv is None else k_inv

    def compute_2d_flow(self, keypoints: np.ndarray, uv_flow: np.ndarray):
        kp = np.asarray(keypoints).astype(np.float64).reshape(-1, 2)
        _check_indexable(kp, uv_flow.shape[1], uv_flow.shape[2])
        nxt, valid = ops.keypoints_flow_2d(_to_gpu(kp, np.float64), _to_gpu(uv_flow, np.float32)[None])
        return nxt[0].cpu().numpy(), valid[0].cpu().numpy()

    def compute_3d_flow(self, keypoints, depth1, depth2, uv_flow):
        kp = np.asarray(keypoints).astype(np.float64).reshape(-1, 2)
        _check_indexable(kp, uv_flow.shape[1], uv_flow.shape[2])
        p2, ok = self.compute_3d_flow_batch(_to_gpu(kp, np.float64), _to_gpu(depth1, np.float32)[None],
                                            _to_gpu(depth2, np.float32)[None], _to_gpu(uv_flow, np.float32)[None])
        return p2[0].cpu().numpy(), ok[0].cpu().numpy()

    def compute_3d_flow_batch(self, keypoints, depth1, depth2, flow):
        """GPU tensors: depth (B,H,W), flow (B,2,H,W) -> points (B,N,3) f64, mask (B,N) bool."""
        return ops.keypoints_flow_3d(keypoints, depth1, depth2, flow, self.K_inv)

    def pair_correspondences_batch(self, keypoints, depth1, depth2, flow,
                                   min_depth=0.0, max_depth=15.0, max_dz=1.0):
        """The tracking loop's point selection (test_camera_tracking.py:176-200) for B
        pairs: P1, P2 (B,N,3) f64, weight (B,N) uint8, n_in_range (B) int32."""
        return ops.pair_correspondences(keypoints, depth1, depth2, flow, self.K_inv,
                                        min_depth, max_depth, max_dz)


class HipKeypointTracker:
    """The multi-frame mode of test_keypoints_flow.py:235-310 for a whole chunk of frames
    in one launch: seed the uniform grid every ``k`` frames (and whenever fewer than
    ``min_points`` rows are left), advance the survivors through flow and depth, and
    re-project them.  Segments between seedings are independent, so a sequence splits
    across devices at multiples of ``k`` without any exchange."""

    def __init__(self, camera_params, keypoints: np.ndarray, k: int = 25, min_points: int = 4):
        self.K = np.asarray(camera_params.K, dtype=np.float64)
        self.K_inv = np.linalg.inv(self.K)
        self.k, self.min_points = int(k), int(min_points)
        self.keypoints = np.asarray(keypoints).astype(np.float64).reshape(-1, 2)
        self._kp_dev = None

    def track_batch(self, depth: torch.Tensor, flow: torch.Tensor, first_index: int = 0):
        """depth (F,H,W) float32 and flow (F-1,2,H,W) float32 GPU tensors (flow[i]: frame i
        -> i+1) -> tracks (F-1,N,3) float64 (x, y, Z), alive (F-1,N) bool, reseeded (F-1) bool."""
        if self._kp_dev is None or self._kp_dev.device != depth.device:
            self._kp_dev = torch.from_numpy(self.keypoints).to(depth.device)
        return ops.track_keypoints(self._kp_dev, depth, flow, self.K, self.K_inv, self.k, self.min_points,
                                   first_index)

    @staticmethod
    def as_track_lists(tracks: np.ndarray, alive: np.ndarray, reseeded: np.ndarray, iteration: int):
        """The driver's ``tracks_2D`` / ``track_start_z`` lists as they stand after
        ``iteration`` (test_keypoints_flow.py:226-228,298-310): one list of (x, y, z) per
        surviving track, oldest point first."""
        start = max(i for i in range(iteration + 1) if reseeded[i])
        out, start_z = [], []
        for j in np.nonzero(alive[iteration])[0]:
            out.append([tuple(tracks[i, j]) for i in range(start, iteration + 1)])
            start_z.append(float(tracks[start, j, 2]))
        return out, start_z


# ---------------------------------------------------------------------------- pose

class HipCameraSvdXform(CameraXformInterface):
    """Kabsch fit Q ~ R P + t with the camera-offset correction t += R offset."""

    def __init__(self, offset=np.array([0.0, 0.0, 0.0])):
        self.offset = np.array(offset)

    def compute_camera_xform(self, P: np.ndarray, Q: np.ndarray):
        T, _ = self.compute_camera_xform_batch(_to_gpu(np.asarray(P)[None], np.float64),
                                               _to_gpu(np.asarray(Q)[None], np.float64))
        T = T[0].cpu().numpy()
        return T[:3, :3].copy(), T[:3, 3].copy()

    def compute_camera_xform_batch(self, P: torch.Tensor, Q: torch.Tensor, weight: torch.Tensor | None = None,
                                   first_pair: int = 0):
        """(B,N,3) float64 GPU point sets (+ optional (B,N) uint8 weights) ->
        T (B,4,4) float64, n_used (B) int32.  ``first_pair`` is accepted for symmetry with the
        RANSAC estimator and ignored (the fit is deterministic)."""
        return ops.kabsch_batch(P, Q, weight, self.offset)


class HipCameraRansacXform(CameraXformInterface):
    """3-point RANSAC + inlier refit.  ``seed`` fixes the hypothesis triples (the
    reference's own sampling is unseeded and therefore not reproducible)."""

    def __init__(self, threshold: float = 0.05, iterations: int = 1000, seed: int = 0):
        self.threshold = threshold
        self.iterations = iterations
        self.seed = seed

    def compute_camera_xform(self, P: np.ndarray, Q: np.ndarray):
        P, Q = np.asarray(P), np.asarray(Q)
        if P.ndim != 2 or Q.ndim != 2 or P.shape[0] != Q.shape[0] or P.shape[1] != 3 or Q.shape[1] != 3:
            raise ValueError("Input arrays must be of shape (N, 3)")
        if P.shape[0] < 3:
            raise ValueError("At least 3 points are required for RANSAC.")
        T, _ = self.compute_camera_xform_batch(_to_gpu(P[None], np.float64), _to_gpu(Q[None], np.float64))
        T = T[0].cpu().numpy()
        return T[:3, :3].copy(), T[:3, 3].copy()

    def compute_camera_xform_batch(self, P: torch.Tensor, Q: torch.Tensor, weight: torch.Tensor | None = None,
                                   first_pair: int = 0):
        """(B,N,3) float64 GPU point sets (+ optional (B,N) uint8 weights) ->
        T (B,4,4) float64, n_inliers (B) int32 (-1: fewer than 3 weighted points).
        ``first_pair``: sequence index of item 0 (keys the hypothesis triples)."""
        return ops.ransac_kabsch_batch(P, Q, weight, self.threshold, self.iterations, self.seed,
                                       first_pair=first_pair)
