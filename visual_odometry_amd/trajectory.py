"""Trajectory files and evaluation: the steps right after the hot path.

  read_trajectory          re-reads the ``frame, [t], [R flat]`` text the tracking loop
                           writes (test_camera_tracking.py:253-274)
  world_poses              composes the relative poses and applies the viewer's axis
                           swap [x, y, z] -> [x, z, -y] (:243-287)
  read_ground_truth*       ``timestamp tx ty tz qx qy qz qw`` rows, sorted by time
                           (utilities/data_utils.py:5-93)
  match_ground_truth_positions  nearest ground-truth row per image timestamp within a
                           tolerance (utilities/data_utils.py:95-154)
  absolute_trajectory_error     RMSE after a rigid alignment (the number the
                           reference only shows as a plot overlay)

Host-side NumPy; nothing here is on the per-frame path.
"""
from __future__ import annotations

import ast
import re
from typing import List, Optional, Tuple

import numpy as np

# viewer convention: camera [x, y, z] -> world [x, z, -y]   (test_camera_tracking.py:248-250)
AXIS_SWAP = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, -1.0, 0.0]])


def read_trajectory(path: str, limit: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """-> (frame indices (n,), T_rel (n,4,4)).  Lines without two bracketed lists are
    ignored, as the reference's regex loop does."""
    frames, Ts = [], []
    with open(path) as f:
        f.readline()
        for line in f:
            if limit and len(Ts) >= limit:
                break
            lists = re.findall(r"\[.*?\]", line)
            if len(lists) < 2:
                continue
            T = np.eye(4)
            T[:3, 3] = np.array(ast.literal_eval(lists[0]), dtype=np.float64)
            T[:3, :3] = np.array(ast.literal_eval(lists[1]), dtype=np.float64).reshape(3, 3)
            frames.append(int(line.split(",")[0]))
            Ts.append(T)
    return np.array(frames, dtype=np.int64), np.array(Ts).reshape(-1, 4, 4)


def world_poses(T_rel: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Relative poses -> (world positions (n,3), world transforms (n,4,4)):
    T_global <- T_global @ T_rel, then p_w = M p, R_w = M R M^T."""
    T_rel = np.asarray(T_rel, dtype=np.float64).reshape(-1, 4, 4)
    G = np.eye(4)
    pos = np.empty((len(T_rel), 3))
    Tw = np.tile(np.eye(4), (len(T_rel), 1, 1))
    for i, T in enumerate(T_rel):
        G = G @ T
        pos[i] = AXIS_SWAP @ G[:3, 3]
        Tw[i, :3, :3] = AXIS_SWAP @ G[:3, :3] @ AXIS_SWAP.T
        Tw[i, :3, 3] = pos[i]
    return pos, Tw


def _load_table(path: str, columns: int) -> np.ndarray:
    rows = []
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0].strip()
            if line:
                rows.append(line.split()[:columns])
    return np.array(rows)


def read_ground_truth(path: str) -> np.ndarray:
    """(n,8) float64 rows [timestamp tx ty tz qx qy qz qw], stably sorted by timestamp."""
    tab = _load_table(path, 8).astype(np.float64).reshape(-1, 8)
    return tab[np.argsort(tab[:, 0], kind="stable")]


def read_ground_truth_positions(path: str, skip: int = 1) -> np.ndarray:
    return read_ground_truth(path)[:, 1:4][::skip]


def quaternion_to_rotation_matrix(qx, qy, qz, qw) -> np.ndarray:
    """Unit-normalised quaternion (x, y, z, w) -> 3x3 rotation."""
    n = np.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
    qx, qy, qz, qw = qx / n, qy / n, qz / n, qw / n
    return np.array([
        [1 - 2 * (qy ** 2 + qz ** 2), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
        [2 * (qx * qy + qz * qw), 1 - 2 * (qx ** 2 + qz ** 2), 2 * (qy * qz - qx * qw)],
        [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx ** 2 + qy ** 2)]])


def read_ground_truth_transforms(path: str, skip: int = 1) -> List[np.ndarray]:
    out = []
    for row in read_ground_truth(path)[::skip]:
        T = np.eye(4)
        T[:3, :3] = quaternion_to_rotation_matrix(*row[4:8])
        T[:3, 3] = row[1:4]
        out.append(T)
    return out


def read_image_timestamps(path: str) -> np.ndarray:
    """``id timestamp image_name`` rows -> timestamps sorted ascending."""
    tab = _load_table(path, 3)
    return np.sort(tab[:, 1].astype(np.float64), kind="stable")


def match_ground_truth_positions(computed_positions, image_list_path: str, gt_file_path: str,
                                 tolerance: float = 0.05):
    """List of (computed position, ground-truth position) per computed pose; the
    ground-truth position is the row nearest in time to image i's timestamp if it is
    within ``tolerance`` seconds and [0,0,0] otherwise (also when the image list is
    shorter than the trajectory).  First row wins a tie, as ``idxmin`` does."""
    stamps = read_image_timestamps(image_list_path)
    gt = read_ground_truth(gt_file_path)
    out = []
    for i, p in enumerate(computed_positions):
        g = np.array([0, 0, 0])
        if i < len(stamps) and len(gt):
            d = np.abs(gt[:, 0] - stamps[i])
            j = int(np.argmin(d))
            if d[j] <= tolerance:
                g = gt[j, 1:4].copy()
        out.append((np.array(p), g))
    return out


def absolute_trajectory_error(estimated: np.ndarray, truth: np.ndarray, valid: Optional[np.ndarray] = None):
    """RMSE of position after the best rigid (rotation + translation, no scale)
    alignment of the estimate onto the ground truth -> (rmse, aligned estimate)."""
    est = np.asarray(estimated, dtype=np.float64).reshape(-1, 3)
    gt = np.asarray(truth, dtype=np.float64).reshape(-1, 3)
    sel = np.ones(len(est), bool) if valid is None else np.asarray(valid, bool)
    ce, cg = est[sel].mean(0), gt[sel].mean(0)
    U, _, Vt = np.linalg.svd((est[sel] - ce).T @ (gt[sel] - cg))
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    aligned = (est - ce) @ R.T + cg
    return float(np.sqrt(np.mean(np.sum((aligned[sel] - gt[sel]) ** 2, axis=1)))), aligned
