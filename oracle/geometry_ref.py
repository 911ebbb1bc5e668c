"""NumPy restatement of the reference's host-side geometry stages.

TEST INFRASTRUCTURE ONLY (checker for the HIP kernels and the timed CPU
baseline); the product package never imports it.

Every function names the reference lines it restates.  The restatement is
pinned by ``tests/golden/*.npz``, which were produced by importing the
reference's own modules (``tests/golden/make_golden.py``); keypoint indices and
validity masks must match bit for bit, floating-point outputs to ~1 ulp.

Two flavours are kept for the stages where the reference loops per point:
``*_loop`` mirrors the reference's per-point Python loop (so the CPU baseline
represents what the reference actually costs) and the un-suffixed function is
the vectorised form used as the checker.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# keypoint grid                                   keypoints/keypoints_uniform.py
# --------------------------------------------------------------------------

def uniform_keypoints(mask: np.ndarray, max_number: int, margin: int = 0) -> np.ndarray:
    """Aspect-aware integer grid filtered by a boolean mask -> (N, 2) int64 (x, y).

    Restates keypoints/keypoints_uniform.py:34-59: grid counts from
    ``int(sqrt(n * aspect))`` (:42-43), ``linspace(lo, hi - 1, k).astype(int)``
    (:44-45), row-major mesh with y outer (:48-49), mask filter (:52-53) and an
    even sub-sampling when more than ``max_number`` survive (:56-57).
    """
    h, w = mask.shape
    x_lo, x_hi = margin, w - margin
    y_lo, y_hi = margin, h - margin
    span_x, span_y = x_hi - x_lo, y_hi - y_lo
    nx = int(np.sqrt(max_number * (span_x / span_y)))
    ny = int(np.sqrt(max_number * (span_y / span_x)))
    xs = np.linspace(x_lo, x_hi - 1, nx).astype(int)
    ys = np.linspace(y_lo, y_hi - 1, ny).astype(int)
    pts = np.empty((ny * nx, 2), dtype=xs.dtype)
    pts[:, 0] = np.tile(xs, ny)
    pts[:, 1] = np.repeat(ys, nx)
    pts = pts[mask[pts[:, 1], pts[:, 0]]]
    if len(pts) > max_number:
        pts = pts[np.linspace(0, len(pts) - 1, max_number, dtype=int)]
    return pts


# --------------------------------------------------------------------------
# disparity -> depth                                     stereo/stereo_depth.py
# --------------------------------------------------------------------------

def disparity_to_depth(disparity: np.ndarray, fx, baseline) -> np.ndarray:
    """depth = fx*B/|d| where |d| > 0, else 0; float32 out (stereo_depth.py:41-49).

    ``fx * baseline`` keeps whatever scalar type the caller hands in: with the
    reference's YAML loader both are ``np.float64`` (stereo_params_YAML.py:85-86)
    so the quotient is evaluated in float64 and rounded once into the float32
    output; with plain Python floats NumPy evaluates it in float32.
    """
    valid = np.abs(disparity) > 0
    depth = np.zeros_like(disparity, dtype=np.float32)
    depth[valid] = (fx * baseline) / abs(disparity[valid])
    return depth


# --------------------------------------------------------------------------
# 2-D -> 3-D lift                                      keypoints/keypoints_3d.py
# --------------------------------------------------------------------------

def lift_to_3d_loop(keypoints: np.ndarray, depth: np.ndarray, K_inv: np.ndarray) -> np.ndarray:
    """Per-point loop exactly as keypoints_3d.py:36-49 (used for the CPU baseline)."""
    if depth.ndim != 2:
        raise ValueError("depth_map must be a 2D array.")
    out = []
    for (u, v) in keypoints:
        z = depth[int(v), int(u)]
        if z <= 0:
            out.append([0, 0, 0])
        else:
            ray = K_inv @ np.array([u, v, 1.0])
            x, y = ray[:2] * z
            out.append([x, y, z])
    return np.array(out)


def lift_to_3d(keypoints: np.ndarray, depth: np.ndarray, K_inv: np.ndarray) -> np.ndarray:
    """Vectorised keypoints_3d.py:36-49: Z from the truncated pixel, X/Y from the
    sub-pixel ray ``K_inv @ [u, v, 1]`` scaled by Z; rows with ``Z <= 0`` are zero."""
    if depth.ndim != 2:
        raise ValueError("depth_map must be a 2D array.")
    kp = np.asarray(keypoints)
    if kp.shape[0] == 0:
        return np.zeros((0,))  # np.array([]) in the reference
    u = kp[:, 0].astype(np.float64)
    v = kp[:, 1].astype(np.float64)
    z = depth[kp[:, 1].astype(int), kp[:, 0].astype(int)].astype(np.float64)
    # same association as a row of the 3x3 @ 3 product: (a*u + b*v) + c*1
    rx = (K_inv[0, 0] * u + K_inv[0, 1] * v) + K_inv[0, 2]
    ry = (K_inv[1, 0] * u + K_inv[1, 1] * v) + K_inv[1, 2]
    pts = np.stack([rx * z, ry * z, z], axis=1)
    pts[~(z > 0)] = 0.0
    return pts


def project_to_2d(points_3d: np.ndarray, K: np.ndarray) -> np.ndarray:
    """keypoints_3d.py:61-70: pinhole projection, rows with ``Z <= 0`` are DROPPED."""
    p = np.asarray(points_3d, dtype=np.float64).reshape(-1, 3)
    keep = p[:, 2] > 0
    p = p[keep]
    if p.shape[0] == 0:
        return np.zeros((0,))
    xn, yn = p[:, 0] / p[:, 2], p[:, 1] / p[:, 2]
    u = (K[0, 0] * xn + K[0, 1] * yn) + K[0, 2]
    v = (K[1, 0] * xn + K[1, 1] * yn) + K[1, 2]
    return np.stack([u, v], axis=1)


# --------------------------------------------------------------------------
# keypoint flow                                   keypoints/keypoints_3d_flow.py
# --------------------------------------------------------------------------

def flow_2d(keypoints: np.ndarray, uv_flow: np.ndarray):
    """keypoints_3d_flow.py:36-47.  ``uv_flow`` is channel-first (2, H, W).

    next = kp + flow[:, int(y), int(x)] (integer + float32 promotes to float64),
    valid = 0 <= x < w and 0 <= y < h evaluated BEFORE the clip to
    [0, w-1] x [0, h-1].
    """
    _, h, w = uv_flow.shape
    iy = keypoints[:, 1].astype(int)
    ix = keypoints[:, 0].astype(int)
    step = np.stack([uv_flow[0, iy, ix], uv_flow[1, iy, ix]], axis=1)
    nxt = keypoints + step
    valid = (nxt[:, 0] >= 0) & (nxt[:, 0] < w) & (nxt[:, 1] >= 0) & (nxt[:, 1] < h)
    nxt[:, 0] = np.clip(nxt[:, 0], 0, w - 1)
    nxt[:, 1] = np.clip(nxt[:, 1], 0, h - 1)
    return nxt, valid


def flow_3d(keypoints, depth1, depth2, uv_flow, K_inv, return_first=False):
    """keypoints_3d_flow.py:66-88: flow-warp the keypoints, gather both depths at
    truncated pixels, mask = in-bounds and both depths > 0, lift the warped
    points with depth2; masked-out rows stay zero.  The reference also lifts the
    frame-1 points and discards them (:81-83); ``return_first`` hands them back."""
    nxt, valid = flow_2d(keypoints, uv_flow)
    d1 = depth1[keypoints[:, 1].astype(int), keypoints[:, 0].astype(int)]
    d2 = depth2[nxt[:, 1].astype(int), nxt[:, 0].astype(int)]
    ok = valid & (d1 > 0) & (d2 > 0)
    p1 = np.zeros((keypoints.shape[0], 3))
    p2 = np.zeros((keypoints.shape[0], 3))
    if np.any(ok):
        p1[ok] = lift_to_3d(keypoints[ok], depth1, K_inv)
        p2[ok] = lift_to_3d(nxt[ok], depth2, K_inv)
    if return_first:
        return p2, ok, p1
    return p2, ok


# --------------------------------------------------------------------------
# rigid pose                                        camera/camera_svd_xform.py
# --------------------------------------------------------------------------

def kabsch(P: np.ndarray, Q: np.ndarray, offset=None):
    """camera_svd_xform.py:20-46: centroids, H = Pc^T Qc, LAPACK SVD, R = V U^T with
    the smallest-sigma row of Vt negated when det R < 0, t = cQ - R cP + R offset."""
    cP = np.mean(P, axis=0)
    cQ = np.mean(Q, axis=0)
    H = (P - cP).T @ (Q - cQ)
    U, _, Vt = np.linalg.svd(H)
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        Vt[2, :] *= -1
        R = Vt.T @ U.T
    t = cQ - R @ cP
    if offset is not None:
        t = t + R @ np.asarray(offset, dtype=np.float64)
    return R, t


def ransac_kabsch(P, Q, threshold=0.05, iterations=1000, choice=None):
    """camera_svd_xform.py:84-149: 3-point hypotheses, inlier = residual norm <
    threshold, keep the FIRST hypothesis with the strictly largest inlier count,
    refit on its inliers (identity when fewer than 3).  ``choice(N)`` must return
    three distinct indices; the reference draws them from the unseeded global
    NumPy stream (:98), so callers that want reproducibility pass their own."""
    if P.shape[0] != Q.shape[0] or P.shape[1] != 3 or Q.shape[1] != 3:
        raise ValueError("Input arrays must be of shape (N, 3)")
    n = P.shape[0]
    if n < 3:
        raise ValueError("At least 3 points are required for RANSAC.")
    if choice is None:
        choice = lambda m: np.random.choice(m, 3, replace=False)
    best, best_count = None, 0
    for _ in range(iterations):
        idx = choice(n)
        R, t = kabsch(P[idx], Q[idx])
        err = np.linalg.norm((R @ P.T).T + t - Q, axis=1)
        inl = err < threshold
        c = int(np.sum(inl))
        if c > best_count:
            best_count, best = c, inl
    if best_count >= 3:
        return kabsch(P[best], Q[best])
    return np.eye(3), np.zeros(3)


_M64 = (1 << 64) - 1


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def counter_triples(seed: int, pair: int):
    """The hypothesis triples of the HIP RANSAC kernel (include/vo_hip.h,
    vo_ransac_kabsch_batch) as a ``choice`` callable for ``ransac_kabsch``: call h
    returns triple h of ``pair`` -- the pair's index in the whole SEQUENCE (taken modulo
    2^32), not its position in a chunk -- three distinct indices in [0, n)."""
    state = {"h": 0}

    def choice(n: int):
        s = _splitmix64((seed & _M64) ^ (((pair & 0xFFFFFFFF) << 32) | state["h"]))
        state["h"] += 1
        i0 = s % n
        s = _splitmix64(s)
        i1 = s % (n - 1)
        if i1 >= i0:
            i1 += 1
        s = _splitmix64(s)
        i2 = s % (n - 2)
        lo, hi = min(i0, i1), max(i0, i1)
        if i2 >= lo:
            i2 += 1
        if i2 >= hi:
            i2 += 1
        return np.array([i0, i1, i2])

    return choice


# --------------------------------------------------------------------------
# one frame pair of the tracking loop                  test_camera_tracking.py
# --------------------------------------------------------------------------

def pair_correspondences(kp, depth1, depth2, uv_flow, K_inv,
                         min_depth=0.0, max_depth=15.0, max_dz=1.0, loop=False):
    """test_camera_tracking.py:176-200 with the three boolean compactions folded
    into one 0/1 weight per grid point.

    Returns (P1, P2, weight, n_in_range): frame-1 lift of every grid point
    (:177), frame-2 lift of its flow-warped position (:185, zero where the 3-D
    flow mask is false), weight = in depth range (:180) and 3-D-flow valid
    (:193) and |Z2 - Z1| <= max_dz (:197-198), and the number of points left
    after the range filter, which is what the ``< 4`` skip test looks at (:188).
    """
    lift = lift_to_3d_loop if loop else lift_to_3d
    P1 = lift(kp, depth1, K_inv)
    in_range = (P1[:, 2] >= min_depth) & (P1[:, 2] <= max_depth)
    # The reference warps only the in-range subset; warping every grid point and
    # zero-weighting the rest gives the same weights and the same weighted sums.
    P2, ok = flow_3d(kp, depth1, depth2, uv_flow, K_inv)
    dz_ok = np.abs(P2[:, 2] - P1[:, 2]) <= max_dz
    weight = in_range & ok & dz_ok
    return P1, P2, weight, int(in_range.sum())


def pair_pose(kp, depth1, depth2, uv_flow, K_inv, offset=None,
              min_depth=0.0, max_depth=15.0, max_dz=1.0, loop=False):
    """test_camera_tracking.py:176-207 for one pair -> (T_rel 4x4 or None, weight).
    ``None`` is the reference's ``continue`` when fewer than 4 points survive the
    range filter (:188-190)."""
    P1, P2, wgt, n_in = pair_correspondences(kp, depth1, depth2, uv_flow, K_inv,
                                             min_depth, max_depth, max_dz, loop)
    if n_in < 4:
        return None, wgt
    R, t = kabsch(P1[wgt], P2[wgt], offset)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T, wgt


def compose_trajectory(T_rel, skipped=None) -> np.ndarray:
    """test_camera_tracking.py:107,210: T_global <- T_global @ T_rel in float64, in
    pair order, skipped pairs contributing nothing.  Returns (n, 4, 4) poses, one
    per input pair (a skipped pair repeats the previous pose)."""
    T_rel = np.asarray(T_rel, dtype=np.float64).reshape(-1, 4, 4)
    out = np.empty_like(T_rel)
    G = np.eye(4)
    for i in range(T_rel.shape[0]):
        if skipped is None or not skipped[i]:
            G = G @ T_rel[i]
        out[i] = G
    return out


# --------------------------------------------------------------------------
# multi-frame keypoint tracks                           test_keypoints_flow.py
# --------------------------------------------------------------------------

def track_keypoints(grid_kp, depth, flow, K, K_inv, k=25, min_points=4, first_index=0):
    """test_keypoints_flow.py:235-310, the multi-frame mode, for frames
    ``first_index .. first_index + F - 1`` with depth (F,H,W) and flow (F-1,2,H,W)
    (flow[i]: frame i -> i+1).

    Iteration i either SEEDS (:261-275 when ``i % k == 0`` or nothing is tracked
    yet; :280-291 when fewer than ``min_points`` rows went into the step) -- grid
    points lifted with depth[i], those with Z > 0 kept -- or STEPS (:277-296):
    compute_3d_flow(current, depth[i], depth[i+1], flow[i]), keep the valid rows,
    re-project them with to_2d.  The reference's quirks are kept: a seeding
    iteration applies no flow, so points seeded in frame i are used as frame-(i+1)
    positions by the next step, and the ``< 4`` test looks at the row count BEFORE
    the validity mask is applied (:280).

    Tracks are reported in fixed slots (slot j = grid point j of the latest
    seeding): returns (tracks (F-1,N,3) = (x, y, Z) appended at each iteration,
    zeros for dead slots; alive (F-1,N) bool; reseeded (F-1,) bool)."""
    grid_kp = np.asarray(grid_kp)
    F, N = depth.shape[0], grid_kp.shape[0]
    tracks = np.zeros((max(F - 1, 0), N, 3))
    alive = np.zeros((max(F - 1, 0), N), dtype=bool)
    reseeded = np.zeros(max(F - 1, 0), dtype=bool)
    cur_kp, cur_slots = None, None
    for i in range(F - 1):
        seed = (first_index + i) % k == 0 or cur_kp is None or cur_kp.shape[0] < min_points
        if seed:
            p = lift_to_3d(grid_kp, depth[i], K_inv)
            keep = p[:, 2] > 0
            cur_slots = np.nonzero(keep)[0]
            cur_kp = grid_kp[keep]
            tracks[i, cur_slots, 0:2] = cur_kp
            tracks[i, cur_slots, 2] = p[keep, 2]
        else:
            p2, ok = flow_3d(cur_kp, depth[i], depth[i + 1], flow[i], K_inv)
            p2 = p2[ok]
            cur_slots = cur_slots[ok]
            cur_kp = project_to_2d(p2, K).reshape(-1, 2)
            tracks[i, cur_slots, 0:2] = cur_kp
            tracks[i, cur_slots, 2] = p2[:, 2]
        alive[i, cur_slots] = True
        reseeded[i] = seed
    return tracks, alive, reseeded
