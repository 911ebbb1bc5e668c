"""Seeded input recipes shared by ``make_golden.py`` (which feeds them to the
reference) and by the tests (which feed the same arrays to the oracle and to the
HIP kernels).  Only OUTPUTS are stored in the ``.npz`` fixtures; every fixture
also stores a float64 checksum of its inputs so a drift in NumPy's generators
would be caught instead of silently comparing different inputs.
"""
from __future__ import annotations

import numpy as np

from visual_odometry_amd import synthetic as syn

# (H, W, max_number, margin)
GRID_CASES = [(480, 640, 320, 0), (480, 640, 200, 0), (480, 640, 320, 16),
              (720, 1280, 20000, 0), (720, 1280, 20000, 8)]


def checksum(*arrays) -> float:
    return float(sum(np.nansum(np.asarray(a, dtype=np.float64)) for a in arrays))


def grid_masks(h, w, seed):
    rng = np.random.default_rng(seed)
    return {"full": np.ones((h, w), bool), "holed": syn.holed_mask(rng, h, w),
            "ellipse": syn.ellipse_mask(h, w)}


def grid_for(h, w):
    """The keypoint grid the drivers use at this resolution (mask all true)."""
    from oracle.geometry_ref import uniform_keypoints
    n = 320 if w == 640 else 20000
    return uniform_keypoints(np.ones((h, w), bool), n)


def depth_case(seed=11, h=480, w=640):
    rng = np.random.default_rng(seed)
    return syn.random_geometry(rng, h, w)["disparity"]


def K_for(w):
    return syn.left_K(1.0 if w == 640 else 2.0)


def flow_case(seed, h, w):
    """Random flow/depth maps plus hand-placed boundary cases on the first grid
    points: x+f just below 0, just below w, exactly w, exactly w-1, a tiny
    negative step at x=0, and the same along y."""
    rng = np.random.default_rng(seed)
    g = syn.random_geometry(rng, h, w)
    kp = grid_for(h, w)
    flow = g["flow"]
    eps = np.float32(1e-3)
    edits = [(0, 0, -float(kp[0, 0]) - eps), (1, 0, (w - float(kp[1, 0])) - eps),
             (2, 0, w - float(kp[2, 0])), (3, 0, (w - 1) - float(kp[3, 0])),
             (4, 1, -float(kp[4, 1]) - eps), (5, 1, (h - float(kp[5, 1])) - eps),
             (6, 1, h - float(kp[6, 1])), (7, 1, (h - 1) - float(kp[7, 1])),
             (8, 0, -1e-7), (9, 1, -1e-7), (10, 0, 0.0), (11, 0, 0.999999)]
    for i, c, val in edits:
        flow[c, kp[i, 1], kp[i, 0]] = np.float32(val)
        flow[1 - c, kp[i, 1], kp[i, 0]] = 0.0
    return kp, g["depth1"], g["depth2"], flow


def float_keypoints(seed, n, h, w):
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], 1)


def kabsch_cases(seed=5):
    """name -> (P, Q, offset).  Exact rigid, noisy, offset, mirrored (forces the
    det<0 branch), planar, minimal N."""
    rng = np.random.default_rng(seed)
    out = {}

    def cloud(n, planar=False):
        P = rng.uniform(-3, 3, (n, 3)); P[:, 2] = rng.uniform(0.5, 15, n)
        if planar:
            P[:, 2] = 4.0 + 0.1 * P[:, 0] - 0.2 * P[:, 1]
        return P

    def motion():
        R = syn.random_rotation(rng, 0.05)
        t = rng.uniform(-0.15, 0.15, 3)
        return R, t

    for n in (3, 4, 300, 19928):
        P = cloud(n); R, t = motion()
        out[f"exact_n{n}"] = (P, P @ R.T + t, np.zeros(3))
    for sig in (1e-3, 1e-2):
        P = cloud(300); R, t = motion()
        out[f"noise_{sig:g}"] = (P, P @ R.T + t + rng.normal(0, sig, P.shape), np.zeros(3))
    P = cloud(300); R, t = motion()
    out["offset"] = (P, P @ R.T + t + rng.normal(0, 1e-3, P.shape), np.array(syn.CAMERA_OFFSET))
    P = cloud(300); R, t = motion()
    out["mirrored"] = (P, (P * np.array([1.0, 1.0, -1.0])) @ R.T + t, np.zeros(3))
    P = cloud(300, planar=True); R, t = motion()
    out["planar"] = (P, P @ R.T + t + rng.normal(0, 1e-3, P.shape), np.zeros(3))
    P = cloud(300, planar=True); R, t = motion()
    out["planar_mirrored"] = (P, (P * np.array([-1.0, 1.0, 1.0])) @ R.T + t, np.zeros(3))
    P = cloud(300); R = syn.random_rotation(rng, 3.0); t = rng.uniform(-2, 2, 3)
    out["large_rotation"] = (P, P @ R.T + t, np.zeros(3))
    return out


def kabsch_batch(seed=9, batch=4096, n=64):
    rng = np.random.default_rng(seed)
    P = rng.uniform(-3, 3, (batch, n, 3)); P[..., 2] = rng.uniform(0.5, 15, (batch, n))
    Q = np.empty_like(P)
    for b in range(batch):
        R = syn.random_rotation(rng, 0.2); t = rng.uniform(-0.3, 0.3, 3)
        Q[b] = P[b] @ R.T + t
    Q += rng.normal(0, 5e-3, Q.shape)
    return P, Q


def ransac_cases(seed=61):
    """name -> (P, Q, threshold, iterations, numpy_seed) for CameraRansacXform.  The reference
    draws its triples from NumPy's global stream (camera/camera_svd_xform.py:98), which
    ``np.random.seed(numpy_seed)`` right before the call makes reproducible.
      outliers     120 points, a quarter of them gross outliers
      small_noisy  12 points with noise of the order of the threshold: inlier counts are small
                   integers, many hypotheses tie, so the first-wins rule decides the refit set
      n3           the minimal input (every draw is a permutation of the same triple)
      fallback     unrelated point sets and a tight threshold: no hypothesis reaches 3 inliers
                   and the identity comes back
      two_inliers  a hypothesis can reach 2 inliers but never 3: still the identity"""
    rng = np.random.default_rng(seed)
    out = {}

    def cloud(n):
        P = rng.uniform(-3, 3, (n, 3)); P[:, 2] = rng.uniform(0.5, 15, n)
        return P

    P = cloud(120); R = syn.random_rotation(rng, 0.1); t = rng.uniform(-0.15, 0.15, 3)
    Q = P @ R.T + t + rng.normal(0, 2e-3, P.shape)
    Q[:30] += rng.uniform(-1, 1, (30, 3))
    out["outliers"] = (P, Q, 0.05, 200, 1234)
    P = cloud(12); R = syn.random_rotation(rng, 0.05); t = rng.uniform(-0.15, 0.15, 3)
    out["small_noisy"] = (P, P @ R.T + t + rng.normal(0, 0.04, P.shape), 0.05, 200, 99)
    P = cloud(3); R = syn.random_rotation(rng, 0.05); t = rng.uniform(-0.15, 0.15, 3)
    out["n3"] = (P, P @ R.T + t + rng.normal(0, 1e-3, P.shape), 0.05, 50, 5)
    out["fallback"] = (cloud(40), cloud(40), 1e-3, 100, 7)
    P = cloud(4); Q = cloud(4)
    Q[:2] = P[:2] + np.array([0.1, -0.2, 0.05])         # two points move rigidly, the other two do not
    out["two_inliers"] = (P, Q, 1e-6, 100, 11)
    return out


def trace_case(seed=21, pairs=8, h=480, w=640, noise_px=0.1):
    """Eight planar-scene frame pairs for the end-to-end geometry trace."""
    rng = np.random.default_rng(seed)
    return [syn.plane_pair(rng, h, w, K_for(w), noise_px=noise_px) for _ in range(pairs)]


def trajectory_files(directory, seed=77, n_images=12, n_gt=400):
    """Write a synthetic ``left_images.txt`` and ``groundtruth.txt`` (the UZH-FPV
    text layouts: ``id timestamp image_name`` and ``timestamp tx ty tz qx qy qz qw``,
    '#' comment header, ground-truth rows deliberately not in time order) and return
    (image_list_path, gt_path, computed_positions)."""
    import os
    rng = np.random.default_rng(seed)
    t0 = 1540821844.0
    gt_t = t0 + np.sort(rng.uniform(0, 2.0, n_gt))
    gt = np.column_stack([gt_t, rng.uniform(-5, 5, (n_gt, 3)), rng.normal(0, 1, (n_gt, 4))])
    order = rng.permutation(n_gt)
    img_t = t0 + np.linspace(0.1, 2.6, n_images)        # the last few fall outside the GT span
    img_t[3] = gt_t[50]                                  # exact hit
    img_t[4] = gt_t[60] + 0.05 * 0.999                   # just inside the tolerance of a sparse spot
    gtp = os.path.join(directory, "groundtruth.txt")
    imp = os.path.join(directory, "left_images.txt")
    with open(gtp, "w") as f:
        f.write("# timestamp tx ty tz qx qy qz qw\n")
        for r in gt[order]:
            f.write(" ".join(repr(float(v)) for v in r) + "\n")
    with open(imp, "w") as f:
        f.write("# id timestamp image_name\n")
        for i, t in enumerate(img_t):
            f.write(f"{i} {float(t)!r} img/image_0_{i}.png\n")
    computed = rng.uniform(-1, 1, (n_images + 2, 3))     # two more poses than images
    return imp, gtp, computed


def track_case(seed=41, frames=14, h=120, w=160, n=48):
    """A short sequence for the multi-frame keypoint tracks: per-pixel random flow
    (so re-projected positions are fractional), depth maps with dead patches (tracks
    die), and one frame whose flow throws every point out of the image (the row
    count drops below 4 and the next iteration re-seeds).  Returns
    (grid keypoints, depth (F,h,w), flow (F-1,2,h,w), K)."""
    from oracle.geometry_ref import uniform_keypoints
    rng = np.random.default_rng(seed)
    K = syn.left_K(0.25)
    kp = uniform_keypoints(np.ones((h, w), bool), n)
    depth = rng.uniform(0.5, 12.0, (frames, h, w)).astype(np.float32)
    for f in range(frames):
        for _ in range(6):
            y, x = int(rng.integers(0, h - 20)), int(rng.integers(0, w - 20))
            depth[f, y:y + 20, x:x + 20] = 0.0 if rng.random() < 0.7 else -1.0
    flow = rng.normal(0.0, 2.5, (frames - 1, 2, h, w)).astype(np.float32)
    flow[8] += 1000.0
    return kp, depth, flow, K
