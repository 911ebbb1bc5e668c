"""Batched stereo visual-odometry loop on one GPU, and its sharding over ranks.

This is the reference's tracking loop (test_camera_tracking.py:116-214) turned
inside out: instead of one frame pair per Python iteration with two host/device
round trips, a CHUNK of consecutive frames stays in HBM and every stage runs
once per chunk:

    RAFT-Stereo on each new frame      -> disparity   (:165)
    depth = fx*B/|d|                   -> depth maps  (:166)  one frame's depth is
                                           reused as depth1 of the next pair (:149-150)
    RAFT-Flow on each consecutive pair -> flow        (:168)  the feature encoder runs
                                           once per FRAME, not twice per pair
    grid keypoints, lift, flow-warp, filters -> weighted correspondences (:176-200)
    Kabsch per pair                    -> T_rel 4x4   (:204-207)

Pairs are independent given the frames, so a sequence is sharded contiguously
over ranks (one process per GPU) with a one-frame halo; the only exchange is an
all-gather of the 4x4 relative poses, composed on the host in float64 (:210).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np
import torch

from . import ops
from .geometry import (HipCameraRansacXform, HipCameraSvdXform, HipKeypoints3DFlow, HipKeypoints3DXform,
                       HipKeypointTracker, UniformKeyPoints)
from .raft.models import RAFTStereoConfig
from .raft_plugins import HipDisparityRAFT, HipOpticalFlowRAFT


@dataclass
class VOConfig:
    iters: int = 12                       # refinement iterations of both networks
    stereo: RAFTStereoConfig = field(default_factory=RAFTStereoConfig)
    max_keypoints: int = 320              # test_camera_tracking.py:176
    min_depth: float = 0.0                # :57
    max_depth: float = 15.0               # :58
    max_dz: float = 1.0                   # :198
    min_points: int = 4                   # :188
    camera_offset: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    pose: str = "svd"                     # "svd": CameraSvdXform; "ransac": CameraRansacXform (:82)
    ransac_threshold: float = 0.05        # camera_svd_xform.py:58
    ransac_iterations: int = 1000
    ransac_seed: int = 0
    mixed_precision: bool = False         # the models' switch; the reference passes False (flow_map_RAFT.py:38)
    # how both networks form their heavy float32 products (raft.layers.Arithmetic.kind): "f16x3" = three float16
    # matrix-core terms, the float32 instruction's error; "bf16x3" = three bf16 terms, 16 significant bits, about
    # 5 % faster; "f32" = float32 matrix instructions only
    conv_arithmetic: str = "f16x3"


class _Rig:
    """Minimal camera/stereo parameter holders for the plugin constructors."""

    def __init__(self, K):
        self.K = np.asarray(K, dtype=np.float64)


class VisualOdometry:
    """All per-frame stages of the tracking loop on the current GPU."""

    def __init__(self, cfg: VOConfig, K: np.ndarray, focal_length_px, baseline, mask: np.ndarray,
                 flow_checkpoint: Optional[str] = None, stereo_checkpoint: Optional[str] = None, seed: int = 0,
                 rectification=None):
        self.cfg = cfg
        # optional HipStereoRectification: raw frames are rectified on the device, a chunk at a time
        # (the reference rectifies inside each solver call, flow_map_RAFT.py:64-66, stereo_disparity_RAFT.py:74-75)
        self.rectification = rectification
        self.flow_solver = HipOpticalFlowRAFT(flow_checkpoint, None, cfg.iters, seed=seed,
                                              mixed_precision=cfg.mixed_precision, conv_arithmetic=cfg.conv_arithmetic)
        self.disparity_solver = HipDisparityRAFT(stereo_checkpoint, None, cfg.iters, config=cfg.stereo, seed=seed + 1,
                                                 mixed_precision=cfg.mixed_precision, conv_arithmetic=cfg.conv_arithmetic)
        self.device = self.flow_solver.device
        self._fxB = focal_length_px * baseline
        self._f64 = isinstance(self._fxB, np.float64)
        cam = _Rig(K)
        self.pts_src = UniformKeyPoints(mask)
        self.pts_xform = HipKeypoints3DXform(cam)
        self.pts_flow = HipKeypoints3DFlow(cam, self.pts_xform, mask)
        if cfg.pose == "ransac":
            self.cam_estimator = HipCameraRansacXform(cfg.ransac_threshold, cfg.ransac_iterations, cfg.ransac_seed)
        elif cfg.pose == "svd":
            self.cam_estimator = HipCameraSvdXform(np.array(cfg.camera_offset))
        else:
            raise ValueError("VOConfig.pose must be 'svd' or 'ransac'")
        self.keypoints = self.pts_src.get_keypoints_device(cfg.max_keypoints)
        self._cam = cam
        self._pose_ws = None

    def set_conv_arithmetic(self, kind: str) -> None:
        """Switch both networks of THIS pipeline to another product arithmetic (see VOConfig.conv_arithmetic)."""
        from .raft.layers import CONV_ARITHMETICS
        if kind not in CONV_ARITHMETICS:
            raise ValueError(f"conv_arithmetic must be one of {CONV_ARITHMETICS}")
        self.cfg.conv_arithmetic = kind
        self.flow_solver.model.conv_arithmetic = kind
        self.disparity_solver.model.conv_arithmetic = kind

    # -- stages ------------------------------------------------------------
    def depth_maps(self, left: torch.Tensor, right: torch.Tensor) -> torch.Tensor:
        """(F,H,W) gray GPU frames -> (F,H,W) float32 depth."""
        disp = self.disparity_solver.compute_disparity_gray_batch(left, right)
        return ops.disparity_to_depth(disp, float(self._fxB), self._f64)

    def flows(self, left: torch.Tensor) -> torch.Tensor:
        """(F,H,W) gray GPU frames -> (F-1,2,H,W) flow between consecutive frames."""
        return self.flow_solver.compute_flow_sequence(left)

    def poses(self, depth: torch.Tensor, flow: torch.Tensor, first_pair: int = 0):
        """(F,H,W) depth + (F-1,2,H,W) flow -> T_rel (F-1,4,4) f64, skipped (F-1) bool.
        ``first_pair``: index of pair 0 in the whole sequence (the RANSAC estimator keys its
        hypothesis triples on the global pair index; the Kabsch fit ignores it)."""
        c = self.cfg
        if c.pose == "svd":
            # correspondences -> Kabsch -> skip flag in one call, the correspondences staying in a
            # workspace that is kept across chunks
            T, skipped, self._pose_ws = ops.pair_poses(
                self.keypoints, depth[:-1], depth[1:], flow, self.pts_xform.K_inv, c.min_depth, c.max_depth,
                c.max_dz, self.cam_estimator.offset, c.min_points, self._pose_ws)
            return T, skipped
        P1, P2, wgt, n_in = self.pts_flow.pair_correspondences_batch(
            self.keypoints, depth[:-1], depth[1:], flow, c.min_depth, c.max_depth, c.max_dz)
        T, n_used = self.cam_estimator.compute_camera_xform_batch(P1, P2, wgt, first_pair=first_pair)
        # the reference skips a pair when fewer than 4 points survive the range filter;
        # a pair whose final selection is empty has no pose at all (NaN from Kabsch, an
        # exception from RANSAC, flagged -1 here) and is skipped too
        skipped = (n_in < c.min_points) | (n_used < (0 if c.pose == "ransac" else 1))
        return T, skipped

    # -- chunk / sequence --------------------------------------------------
    @torch.no_grad()
    def process_chunk(self, left: torch.Tensor, right: torch.Tensor, first_depth: Optional[torch.Tensor] = None,
                      first_pair: int = 0):
        """F consecutive frames -> (T_rel (F-1,4,4), skipped (F-1), depth of the last
        frame).  ``first_depth`` is the cached depth of frame 0 (the previous chunk's
        last frame); without it frame 0's stereo pair is evaluated too.  ``first_pair`` is
        the sequence index of the chunk's first pair."""
        if left.shape[0] < 2:
            raise ValueError("a chunk needs at least two frames")
        if self.rectification is not None:
            left, right = self.rectification.rectify_batch(left, right)
        if first_depth is None:
            depth = self.depth_maps(left, right)
        else:
            depth = torch.cat([first_depth[None], self.depth_maps(left[1:], right[1:])], dim=0)
        flow = self.flows(left)
        T, skipped = self.poses(depth, flow, first_pair)
        return T, skipped, depth[-1]

    @torch.no_grad()
    def run(self, left: torch.Tensor, right: torch.Tensor, chunk_pairs: int = 8, first_pair: int = 0):
        """Whole (F,H,W) uint8 sequence -> (T_rel (F-1,4,4), skipped (F-1)) on the GPU, a chunk
        of ``chunk_pairs`` pairs at a time with the last frame's depth carried over
        (test_camera_tracking.py:143-150).  ``first_pair``: sequence index of frame 0 when the
        frames are a shard of a longer sequence.

        Frames on the GPU are sliced in place.  HOST frames (ideally pinned) are uploaded a
        chunk at a time on a second stream, chunk k+1 while chunk k computes -- the
        reference pays one blocking upload and one download per call
        (flow/flow_map_RAFT.py:85,104-106); here the only download is the poses."""
        F = left.shape[0]
        if F < 2:
            z = torch.zeros((0, 4, 4), dtype=torch.float64, device=self.device)
            return z, torch.zeros((0,), dtype=torch.bool, device=self.device)
        bounds = [(s, min(s + chunk_pairs, F - 1)) for s in range(0, F - 1, chunk_pairs)]
        on_host = not left.is_cuda
        if on_host:
            copy_stream = torch.cuda.Stream(device=self.device)
            main = torch.cuda.current_stream(self.device)

            def upload(s, e):
                with torch.cuda.stream(copy_stream):
                    l = left[s:e + 1].to(self.device, non_blocking=True)
                    r = right[s:e + 1].to(self.device, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(copy_stream)
                return l, r, done
            pending = upload(*bounds[0])
        Ts, sk, cached = [], [], None
        for i, (s, e) in enumerate(bounds):
            if on_host:
                l, r, done = pending
                main.wait_event(done)
                l.record_stream(main); r.record_stream(main)      # allocated on the copy stream, used on this one
                if i + 1 < len(bounds):
                    pending = upload(*bounds[i + 1])
            else:
                l, r = left[s:e + 1], right[s:e + 1]
            T, k, cached = self.process_chunk(l, r, cached, first_pair + s)
            Ts.append(T); sk.append(k)
        return torch.cat(Ts), torch.cat(sk)

    @torch.no_grad()
    def run_shard(self, left, right, rank: int, world: int, chunk_pairs: int = 8):
        """This rank's contiguous block of the sequence's pairs: see ``run_shard``."""
        return run_shard(self.run, left, right, rank, world, chunk_pairs)

    @torch.no_grad()
    def run_sharded(self, left, right, chunk_pairs: int = 8, group=None):
        """The whole tracking loop over all ranks of ``group``: see ``run_sharded``."""
        return run_sharded(self.run, left, right, chunk_pairs, group)


    @torch.no_grad()
    def track_keypoints(self, left: torch.Tensor, right: torch.Tensor, k: int = 25, max_keypoints: int = 200,
                        first_index: int = 0):
        """The multi-frame keypoint tracks of test_keypoints_flow.py:235-310 for F
        consecutive frames: depth for every frame, flow for every consecutive pair, then
        all k-frame segments in one launch -> (tracks (F-1,N,3) f64, alive (F-1,N) bool,
        reseeded (F-1) bool) on the GPU.  ``first_index`` is frame 0's index in the whole
        sequence (seedings fall on multiples of ``k`` of the GLOBAL index)."""
        if self.rectification is not None:
            left, right = self.rectification.rectify_batch(left, right)
        grid = self.pts_src.get_keypoints(None, max_keypoints)
        tracker = HipKeypointTracker(self._cam, grid, k=k, min_points=self.cfg.min_points)
        return tracker.track_batch(self.depth_maps(left, right), self.flows(left), first_index)


# ------------------------------------------------------------------ sharding

def shard_segments(n_iterations: int, k: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [start, end) of tracking iterations for ``rank`` with ``start`` a
    multiple of ``k``: keypoint tracks restart at every multiple of k, so blocks cut there
    are independent.  Rank r needs frames start..end inclusive."""
    segments = -(-n_iterations // k)
    per = -(-segments // world)
    start = min(rank * per * k, n_iterations)
    return start, min(start + per * k, n_iterations)


def shard_pairs(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [start, end) of pair indices for ``rank``; blocks have
    ceil(n/world) pairs, the tail ranks may get fewer (or none).  Rank r needs
    frames start..end inclusive (one halo frame)."""
    per = -(-n_pairs // world)
    start = min(rank * per, n_pairs)
    return start, min(start + per, n_pairs)


def run_shard(run, left, right, rank: int, world: int, chunk_pairs: int = 8):
    """Rank ``rank``'s contiguous block of the sequence's pairs (shard_pairs): frames
    start..end inclusive -- the first one is the halo frame whose depth the rank
    recomputes, as the reference does for the first frame it sees
    (test_camera_tracking.py:136-139).  ``left`` / ``right`` hold the WHOLE sequence (host
    or GPU tensors; only the shard's frames are touched); ``run`` is
    ``VisualOdometry.run`` (frames, frames, chunk_pairs, first_pair=) -> (T_rel, skipped).
    Returns (start, end, T_rel (end-start,4,4), skipped (end-start))."""
    n_pairs = left.shape[0] - 1
    start, end = shard_pairs(n_pairs, rank, world)
    if end == start:                      # more ranks than pairs: an empty block still joins the gather
        T, sk = run(left[:0], right[:0], chunk_pairs, first_pair=start)
    else:
        T, sk = run(left[start:end + 1], right[start:end + 1], chunk_pairs, first_pair=start)
    return start, end, T, sk


def run_sharded(run, left, right, chunk_pairs: int = 8, group=None):
    """The whole tracking loop over all ranks of ``group`` (one process per GPU): every
    rank runs its shard, the relative poses are all-gathered (RCCL with the nccl backend)
    and every rank composes the same trajectory on the host in float64, in pair order,
    skipped pairs contributing nothing (test_camera_tracking.py:188-190,205-210).  Without
    an initialised process group this is the single-GPU loop.
    Returns host arrays (T_rel (n,4,4), skipped (n), T_global (n,4,4))."""
    import torch.distributed as dist
    live = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if live else 0
    world = dist.get_world_size(group) if live else 1
    n_pairs = left.shape[0] - 1
    _, _, T, sk = run_shard(run, left, right, rank, world, chunk_pairs)
    T_rel, skipped = gather_poses(T, sk, n_pairs, group)
    return T_rel, skipped, compose_trajectory(T_rel, skipped)


def gather_poses(T_local: torch.Tensor, skipped_local: torch.Tensor, n_pairs: int, group=None):
    """All-gather the per-rank (k,4,4) float64 relative poses and skip flags into
    the full (n_pairs,4,4) / (n_pairs,) arrays on the host, in pair order.  Blocks
    are padded to ceil(n/world) so one equal-sized all_gather suffices (RCCL over
    xGMI with the nccl backend; gloo works the same for CPU tensors)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return T_local.cpu().numpy(), skipped_local.cpu().numpy().astype(bool)
    world = dist.get_world_size(group)
    per = -(-n_pairs // world)
    # the collective runs where the backend works: device memory for nccl (RCCL), host memory for gloo
    dev = T_local.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
    payload = torch.zeros((per, 17), dtype=torch.float64, device=dev)
    k = T_local.shape[0]
    if k > per:
        raise ValueError(f"this rank holds {k} pairs, more than its block of {per}")
    payload[:k, :16] = T_local.reshape(k, 16)
    payload[:k, 16] = skipped_local.to(torch.float64)
    payload[k:, 16] = 1.0                                   # padding rows count as skipped
    out = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(out, payload, group=group)
    full = torch.cat(out)[:n_pairs].cpu().numpy()
    return full[:, :16].reshape(-1, 4, 4).copy(), full[:, 16] > 0.5


def compose_trajectory(T_rel: np.ndarray, skipped: Optional[np.ndarray] = None) -> np.ndarray:
    """T_global <- T_global @ T_rel in float64, in pair order (the product is not
    commutative); skipped pairs leave the pose unchanged
    (test_camera_tracking.py:107,188-190,210).  Returns (n,4,4) global poses."""
    T_rel = np.asarray(T_rel, dtype=np.float64).reshape(-1, 4, 4)
    out = np.empty_like(T_rel)
    G = np.eye(4)
    for i in range(T_rel.shape[0]):
        if skipped is None or not skipped[i]:
            G = G @ T_rel[i]
        out[i] = G
    return out


def write_trajectory(path: str, T_rel: np.ndarray, skipped: Optional[np.ndarray] = None) -> None:
    """The reference's text format: header, then ``i, [t], [R row-major]`` per
    non-skipped pair (test_camera_tracking.py:105,213-214)."""
    with open(path, "w") as f:
        f.write("frame, translation, rotation_matrix_flat\n")
        for i, T in enumerate(np.asarray(T_rel).reshape(-1, 4, 4)):
            if skipped is not None and skipped[i]:
                continue
            f.write(f"{i}, {T[:3, 3].tolist()}, {T[:3, :3].flatten().tolist()}\n")
