"""The batched pipeline on the GPU: chunked processing with the cached halo depth
equals one big chunk, and the per-pair poses equal the NumPy oracle run on the
same device-produced depth / flow maps."""
import numpy as np
import pytest
import torch

from oracle import geometry_ref as ref
from visual_odometry_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _vo(H, W, iters=2, n_kp=120, pose="svd", stereo=None):
    from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
    from visual_odometry_amd.raft.models import RAFTStereoConfig
    cfg = VOConfig(iters=iters, stereo=stereo or RAFTStereoConfig.realtime(), max_keypoints=n_kp,
                   camera_offset=syn.CAMERA_OFFSET, pose=pose, ransac_iterations=200)
    return VisualOdometry(cfg, syn.left_K(W / 640.0), syn.focal_length_px() * (W / 640.0), syn.baseline(),
                          np.ones((H, W), bool), seed=0)


def test_chunking_and_oracle_poses(vo):
    H, W, F = 96, 128, 6
    pipe = _vo(H, W)
    left, right = syn.stereo_sequence(11, F, H, W)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    T_all, sk_all = pipe.run(l, r, chunk_pairs=F - 1)
    T_chunks, sk_chunks = pipe.run(l, r, chunk_pairs=2)
    assert T_all.shape == (F - 1, 4, 4)
    ok = ~(sk_all | sk_chunks)
    assert torch.equal(sk_all, sk_chunks)
    # the batch size changes the order of the plane statistics' atomic sums (and MIOpen's reductions on this small
    # size) a little; poses stay a factor ten inside the north-star tolerance of 1e-3
    assert (T_all[ok] - T_chunks[ok]).abs().max().item() < 1e-4
    # the geometry half against the NumPy oracle on the same maps
    depth = pipe.depth_maps(l, r)
    flow = pipe.flows(l)
    T, skipped = pipe.poses(depth, flow)
    kp = pipe.pts_src.get_keypoints(None, pipe.cfg.max_keypoints)
    K_inv = pipe.pts_xform.K_inv
    d, f = depth.cpu().numpy(), flow.cpu().numpy()
    for i in range(F - 1):
        Tr, wgt = ref.pair_pose(kp, d[i], d[i + 1], f[i], K_inv, offset=np.array(syn.CAMERA_OFFSET))
        if Tr is None or wgt.sum() < 1:
            assert bool(skipped[i])
        elif wgt.sum() >= 3:
            assert not bool(skipped[i])
            np.testing.assert_allclose(T[i].cpu().numpy(), Tr, atol=1e-6)


def test_full_size_chunk_invariance(vo):
    """The benchmark configuration itself (640x480, 12 iterations, 300 keypoints): 20 frame pairs as
    one chunk and as ragged chunks of 7, 7 and 6 with the cached halo depth give the same skip flags
    and the same poses (every convolution runs on the hand-written kernels, whose per-image result
    does not depend on the batch; the tolerance covers the order of the float64 statistics atomics)."""
    H, W, F = 480, 640, 21
    pipe = _vo(H, W, iters=12, n_kp=300)
    left, right = syn.stereo_sequence(21, F, H, W)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    T_all, sk_all = pipe.run(l, r, chunk_pairs=F - 1)
    T_chunks, sk_chunks = pipe.run(l, r, chunk_pairs=7)
    assert T_all.shape == (F - 1, 4, 4) and torch.equal(sk_all, sk_chunks)
    ok = ~sk_all
    assert int(ok.sum()) > 0
    assert (T_all[ok] - T_chunks[ok]).abs().max().item() < 1e-6
    # every pose is a rigid transform
    R = T_all[ok][:, :3, :3]
    eye = torch.eye(3, dtype=R.dtype, device=R.device)
    assert (R @ R.transpose(1, 2) - eye).abs().max().item() < 1e-9
    assert (torch.linalg.det(R) - 1).abs().max().item() < 1e-9


@pytest.mark.parametrize("pose", ["svd", "ransac"])
def test_virtual_shards_equal_the_unsharded_run(vo, pose):
    """Two ranks' shards run one after the other on this GPU -- each from its own halo frame,
    whose depth it recomputes (test_camera_tracking.py:136-150), ragged chunks inside -- give the
    skip flags and poses of the unsharded run, and so does a different chunk size.  With the
    RANSAC estimator this holds because the hypothesis triples follow the pair's index in the
    sequence, not its position in a chunk or shard."""
    from visual_odometry_amd.pipeline import compose_trajectory
    H, W, F = 96, 128, 10
    pipe = _vo(H, W, pose=pose)
    left, right = syn.stereo_sequence(17, F, H, W)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    T_all, sk_all = pipe.run(l, r, chunk_pairs=F - 1)
    T_c, sk_c = pipe.run(l, r, chunk_pairs=4)
    assert torch.equal(sk_all, sk_c)
    parts = [pipe.run_shard(l, r, rank, 2, chunk_pairs=2) for rank in range(2)]
    assert [(p[0], p[1]) for p in parts] == [(0, 5), (5, 9)]
    T_s, sk_s = torch.cat([p[2] for p in parts]), torch.cat([p[3] for p in parts])
    assert torch.equal(sk_all, sk_s) and int((~sk_all).sum()) > 0
    ok = ~sk_all
    # same frames, same weights; at this small size some layers run on MIOpen, whose reduction order
    # depends on the batch: 1e-4 is a tenth of the pose tolerance (the full-size tests, where every
    # convolution is a hand-written kernel with batch-independent per-image results, use 1e-6)
    err_c = (T_all[ok] - T_c[ok]).abs().max().item()
    err_s = (T_all[ok] - T_s[ok]).abs().max().item()
    assert err_c < 1e-4 and err_s < 1e-4, (err_c, err_s)
    Ga = compose_trajectory(T_all.cpu().numpy(), sk_all.cpu().numpy())
    Gs = compose_trajectory(T_s.cpu().numpy(), sk_s.cpu().numpy())
    assert np.abs(Gs - Ga).max() < 1e-3
    # three ranks over 9 pairs, and more ranks than pairs: empty blocks are legal
    parts3 = [pipe.run_shard(l[:3], r[:3], rank, 4, chunk_pairs=8) for rank in range(4)]
    assert [p[2].shape[0] for p in parts3] == [1, 1, 0, 0]
    ok2 = ~sk_all[:2]
    assert torch.equal(torch.cat([p[3] for p in parts3]), sk_all[:2])
    if bool(ok2.any()):
        err3 = (torch.cat([p[2] for p in parts3])[ok2] - T_all[:2][ok2]).abs().max().item()
        assert err3 < 1e-4, err3


def test_eight_virtual_shards_at_full_size(vo):
    """The 8-rank partition of a 640x480 sequence (full RAFT-Stereo arguments, 12 iterations, RANSAC pose as the
    reference's driver runs it), the ranks' shards run one after the other on this GPU: every rank recomputes its
    halo frame's depth, 19 pairs split 3,3,3,3,3,3,1,0, and the gathered rows compose into the unsharded trajectory.
    At this size every convolution is a hand-written kernel whose per-image result does not depend on the batch."""
    from visual_odometry_amd.pipeline import compose_trajectory, shard_pairs
    from visual_odometry_amd.raft.models import RAFTStereoConfig
    H, W, F = 480, 640, 20
    pipe = _vo(H, W, iters=12, n_kp=320, pose="ransac", stereo=RAFTStereoConfig())
    left, right = syn.stereo_sequence(23, F, H, W)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    T_all, sk_all = pipe.run(l, r, chunk_pairs=F - 1)
    parts = [pipe.run_shard(l, r, rank, 8, chunk_pairs=2) for rank in range(8)]
    assert [(p[0], p[1]) for p in parts] == [shard_pairs(F - 1, rank, 8) for rank in range(8)]
    assert [p[2].shape[0] for p in parts] == [3, 3, 3, 3, 3, 3, 1, 0]
    T_s, sk_s = torch.cat([p[2] for p in parts]), torch.cat([p[3] for p in parts])
    assert torch.equal(sk_all, sk_s) and not bool(sk_all.any())
    assert (T_all - T_s).abs().max().item() < 1e-6
    Ga = compose_trajectory(T_all.cpu().numpy(), sk_all.cpu().numpy())
    Gs = compose_trajectory(T_s.cpu().numpy(), sk_s.cpu().numpy())
    assert np.abs(Gs - Ga).max() < 1e-5


def test_host_frames_are_uploaded_chunk_by_chunk(vo):
    """Pinned host frames go up on a second stream, one chunk ahead of the computation; the
    poses equal those from frames already resident in HBM bit for bit."""
    H, W, F = 96, 128, 9
    pipe = _vo(H, W)
    left, right = syn.stereo_sequence(19, F, H, W)
    lh, rh = torch.from_numpy(left).pin_memory(), torch.from_numpy(right).pin_memory()
    T_dev, sk_dev = pipe.run(lh.cuda(), rh.cuda(), chunk_pairs=3)
    T_host, sk_host = pipe.run(lh, rh, chunk_pairs=3)
    assert T_host.is_cuda and torch.equal(sk_dev, sk_host)
    err = (T_host[~sk_host] - T_dev[~sk_dev]).abs().max().item()
    assert err < 1e-4, err      # same chunks; run-to-run rounding of the small-map MIOpen layers and the atomics' order


def test_run_sharded_through_a_single_rank_rccl_group(vo):
    """The product entry point over a real nccl (RCCL) process group of one rank: shard ->
    run -> all_gather on device memory -> host composition equals run() + compose."""
    import os
    import torch.distributed as dist
    from visual_odometry_amd.pipeline import compose_trajectory
    H, W, F = 96, 128, 6
    pipe = _vo(H, W)
    left, right = syn.stereo_sequence(23, F, H, W)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    T, sk = pipe.run(l, r, chunk_pairs=3)
    want = compose_trajectory(T.cpu().numpy(), sk.cpu().numpy())
    # without a process group: the single-GPU loop
    T0, s0, G0 = pipe.run_sharded(l, r, chunk_pairs=3)
    assert np.abs(G0 - want).max() < 1e-3
    assert not dist.is_initialized()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        T1, s1, G1 = pipe.run_sharded(l, r, chunk_pairs=3)
    finally:
        dist.destroy_process_group()
    assert T1.shape == (F - 1, 4, 4) and np.array_equal(s1, sk.cpu().numpy())
    assert np.abs(T1[~s1] - T.cpu().numpy()[~s1]).max() < 1e-4
    assert np.abs(G1 - want).max() < 1e-3


def test_full_args_chunk_invariance_at_full_size(vo):
    """The reference's default RAFT-Stereo arguments (stereo/stereo_disparity_RAFT.py:33-44) at
    640x480, 12 iterations: 8 frame pairs as one chunk and as ragged chunks of 3, 3 and 2."""
    from visual_odometry_amd.raft.models import RAFTStereoConfig
    H, W, F = 480, 640, 9
    pipe = _vo(H, W, iters=12, n_kp=320, stereo=RAFTStereoConfig())
    assert pipe.disparity_solver.model.cfg.n_gru_layers == 3 and pipe.disparity_solver.model.cfg.hidden_dims == [128] * 3
    left, right = syn.stereo_sequence(29, F, H, W)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    T_all, sk_all = pipe.run(l, r, chunk_pairs=F - 1)
    T_chunks, sk_chunks = pipe.run(l, r, chunk_pairs=3)
    assert torch.equal(sk_all, sk_chunks)
    ok = ~sk_all
    assert int(ok.sum()) > 0
    err = (T_all[ok] - T_chunks[ok]).abs().max().item()
    assert err < 1e-5, err


def test_recovers_known_motion_from_exact_maps(vo):
    """Planar scenes with exact depth/flow: the pose kernel chain recovers the
    synthetic camera motion (accuracy is limited by the truncated-pixel depth
    lookups the reference prescribes, not by the kernels)."""
    from tests.conftest import rot_angle
    pipe = _vo(480, 640, n_kp=320)
    rng = np.random.default_rng(5)
    pairs = [syn.plane_pair(rng, 480, 640, holes=False) for _ in range(4)]
    # two-frame "sequences": depth maps of both frames per pair
    for p in pairs:
        depth = torch.from_numpy(np.stack([p["depth1"], p["depth2"]])).cuda()
        flow = torch.from_numpy(p["flow"][None]).cuda()
        T, skipped = pipe.poses(depth, flow)
        assert not bool(skipped[0])
        T = T[0].cpu().numpy()
        t_true = p["t"] + T[:3, :3] @ np.array(syn.CAMERA_OFFSET)
        assert rot_angle(T[:3, :3], p["R"]) < 5e-3
        assert np.abs(T[:3, 3] - t_true).max() < 5e-2


def test_short_chunk_is_rejected(vo):
    pipe = _vo(64, 96)
    z = torch.zeros((1, 64, 96), dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        pipe.process_chunk(z, z)


def test_high_resolution_dense_grid(vo):
    """1280x720 with the dense ~20k-point grid: stereo pads 720 -> 736 rows, the flow
    volume is 14400^2 per pair, Kabsch reduces 19928 weighted points per pair."""
    from oracle import raft_ref
    from visual_odometry_amd.raft_plugins import gray_batch_to_rgb
    H, W, F = 720, 1280, 2
    pipe = _vo(H, W, iters=2, n_kp=20000)
    assert pipe.keypoints.shape == (19928, 2)
    left, right = syn.stereo_sequence(13, F, H, W)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    depth = pipe.depth_maps(l, r)
    flow = pipe.flows(l)
    assert depth.shape == (F, H, W) and flow.shape == (F - 1, 2, H, W)
    # networks against the stock-op restatement at this size
    rgb_l, rgb_r = gray_batch_to_rgb(l), gray_batch_to_rgb(r)
    _, flow_ref = raft_ref.raft_flow_forward(pipe.flow_solver.model, rgb_l[:1], rgb_l[1:], 2)
    assert (flow - flow_ref).pow(2).sum(1).sqrt().mean().item() < 0.05
    pad_l, pads = raft_ref.pad_to_multiple(rgb_l, 32)
    pad_r, _ = raft_ref.pad_to_multiple(rgb_r, 32)
    assert pad_l.shape[-2:] == (736, 1280)
    _, disp_ref = raft_ref.raft_stereo_forward(pipe.disparity_solver.model, pad_l, pad_r, 2)
    disp_ref = raft_ref.unpad(disp_ref, pads)[:, 0]
    disp = pipe.disparity_solver.compute_disparity_gray_batch(l, r)
    assert (disp - disp_ref).abs().mean().item() < 0.05
    # geometry half on the dense grid against the NumPy oracle (bit-exact selection)
    T, skipped = pipe.poses(depth, flow)
    kp = pipe.pts_src.get_keypoints(None, 20000)
    P1, P2, wgt, n_in = ref.pair_correspondences(kp, depth[0].cpu().numpy(), depth[1].cpu().numpy(),
                                                 flow[0].cpu().numpy(), pipe.pts_xform.K_inv)
    from visual_odometry_amd import ops
    _, _, wd, nd = ops.pair_correspondences(pipe.keypoints, depth[:-1], depth[1:], flow, pipe.pts_xform.K_inv)
    assert np.array_equal(wd[0].cpu().numpy().astype(bool), wgt) and int(nd[0]) == n_in
    if wgt.sum() >= 3:
        R, t = ref.kabsch(P1[wgt], P2[wgt], np.array(syn.CAMERA_OFFSET))
        np.testing.assert_allclose(T[0, :3, :3].cpu().numpy(), R, atol=1e-8)
        np.testing.assert_allclose(T[0, :3, 3].cpu().numpy(), t, atol=1e-8)


def test_keypoint_tracks_through_the_pipeline(vo):
    """Depth and flow from the networks, tracks from the one-launch tracker: equal to the
    oracle's frame-by-frame loop on the same maps, and to two k-aligned shards."""
    from visual_odometry_amd.pipeline import shard_segments
    H, W, F, k = 96, 128, 8, 3
    pipe = _vo(H, W)
    left, right = syn.stereo_sequence(12, F, H, W)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    from visual_odometry_amd.geometry import HipKeypointTracker
    kp = pipe.pts_src.get_keypoints(None, 60)
    K = syn.left_K(W / 640.0)
    depth, flow = pipe.depth_maps(l, r), pipe.flows(l)
    tracker = HipKeypointTracker(pipe._cam, kp, k=k)
    tracks, alive, reseeded = [t.cpu().numpy() for t in tracker.track_batch(depth, flow)]
    want = ref.track_keypoints(kp, depth.cpu().numpy(), flow.cpu().numpy(), K, np.linalg.inv(K), k=k)
    assert np.array_equal(alive, want[1]) and np.array_equal(reseeded, want[2]) and alive.any()
    np.testing.assert_allclose(tracks, want[0], rtol=1e-12, atol=1e-10)
    # the one-call entry runs the networks again (MIOpen may pick another solver: rounding-level changes)
    again = [t.cpu().numpy() for t in pipe.track_keypoints(l, r, k=k, max_keypoints=60)]
    assert (again[1] != alive).mean() < 0.02 and np.array_equal(again[2], reseeded)
    assert np.abs(again[0] - tracks)[again[1] & alive].max() < 1e-2
    parts = []
    for rank in range(2):
        s, e = shard_segments(F - 1, k, rank, 2)
        parts.append(pipe.track_keypoints(l[s:e + 1], r[s:e + 1], k=k, max_keypoints=60, first_index=s))
    got_alive = torch.cat([p[1] for p in parts]).cpu().numpy()
    # different batch sizes change MIOpen's reduction order by rounding only
    assert (got_alive != alive).mean() < 0.02
    same = got_alive & alive
    assert np.abs(torch.cat([p[0] for p in parts]).cpu().numpy() - tracks)[same].max() < 1e-2


def test_chunk_with_device_rectification(vo):
    """Raw frames + a rectifier == rectifying first and running the plain pipeline."""
    import yaml
    from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
    from visual_odometry_amd.raft.models import RAFTStereoConfig
    from visual_odometry_amd.rectification import HipStereoRectification

    class P:
        pass
    d = yaml.safe_load(syn.RIG_YAML)
    mk = lambda i: np.array([[i[0], 0, i[2]], [0, i[1], i[3]], [0, 0, 1.0]]) * np.array([[0.25], [0.25], [1.0]])
    p = P()
    p.K_l, p.K_r = mk(d["cam0"]["intrinsics"]), mk(d["cam1"]["intrinsics"])
    p.D_l, p.D_r = np.array(d["cam0"]["distortion_coeffs"]), np.array(d["cam1"]["distortion_coeffs"])
    T = np.array(d["cam1"]["T_cn_cnm1"]); p.R, p.T = T[:3, :3], T[:3, 3]
    H, W, F = 120, 160, 4
    p.resolution = (W, H)
    rect = HipStereoRectification(p)
    stereo_mask = rect.get_rectification_masks()[0]
    cfg = VOConfig(iters=2, stereo=RAFTStereoConfig.realtime(), max_keypoints=80)
    K = rect.P1[:3, :3]
    args = (K, K[0, 0], float(np.linalg.norm(p.T)), stereo_mask)
    left, right = syn.stereo_sequence(13, F, H, W)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    a = VisualOdometry(cfg, *args, seed=0, rectification=rect).process_chunk(l, r)
    b = VisualOdometry(cfg, *args, seed=0).process_chunk(*rect.rectify_batch(l, r))
    # MIOpen may pick another solver on the first call of a shape: rounding-level differences only
    assert torch.equal(a[1], b[1]) and torch.allclose(a[0][~a[1]], b[0][~b[1]], atol=1e-3)
    assert torch.allclose(a[2], b[2], rtol=1e-3, atol=1e-3)


def test_a_step_captures_into_a_hip_graph(vo):
    """One whole step (both networks' refinement loops, depth, poses) records into a HIP graph -- no host
    synchronisation, no allocation outside the capture pool, every launch on the current stream -- and
    the replay gives the eager step's poses (to the run-to-run spread of the plane statistics' atomic sums),
    also after the input frames change in place."""
    H, W, F = 96, 128, 3
    pipe = _vo(H, W, iters=3)
    left, right = syn.stereo_sequence(21, F, H, W)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    _, _, cached = pipe.process_chunk(l, r, None)
    T0, sk0, _ = pipe.process_chunk(l, r, cached)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pipe.process_chunk(l, r, cached)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        T1, sk1, _ = pipe.process_chunk(l, r, cached)
    g.replay()
    torch.cuda.synchronize()
    assert (T1 - T0).abs().max().item() < 1e-5 and torch.equal(sk1, sk0)
    left2, right2 = syn.stereo_sequence(22, F, H, W)
    l2, r2 = torch.from_numpy(left2).cuda(), torch.from_numpy(right2).cuda()
    _, _, cached2 = pipe.process_chunk(l2, r2, None)
    T2, sk2, _ = pipe.process_chunk(l2, r2, cached2)
    l.copy_(l2); r.copy_(r2); cached.copy_(cached2)               # the graph reads the buffers it was recorded on
    g.replay()
    torch.cuda.synchronize()
    assert (T1 - T2).abs().max().item() < 1e-5 and torch.equal(sk1, sk2) and (T2 - T0).abs().max().item() > 1e-3


def test_long_sequence_in_many_chunks_keeps_memory_flat(vo):
    """The long-sequence configuration's loop structure on one GPU: 120 frame pairs from PINNED HOST memory in 60
    chunks of 2 (uploaded chunk by chunk on the copy stream, the halo depth carried from chunk to chunk,
    test_camera_tracking.py:116,143-150) against the same sequence in 8 chunks of 15: same skip flags, the same
    poses, the same composed trajectory; and the peak of device memory over the 60-chunk run is the peak of its
    first few chunks -- nothing accumulates with the length of the sequence but the 128-byte pose rows."""
    from visual_odometry_amd.pipeline import compose_trajectory
    H, W, F = 96, 128, 121
    pipe = _vo(H, W, iters=2)
    left, right = syn.stereo_sequence(31, F, H, W)
    lh, rh = torch.from_numpy(left).pin_memory(), torch.from_numpy(right).pin_memory()
    pipe.run(lh[:7], rh[:7], chunk_pairs=2)                        # warm-up: caches of arranged weights, allocator pools
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    pipe.run(lh[:9], rh[:9], chunk_pairs=2)                        # 4 chunks
    torch.cuda.synchronize()
    peak_short = torch.cuda.max_memory_allocated() - base
    torch.cuda.reset_peak_memory_stats()
    T60, sk60 = pipe.run(lh, rh, chunk_pairs=2)                    # 60 chunks
    torch.cuda.synchronize()
    peak_long = torch.cuda.max_memory_allocated() - base
    assert T60.shape == (F - 1, 4, 4) and sk60.shape == (F - 1,)
    pose_bytes = (F - 1) * (16 * 8 + 1)
    assert peak_long <= peak_short + 4 * pose_bytes + (1 << 20), (peak_short, peak_long)
    T8, sk8 = pipe.run(lh, rh, chunk_pairs=15)
    assert torch.equal(sk60, sk8)
    ok = ~sk60
    assert (T60[ok] - T8[ok]).abs().max().item() < 1e-4
    G60 = compose_trajectory(T60.cpu().numpy(), sk60.cpu().numpy())
    G8 = compose_trajectory(T8.cpu().numpy(), sk8.cpu().numpy())
    assert np.abs(G60[-1] - G8[-1]).max() < 2e-3                   # 120 products of poses that agree to 1e-4
    assert np.abs(G60[-1][:3, :3] @ G60[-1][:3, :3].T - np.eye(3)).max() < 1e-9
