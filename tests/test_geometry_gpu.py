"""HIP geometry kernels (through the C ABI) against the golden vectors captured
from the reference and against the NumPy oracle.  Masks and indices must be
bit-exact; float64 points agree to rounding; poses within 1e-3 rad / 1e-3
(north-star tolerance; the asserts below are far tighter)."""
import numpy as np
import pytest
import torch

from oracle import geometry_ref as ref
from tests.conftest import golden, rot_angle
from tests.golden import cases
from visual_odometry_amd import synthetic as syn

pytestmark = pytest.mark.gpu


class _Cam:
    def __init__(self, K):
        self.K = K


def _params():
    from visual_odometry_amd.interfaces import StereoParamsInterface

    class P(StereoParamsInterface):
        focal_length_px = syn.focal_length_px()
        def load_params(self, input_data): pass
        def get_intrinsics(self): return {}
        def get_baseline(self): return syn.baseline()
        def get_z_max(self, d_min=1.0): return self.focal_length_px * self.get_baseline() / d_min
    return P()


def test_depth_bit_exact(vo):
    from visual_odometry_amd.geometry import HipStereoDepth
    g = golden("g2_depth")
    disp = cases.depth_case()
    solver = HipStereoDepth(_params())
    got = solver.compute_depth(disp)
    assert got.dtype == np.float32 and got.shape == disp.shape
    assert np.array_equal(got, g["depth"])
    # float32 division path (plain Python floats) against NumPy doing the same
    class PF(type(_params())):
        focal_length_px = float(syn.focal_length_px())
        def get_baseline(self): return float(syn.baseline())
    got32 = HipStereoDepth(PF()).compute_depth(disp)
    want32 = ref.disparity_to_depth(disp, float(syn.focal_length_px()), float(syn.baseline()))
    assert np.array_equal(got32, want32)
    # ragged sizes / unaligned views
    for n in (0, 1, 3, 5, 1023):
        d = disp.reshape(-1)[1:1 + n].copy()
        assert np.array_equal(solver.compute_depth(d), ref.disparity_to_depth(d, syn.focal_length_px(), syn.baseline()))
    with pytest.raises(ValueError):
        solver.compute_depth(None)
    with pytest.raises(ValueError):
        HipStereoDepth(object())


def test_keypoint_grid_bit_exact(vo):
    from visual_odometry_amd.geometry import UniformKeyPoints
    g = golden("g1_keypoint_grid")
    for (h, w, n, margin) in cases.GRID_CASES:
        for mname, mask in cases.grid_masks(h, w, seed=h + n + margin).items():
            src = UniformKeyPoints(mask, margin)
            got = src.get_keypoints(None, n)
            assert got.dtype == np.int64
            assert np.array_equal(got, g[f"{h}x{w}_n{n}_m{margin}_{mname}"])
            dev = src.get_keypoints_device(n)
            assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), got.astype(np.float64))


@pytest.mark.parametrize("h,w,seed", [(480, 640, 31), (720, 1280, 32)])
def test_keypoint_flow_matches_reference(vo, h, w, seed):
    from visual_odometry_amd.geometry import HipKeypoints3DFlow, HipKeypoints3DXform
    g = golden(f"g3_flow_{h}x{w}")
    kp, d1, d2, flow = cases.flow_case(seed, h, w)
    cam = _Cam(cases.K_for(w))
    xform = HipKeypoints3DXform(cam)
    assert np.array_equal(xform.K_inv, g["K_inv"])
    kf = HipKeypoints3DFlow(cam, xform, np.ones((h, w), bool))
    nxt, valid = kf.compute_2d_flow(kp, flow)
    assert valid.dtype == np.bool_ and np.array_equal(valid, g["valid2d"])
    assert np.array_equal(nxt, g["next2d"])
    p2, ok = kf.compute_3d_flow(kp, d1, d2, flow)
    assert np.array_equal(ok, g["ok3d"])
    np.testing.assert_allclose(p2, g["p2"], rtol=1e-13, atol=0)
    p1 = xform.to_3d(kp, d1)
    np.testing.assert_allclose(p1, g["p1"], rtol=1e-13, atol=0)
    fkp = cases.float_keypoints(seed + 100, 500, h, w)
    pf = xform.to_3d(fkp, d1)
    np.testing.assert_allclose(pf, g["lift_float"], rtol=1e-13, atol=0)
    np.testing.assert_allclose(xform.to_2d(pf), g["reproj"], rtol=1e-12, atol=1e-10)
    # interface edge cases
    assert xform.to_3d(np.zeros((0, 2)), d1).shape == (0,)
    with pytest.raises(ValueError):
        xform.to_3d(kp, d1[None])
    with pytest.raises(IndexError):
        xform.to_3d(np.array([[w, 0]]), d1)
    # negative indices wrap like NumPy's
    neg = np.array([[-1.0, -2.0], [-w, -h]])
    np.testing.assert_allclose(xform.to_3d(neg, d1), ref.lift_to_3d(neg, d1, xform.K_inv), rtol=1e-13)


def test_pair_correspondences_batch(vo):
    from visual_odometry_amd import ops
    g = golden("g7_trace")
    pairs = cases.trace_case()
    K_inv = np.linalg.inv(cases.K_for(640))
    kp = ref.uniform_keypoints(np.ones((480, 640), bool), 320)
    dev = torch.device("cuda")
    d1 = torch.from_numpy(np.stack([p["depth1"] for p in pairs])).to(dev)
    d2 = torch.from_numpy(np.stack([p["depth2"] for p in pairs])).to(dev)
    fl = torch.from_numpy(np.stack([p["flow"] for p in pairs])).to(dev)
    kpd = torch.from_numpy(kp.astype(np.float64)).to(dev)
    P1, P2, wgt, n_in = ops.pair_correspondences(kpd, d1, d2, fl, K_inv)
    assert np.array_equal(wgt.cpu().numpy().astype(bool), g["weight"])          # bit-exact selection
    assert np.array_equal(n_in.cpu().numpy(), g["n_in_range"])
    T, n_used = ops.kabsch_batch(P1, P2, wgt, syn.CAMERA_OFFSET)
    T = T.cpu().numpy()
    assert np.array_equal(n_used.cpu().numpy(), g["weight"].sum(1))
    for i in range(len(pairs)):
        want1, want2, _, _ = ref.pair_correspondences(kp, pairs[i]["depth1"], pairs[i]["depth2"], pairs[i]["flow"], K_inv)
        np.testing.assert_allclose(P1[i].cpu().numpy(), want1, rtol=1e-13, atol=0)
        np.testing.assert_allclose(P2[i].cpu().numpy(), want2, rtol=1e-13, atol=0)
        assert rot_angle(T[i, :3, :3], g["T_rel"][i, :3, :3]) < 1e-9
        np.testing.assert_allclose(T[i], g["T_rel"][i], atol=1e-9)
    G = ref.compose_trajectory(T)
    np.testing.assert_allclose(G, g["T_global"], atol=1e-8)
    # per-item keypoints (B,N,2) give the same answer as the shared grid
    P1b, P2b, wb, nb = ops.pair_correspondences(kpd[None].repeat(len(pairs), 1, 1).contiguous(), d1, d2, fl, K_inv)
    assert torch.equal(P1b, P1) and torch.equal(P2b, P2) and torch.equal(wb, wgt) and torch.equal(nb, n_in)
    # tight depth range: everything filtered -> n_in_range 0, weights 0, NaN pose (reference: mean of empty)
    _, _, w0, n0 = ops.pair_correspondences(kpd, d1, d2, fl, K_inv, 100.0, 200.0, 1.0)
    assert int(w0.sum()) == 0 and int(n0.sum()) == 0


def test_pair_poses_one_call_equals_the_staged_path_and_the_reference_trace(vo):
    """vo_pair_poses (correspondences in a caller workspace) against the reference's 8-pair trace
    and, bit for bit, against vo_pair_correspondences + vo_kabsch_batch; skip flags for an empty
    depth range; a workspace that is too small or misaligned is rejected."""
    from visual_odometry_amd import _lib, ops
    g = golden("g7_trace")
    pairs = cases.trace_case()
    K_inv = np.linalg.inv(cases.K_for(640))
    kp = ref.uniform_keypoints(np.ones((480, 640), bool), 320)
    dev = torch.device("cuda")
    d1 = torch.from_numpy(np.stack([p["depth1"] for p in pairs])).to(dev)
    d2 = torch.from_numpy(np.stack([p["depth2"] for p in pairs])).to(dev)
    fl = torch.from_numpy(np.stack([p["flow"] for p in pairs])).to(dev)
    kpd = torch.from_numpy(kp.astype(np.float64)).to(dev)
    T, skipped, ws = ops.pair_poses(kpd, d1, d2, fl, K_inv, offset=syn.CAMERA_OFFSET)
    assert not bool(skipped.any()) and ws.numel() == _lib.load().vo_pair_poses_workspace_bytes(len(pairs), kp.shape[0])
    np.testing.assert_allclose(T.cpu().numpy(), g["T_rel"], atol=1e-9)
    P1, P2, wgt, n_in = ops.pair_correspondences(kpd, d1, d2, fl, K_inv)
    T2, _ = ops.kabsch_batch(P1, P2, wgt, syn.CAMERA_OFFSET)
    assert torch.equal(T, T2)
    # the workspace is reused (dirty) by a smaller second call
    T3, sk3, ws3 = ops.pair_poses(kpd[:100].contiguous(), d1[:3], d2[:3], fl[:3], K_inv, offset=syn.CAMERA_OFFSET, workspace=ws)
    assert ws3 is ws
    P1, P2, wgt, n_in = ops.pair_correspondences(kpd[:100].contiguous(), d1[:3], d2[:3], fl[:3], K_inv)
    assert torch.equal(T3, ops.kabsch_batch(P1, P2, wgt, syn.CAMERA_OFFSET)[0]) and not bool(sk3.any())
    # nothing inside the depth range: every pair skipped (test_camera_tracking.py:188-190)
    _, sk0, _ = ops.pair_poses(kpd, d1, d2, fl, K_inv, 100.0, 200.0, 1.0, workspace=ws)
    assert bool(sk0.all())
    # min_points counts the rows left by the range filter
    _, skm, _ = ops.pair_poses(kpd, d1, d2, fl, K_inv, min_points=int(g["n_in_range"].max()) + 1, workspace=ws)
    assert bool(skm.all())
    small = torch.empty((1024,), dtype=torch.uint8, device=dev)
    args = lambda w, n: ("vo_pair_poses", kpd.data_ptr(), 0, d1.data_ptr(), d2.data_ptr(), fl.data_ptr(), ops._host9(K_inv),
                         0.0, 15.0, 1.0, None, 4, T.data_ptr(), skipped.data_ptr(), w, n, len(pairs), kp.shape[0], 480, 640, 0)
    with pytest.raises(ValueError):
        _lib.call(*args(small.data_ptr(), small.numel()))
    with pytest.raises(ValueError):
        _lib.call(*args(ws.data_ptr() + 8, ws.numel() - 8))


def test_kabsch_cases_and_batch(vo):
    from visual_odometry_amd import ops
    from visual_odometry_amd.geometry import HipCameraSvdXform
    g = golden("g5_kabsch")
    for name, (P, Q, off) in cases.kabsch_cases().items():
        R, t = HipCameraSvdXform(off).compute_camera_xform(P, Q)
        assert R.shape == (3, 3) and t.shape == (3,)
        assert rot_angle(R, g[name + "_R"]) < 1e-7, name
        np.testing.assert_allclose(t, g[name + "_t"], atol=1e-8, err_msg=name)
        assert abs(np.linalg.det(R) - 1) < 1e-12
    P, Q = cases.kabsch_batch()
    dev = torch.device("cuda")
    T, n = ops.kabsch_batch(torch.from_numpy(P).to(dev), torch.from_numpy(Q).to(dev))
    np.testing.assert_allclose(T.cpu().numpy(), g["batch_T"], atol=1e-11)
    assert int(n.min()) == P.shape[1] == int(n.max())
    # 0/1 weights == boolean compaction in the reference
    rng = np.random.default_rng(0)
    w = rng.random(P.shape[:2]) < 0.6
    Tw, nw = ops.kabsch_batch(torch.from_numpy(P).to(dev), torch.from_numpy(Q).to(dev),
                              torch.from_numpy(w.astype(np.uint8)).to(dev))
    Tw = Tw.cpu().numpy()
    for b in range(0, P.shape[0], 257):
        R, t = ref.kabsch(P[b][w[b]], Q[b][w[b]])
        np.testing.assert_allclose(Tw[b, :3, :3], R, atol=1e-11)
        np.testing.assert_allclose(Tw[b, :3, 3], t, atol=1e-11)
    assert np.array_equal(nw.cpu().numpy(), w.sum(1))
    # ragged batch sizes around the 4-pairs-per-workgroup packing
    for b in (1, 2, 3, 5):
        Tb, _ = ops.kabsch_batch(torch.from_numpy(P[:b]).to(dev), torch.from_numpy(Q[:b]).to(dev))
        np.testing.assert_allclose(Tb.cpu().numpy(), g["batch_T"][:b], atol=1e-11)
    # no points at all: n_used == 0 and a NaN translation (the reference's mean of an empty set is NaN)
    Tz, nz = ops.kabsch_batch(torch.from_numpy(P[:1]).to(dev), torch.from_numpy(Q[:1]).to(dev),
                              torch.zeros((1, P.shape[1]), dtype=torch.uint8, device=dev))
    assert int(nz[0]) == 0 and bool(torch.isnan(Tz[0, :3, 3]).all())


def test_rejects_cpu_tensors_and_bad_shapes(vo):
    from visual_odometry_amd import ops
    with pytest.raises(ValueError):
        ops.disparity_to_depth(torch.zeros(4), 1.0)
    dev = torch.device("cuda")
    with pytest.raises(ValueError):
        ops.kabsch_batch(torch.zeros((2, 5, 3), dtype=torch.float64, device=dev),
                         torch.zeros((2, 4, 3), dtype=torch.float64, device=dev))
    with pytest.raises(ValueError):
        ops.keypoints_to_3d(torch.zeros((5, 3), dtype=torch.float64, device=dev),
                            torch.zeros((1, 8, 8), device=dev), np.eye(3))


def test_ransac_matches_oracle_with_same_triples(vo):
    """Batched RANSAC against the NumPy restatement fed the SAME hypothesis triples:
    identical winner (inlier mask bit-exact) and refit pose to rounding."""
    from visual_odometry_amd import ops
    from visual_odometry_amd.geometry import HipCameraRansacXform
    rng = np.random.default_rng(17)
    B, N = 6, 300
    P = rng.uniform(-3, 3, (B, N, 3)); P[..., 2] = rng.uniform(0.5, 15, (B, N))
    Q = np.empty_like(P)
    truth = []
    for b in range(B):
        R = syn.random_rotation(rng, 0.1); t = rng.uniform(-0.2, 0.2, 3)
        Q[b] = P[b] @ R.T + t + rng.normal(0, 4e-3, (N, 3))
        bad = rng.choice(N, N // 4, replace=False)
        Q[b, bad] += rng.uniform(-1, 1, (len(bad), 3))
        truth.append((R, t))
    w = rng.random((B, N)) < 0.8
    dev = torch.device("cuda")
    Pd, Qd = torch.from_numpy(P).to(dev), torch.from_numpy(Q).to(dev)
    wd = torch.from_numpy(w.astype(np.uint8)).to(dev)
    for weights, wt in ((None, None), (w, wd)):
        T, n_inl, mask = ops.ransac_kabsch_batch(Pd, Qd, wt, 0.05, 200, seed=1234, want_mask=True)
        T, n_inl, mask = T.cpu().numpy(), n_inl.cpu().numpy(), mask.cpu().numpy()
        for b in range(B):
            sel = np.ones(N, bool) if weights is None else weights[b]
            Rr, tr = ref.ransac_kabsch(P[b][sel], Q[b][sel], 0.05, 200, choice=ref.counter_triples(1234, b))
            assert rot_angle(T[b, :3, :3], Rr) < 1e-9
            np.testing.assert_allclose(T[b, :3, 3], tr, atol=1e-9)
            assert rot_angle(T[b, :3, :3], truth[b][0]) < 5e-3          # outliers rejected
            assert mask[b].sum() == n_inl[b] and not mask[b][~sel].any()
            err = np.linalg.norm(P[b] @ T[b, :3, :3].T + T[b, :3, 3] - Q[b], axis=1)
            assert n_inl[b] > 0.6 * sel.sum() and np.median(err[mask[b]]) < 0.02
    # the triples are keyed on the pair's index in the SEQUENCE: items 2.. of the batch handed
    # over as a batch of their own with first_pair=2 (and 2 + 2^32, the index is taken modulo
    # 2^32) reproduce the same poses bit for bit, and match the oracle fed the global index
    T_all, n_all = ops.ransac_kabsch_batch(Pd, Qd, wd, 0.05, 200, seed=1234)
    for base in (2, 2 + 2 ** 32):
        T_tail, n_tail = ops.ransac_kabsch_batch(Pd[2:].contiguous(), Qd[2:].contiguous(), wd[2:].contiguous(),
                                                 0.05, 200, seed=1234, first_pair=base)
        assert torch.equal(T_tail, T_all[2:]) and torch.equal(n_tail, n_all[2:])
    T_off, _ = ops.ransac_kabsch_batch(Pd[:1], Qd[:1], None, 0.05, 200, seed=1234, first_pair=41)
    Rr, tr = ref.ransac_kabsch(P[0], Q[0], 0.05, 200, choice=ref.counter_triples(1234, 41))
    np.testing.assert_allclose(T_off[0, :3, :3].cpu().numpy(), Rr, atol=1e-9)
    np.testing.assert_allclose(T_off[0, :3, 3].cpu().numpy(), tr, atol=1e-9)
    with pytest.raises(ValueError):
        ops.ransac_kabsch_batch(Pd, Qd, None, 0.05, 10, first_pair=-1)
    # the drop-in class: same signature, same exceptions as the reference
    est = HipCameraRansacXform(threshold=0.05, iterations=200, seed=1234)
    R0, t0 = est.compute_camera_xform(P[0], Q[0])
    Rr, tr = ref.ransac_kabsch(P[0], Q[0], 0.05, 200, choice=ref.counter_triples(1234, 0))
    assert rot_angle(R0, Rr) < 1e-9 and np.abs(t0 - tr).max() < 1e-9
    with pytest.raises(ValueError):
        est.compute_camera_xform(P[0][:2], Q[0][:2])
    with pytest.raises(ValueError):
        est.compute_camera_xform(P[0], Q[0][:, :2])
    # no hypothesis reaches 3 inliers -> identity, like the reference's fallback
    Tn, nn = ops.ransac_kabsch_batch(Pd[:1], (Qd[:1] * 50).contiguous(), None, 1e-6, 50, seed=1)
    np.testing.assert_array_equal(Tn[0].cpu().numpy(), np.eye(4))
    # fewer than 3 weighted points -> flagged with -1
    few = torch.zeros((1, N), dtype=torch.uint8, device=dev); few[0, :2] = 1
    _, nf = ops.ransac_kabsch_batch(Pd[:1], Qd[:1], few, 0.05, 50, seed=1)
    assert int(nf[0]) == -1
    # large N (dense 1280x720 grid size) runs and agrees with Kabsch on clean data
    Pl = rng.uniform(-3, 3, (2, 19928, 3)); Rl = syn.random_rotation(rng, 0.05)
    Ql = Pl @ Rl.T + 0.1
    Tl, nl = ops.ransac_kabsch_batch(torch.from_numpy(Pl).to(dev), torch.from_numpy(Ql).to(dev), None, 0.05, 64, seed=3)
    assert int(nl.min()) == 19928 and rot_angle(Tl[0, :3, :3].cpu().numpy(), Rl) < 1e-9


def test_to_2d_drops_nonpositive_depth(vo):
    from visual_odometry_amd.geometry import HipKeypoints3DXform
    K = cases.K_for(640)
    xform = HipKeypoints3DXform(_Cam(K))
    rng = np.random.default_rng(3)
    pts = rng.uniform(-2, 2, (257, 3)); pts[:, 2] = rng.uniform(-1, 8, 257); pts[5, 2] = 0.0
    got = xform.to_2d(pts)
    want = ref.project_to_2d(pts, K)
    assert got.shape == want.shape and got.shape[0] == int((pts[:, 2] > 0).sum())
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-12)
    assert xform.to_2d(np.zeros((0, 3))).shape == (0,)
    assert xform.to_2d(np.array([[1.0, 1.0, -1.0]])).shape == (0,)


@pytest.mark.parametrize("k_seg,first", [(5, 0), (25, 0), (4, 3)])
def test_keypoint_tracks_match_reference(vo, k_seg, first):
    from visual_odometry_amd.geometry import HipKeypointTracker
    g = golden(f"g9_tracks_k{k_seg}_f{first}")
    kp, depth, flow, K = cases.track_case()
    tracker = HipKeypointTracker(_Cam(K), kp, k=k_seg)
    tracks, alive, reseeded = tracker.track_batch(torch.from_numpy(depth).cuda(), torch.from_numpy(flow).cuda(), first)
    tracks, alive, reseeded = tracks.cpu().numpy(), alive.cpu().numpy(), reseeded.cpu().numpy()
    assert np.array_equal(alive, g["alive"]) and np.array_equal(reseeded, g["reseeded"])
    np.testing.assert_allclose(tracks, g["tracks"], rtol=1e-12, atol=1e-10)
    lists, start_z = HipKeypointTracker.as_track_lists(tracks, alive, reseeded, 3)
    assert len(lists) == int(alive[3].sum()) and all(len(t) == 4 - max(i for i in range(4) if reseeded[i]) for t in lists)
    assert all(z > 0 for z in start_z)


def test_keypoint_tracks_dense_grid_and_sharding(vo):
    """A dense grid (more slots than lanes) at full resolution against the oracle, and the
    same sequence tracked as two k-aligned shards: segments are independent."""
    from visual_odometry_amd.geometry import HipKeypointTracker
    rng = np.random.default_rng(8)
    h, w, F, k = 240, 320, 13, 4
    K = syn.left_K(0.5)
    kp = ref.uniform_keypoints(np.ones((h, w), bool), 3000)
    depth = rng.uniform(0.5, 12.0, (F, h, w)).astype(np.float32)
    depth[:, 50:90, 100:180] = 0.0
    flow = rng.normal(0, 1.5, (F - 1, 2, h, w)).astype(np.float32)
    want = ref.track_keypoints(kp, depth, flow, K, np.linalg.inv(K), k=k)
    tracker = HipKeypointTracker(_Cam(K), kp, k=k)
    d, f = torch.from_numpy(depth).cuda(), torch.from_numpy(flow).cuda()
    got = [t.cpu().numpy() for t in tracker.track_batch(d, f)]
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-12, atol=1e-10)
    a = [t.cpu().numpy() for t in tracker.track_batch(d[:9], f[:8], 0)]
    b = [t.cpu().numpy() for t in tracker.track_batch(d[8:], f[8:], 8)]
    for i in range(3):
        assert np.array_equal(np.concatenate([a[i], b[i]]), got[i])
    # degenerate sizes
    one = tracker.track_batch(d[:1], f[:0])
    assert one[0].shape == (0, kp.shape[0], 3)
    with pytest.raises(ValueError):
        tracker.track_batch(d, f[:-2])
