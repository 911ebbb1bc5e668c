"""The allocation-free pose entry point (vo_pair_poses_static: workspace inside the library)
against vo_pair_poses (workspace from the caller), through the C ABI, bit for bit."""
import numpy as np
import pytest
import torch

from tests.golden import cases
from visual_odometry_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _inputs(pairs, n_kp, seed):
    from oracle import geometry_ref as ref
    rng = np.random.default_rng(seed)
    K = cases.K_for(640)
    scenes = [syn.plane_pair(rng, 480, 640, K, noise_px=0.1) for _ in range(pairs)]
    kp = ref.uniform_keypoints(np.ones((480, 640), bool), n_kp).astype(np.float64)
    dev = torch.device("cuda")
    st = lambda k: torch.from_numpy(np.stack([s[k] for s in scenes])).to(dev)
    return torch.from_numpy(kp).to(dev), st("depth1"), st("depth2"), st("flow"), np.linalg.inv(K)


def _both(kp, d1, d2, fl, K_inv, min_depth=0.0, max_depth=15.0):
    """(T, skipped) from the caller-workspace entry point and from the static one."""
    from visual_odometry_amd import _lib, ops
    B, N = d1.shape[0], kp.shape[0]
    T_ws, sk_ws, _ = ops.pair_poses(kp, d1, d2, fl, K_inv, min_depth, max_depth, 1.0, syn.CAMERA_OFFSET, 4)
    T = torch.full((B, 4, 4), float("nan"), dtype=torch.float64, device=d1.device)
    sk = torch.full((B,), 7, dtype=torch.uint8, device=d1.device)
    off = ops._host_doubles(syn.CAMERA_OFFSET, 3)
    _lib.call("vo_pair_poses_static", kp.data_ptr(), 0, d1.data_ptr(), d2.data_ptr(), fl.data_ptr(), ops._host9(K_inv),
              float(min_depth), float(max_depth), 1.0, off, 4, T.data_ptr(), sk.data_ptr(), B, N, 480, 640,
              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return T_ws, sk_ws, T, sk.bool()


def test_static_workspace_entry_point_matches_the_caller_workspace_one(vo):
    from visual_odometry_amd import _lib
    # first call: 6 pairs x 300 keypoints; second call right after it: fewer pairs, fewer
    # keypoints and a depth range that skips everything for some -- the shared array still holds
    # the first call's correspondences and counters
    a = _inputs(6, 320, seed=3)
    T_ws, sk_ws, T, sk = _both(*a)
    assert torch.equal(sk, sk_ws) and not bool(sk.any())
    assert torch.equal(T, T_ws)                                   # same kernels, same order: bit for bit
    b = _inputs(2, 200, seed=4)
    T_ws, sk_ws, T, sk = _both(*b)
    assert torch.equal(sk, sk_ws) and torch.equal(T, T_ws)
    T_ws, sk_ws, T, sk = _both(*b, min_depth=100.0, max_depth=200.0)
    assert bool(sk.all()) and torch.equal(sk, sk_ws)              # counters were cleared, not inherited
    # beyond the fixed workspace: a clear refusal, nothing launched
    kp, d1, d2, fl, K_inv = a
    from visual_odometry_amd import ops
    big_n = 262144 // 6 + 1
    with pytest.raises(_lib.VoHipError, match="exceeds the fixed workspace"):
        _lib.call("vo_pair_poses_static", kp.data_ptr(), 0, d1.data_ptr(), d2.data_ptr(), fl.data_ptr(), ops._host9(K_inv),
                  0.0, 15.0, 1.0, None, 4, T.data_ptr(), sk.data_ptr(), 6, big_n, 480, 640, 0)


def test_static_entry_point_allocates_nothing_and_replays_from_a_hip_graph(vo):
    """What the entry point exists for: after one warm-up call on the device (the lookup of the array's address
    happens there, outside any capture) a call neither takes nor returns device memory, and it can be recorded
    into a HIP graph whose replays give the poses of a direct call."""
    from visual_odometry_amd import _lib, ops
    kp, d1, d2, fl, K_inv = _inputs(4, 320, seed=5)
    B, N = d1.shape[0], kp.shape[0]
    off, k9 = ops._host_doubles(syn.CAMERA_OFFSET, 3), ops._host9(K_inv)
    T = torch.zeros((B, 4, 4), dtype=torch.float64, device=d1.device)
    sk = torch.zeros((B,), dtype=torch.uint8, device=d1.device)

    def call(stream):
        _lib.call("vo_pair_poses_static", kp.data_ptr(), 0, d1.data_ptr(), d2.data_ptr(), fl.data_ptr(), k9, 0.0, 15.0, 1.0, off, 4,
                  T.data_ptr(), sk.data_ptr(), B, N, 480, 640, stream)

    call(torch.cuda.current_stream().cuda_stream)                      # warm-up: the address look-up of this device
    torch.cuda.synchronize()
    want_T, want_sk = T.clone(), sk.clone()
    free0, _ = torch.cuda.mem_get_info()
    reserved0 = torch.cuda.memory_reserved()
    for _ in range(3):
        call(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0 and torch.cuda.memory_reserved() == reserved0      # nothing allocated, nothing freed
    assert torch.equal(T, want_T) and torch.equal(sk, want_sk)
    # under capture: a stream that is capturing refuses allocation, so a recorded call proves there is none
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            call(side.cuda_stream)
    T.zero_(); sk.fill_(9)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(T, want_T) and torch.equal(sk, want_sk)
    # a replay on new contents of the same buffers
    kp2, e1, e2, fl2, _ = _inputs(4, 320, seed=6)
    d1.copy_(e1); d2.copy_(e2); fl.copy_(fl2)
    g.replay()
    torch.cuda.synchronize()
    got_T = T.clone()
    call(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(got_T, T) and not torch.equal(got_T, want_T)
