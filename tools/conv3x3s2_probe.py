#!/usr/bin/env python3
"""Streaming MFMA 3x3 stride-2 convolution against the framework's convolution."""
import ctypes, os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import _lib
lib = _lib.load()
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.vo_conv3x3s2_filter.argtypes = [vp, vp, ci, ci, vp]; lib.vo_conv3x3s2_filter.restype = ci
lib.vo_conv3x3s2.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]; lib.vo_conv3x3s2.restype = ci
shapes = [(1, 8, 5, 7, 64), (2, 16, 33, 70, 40), (2, 24, 9, 131, 100), (64, 64, 480, 640, 96), (64, 96, 240, 320, 128),
          (64, 64, 240, 320, 96), (64, 128, 120, 160, 128),
          (64, 64, 480, 640, 64), (64, 64, 480, 640, 32)]          # the two launches of 64 -> 96, separately
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for (N, C, H, W, K) in shapes:
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") / (9 * C) ** 0.5
    b = torch.randn(K, device="cuda")
    Wg = torch.zeros((K + 63) // 64 * 64 * C * 9, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    assert lib.vo_conv3x3s2_filter(w.data_ptr(), Wg.data_ptr(), K, C, s) == 0
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(N, K, Ho, Wo, device="cuda")
    def run():
        assert lib.vo_conv3x3s2(x.data_ptr(), Wg.data_ptr(), y.data_ptr(), None, b.data_ptr(), 1, N, C, H, W, K, s) == 0
    run()
    want = torch.relu(F.conv2d(x.double(), w.double(), b.double(), 2, 1))
    assert want.shape == y.shape, (want.shape, y.shape)
    scale = want.abs().max().item()
    e1 = (y.double() - want).abs().max().item() / scale
    t1, t0 = timeit(run), timeit(lambda: F.conv2d(x, w, None, 2, 1))
    gf = 2 * N * K * Ho * Wo * C * 9 / 1e9
    print(f"N{N:3d} {C:3d}->{K:3d} @{H}x{W}: err {e1:.1e} | ours {t1:7.3f} ms ({gf/t1:6.1f} TF/s)  stock conv alone {t0:7.3f} ms ({gf/t0:6.1f})  x{t0/t1:.2f}", flush=True)
