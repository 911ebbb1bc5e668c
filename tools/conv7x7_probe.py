#!/usr/bin/env python3
"""Streaming MFMA 7x7 convolution (few input channels) against the framework's convolution."""
import ctypes, os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import _lib
lib = _lib.load()
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.vo_conv7x7_filter.argtypes = [vp, vp, ci, ci, vp]; lib.vo_conv7x7_filter.restype = ci
lib.vo_conv7x7.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]; lib.vo_conv7x7.restype = ci
shapes = [(1, 3, 5, 7, 64), (2, 2, 33, 70, 40), (2, 3, 9, 130, 100), (64, 3, 480, 640, 64), (32, 3, 480, 640, 64),
          (32, 2, 120, 160, 64), (32, 2, 60, 80, 128)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for (N, C, H, W, K) in shapes:
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, 7, 7, device="cuda") / (49 * C) ** 0.5
    b = torch.randn(K, device="cuda")
    Wg = torch.zeros((K + 63) // 64 * 64 * C * 56, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    assert lib.vo_conv7x7_filter(w.data_ptr(), Wg.data_ptr(), K, C, s) == 0
    y = torch.empty(N, K, H, W, device="cuda")
    def run():
        assert lib.vo_conv7x7(x.data_ptr(), Wg.data_ptr(), y.data_ptr(), None, b.data_ptr(), 1, N, C, H, W, K, 1, s) == 0
    run()
    want = torch.relu(F.conv2d(x.double(), w.double(), b.double(), 1, 3))
    scale = want.abs().max().item()
    e1 = (y.double() - want).abs().max().item() / scale
    e0 = (torch.relu(F.conv2d(x, w, b, 1, 3)).double() - want).abs().max().item() / scale
    t1, t0 = timeit(run), timeit(lambda: torch.relu_(F.conv2d(x, w, b, 1, 3)))
    gf = 2 * N * K * H * W * C * 49 / 1e9
    print(f"N{N:3d} {C}->{K:3d} @{H}x{W}: err ours {e1:.1e} stock {e0:.1e} | ours {t1:7.3f} ms ({gf/t1:6.1f} TF/s)  stock conv+bias+relu {t0:7.3f} ms ({gf/t0:6.1f})  x{t0/t1:.2f}", flush=True)
