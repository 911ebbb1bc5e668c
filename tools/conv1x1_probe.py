#!/usr/bin/env python3
"""Streaming MFMA 1x1 convolution against the framework's convolution: error and time."""
import ctypes, os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import _lib
lib = _lib.load()
vp, ci, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib.vo_conv1x1_filter.argtypes = [vp, vp, ci, ci, vp]; lib.vo_conv1x1_filter.restype = ci
lib.vo_conv1x1.argtypes = [vp, vp, vp, i64, vp, vp, ci, ci, ci, i64, ci, vp]; lib.vo_conv1x1.restype = ci
shapes = [(1, 8, 5, 7, 64), (2, 13, 33, 21, 40), (2, 36, 30, 41, 18), (32, 256, 120, 160, 18), (32, 324, 60, 80, 256),
          (32, 36, 120, 160, 64), (64, 64, 240, 320, 96), (64, 128, 120, 160, 256), (32, 256, 120, 160, 144), (32, 256, 60, 80, 576)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for (N, C, H, W, K) in shapes:
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, 1, 1, device="cuda") / C ** 0.5
    b = torch.randn(K, device="cuda")
    Wg = torch.zeros((K + 63) // 64 * 64 * ((C + 7) // 8 * 8), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    assert lib.vo_conv1x1_filter(w.data_ptr(), Wg.data_ptr(), K, C, s) == 0
    y = torch.empty(N, K, H, W, device="cuda")
    def run():
        assert lib.vo_conv1x1(x.data_ptr(), Wg.data_ptr(), y.data_ptr(), 0, None, b.data_ptr(), 1, N, C, H * W, K, s) == 0
    run()
    want = torch.relu(F.conv2d(x.double(), w.double(), b.double()))
    scale = want.abs().max().item()
    e1 = (y.double() - want).abs().max().item() / scale
    e0 = (torch.relu(F.conv2d(x, w, b)).double() - want).abs().max().item() / scale
    t1, t0 = timeit(run), timeit(lambda: torch.relu_(F.conv2d(x, w, b)))
    gf = 2 * N * K * H * W * C / 1e9; gb = 4 * N * H * W * (C + K) / 1e9
    print(f"N{N:3d} {C:3d}->{K:3d} @{H}x{W}: err ours {e1:.1e} stock {e0:.1e} | ours {t1:7.3f} ms ({gf/t1:6.1f} TF/s, {gb/t1*1e3:6.0f} GB/s)  stock conv+bias+relu {t0:7.3f} ms  x{t0/t1:.2f}", flush=True)
