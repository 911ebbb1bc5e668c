#!/usr/bin/env python3
"""1-D Winograd F(4,5) kernel for 1x5 / 5x1 convolutions against MIOpen: error and time."""
import ctypes, os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import _lib
lib = _lib.load()
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.vo_winograd1d_filter.argtypes = [vp, vp, ci, ci, vp]; lib.vo_winograd1d_filter.restype = ci
lib.vo_conv5_winograd.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]; lib.vo_conv5_winograd.restype = ci
shapes = [(1, 8, 5, 7, 64), (2, 16, 33, 21, 40), (2, 64, 60, 80, 126), (32, 384, 60, 80, 256), (32, 384, 60, 80, 128),
          (32, 256, 120, 160, 128)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for (N, C, H, W, K) in shapes:
    for vertical in (0, 1):
        x = torch.randn(N, C, H, W, device="cuda")
        w = torch.randn(K, C, 5, device="cuda") / (5 * C) ** 0.5
        w4 = w.view(K, C, 5, 1) if vertical else w.view(K, C, 1, 5)
        pad = (2, 0) if vertical else (0, 2)
        U = torch.zeros((K + 63) // 64 * 64 * C * 8, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        assert lib.vo_winograd1d_filter(w.data_ptr(), U.data_ptr(), K, C, s) == 0
        y = torch.empty(N, K, H, W, device="cuda")
        def run():
            assert lib.vo_conv5_winograd(x.data_ptr(), U.data_ptr(), y.data_ptr(), N, C, H, W, K, vertical, s) == 0
        run()
        want = F.conv2d(x.double(), w4.double(), None, 1, pad)
        scale = want.abs().max().item()
        e1 = (y.double() - want).abs().max().item() / scale
        e0 = (F.conv2d(x, w4, None, 1, pad).double() - want).abs().max().item() / scale
        t1, t0 = timeit(run), timeit(lambda: F.conv2d(x, w4, None, 1, pad))
        gf = 2 * N * K * H * W * C * 5 / 1e9
        print(f"N{N:3d} {C:3d}->{K:3d} @{H}x{W} {'5x1' if vertical else '1x5'}: err ours {e1:.1e} miopen {e0:.1e} | ours {t1:7.3f} ms ({gf/t1:6.1f} TF/s eff)  miopen {t0:7.3f} ms ({gf/t0:6.1f})  x{t0/t1:.2f}", flush=True)
