#!/usr/bin/env python3
"""Correctness and speed of the Winograd/MFMA 3x3 convolution against MIOpen (F.conv2d)."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops

shapes = [  # N, C, H, W, K
    (2, 8, 16, 16, 64), (1, 16, 37, 53, 40), (2, 64, 60, 80, 126),
    (16, 64, 480, 640, 64), (32, 64, 480, 640, 64), (16, 96, 240, 320, 96), (16, 128, 120, 160, 128),
    (17, 64, 240, 320, 64), (16, 256, 60, 80, 192), (16, 256, 60, 80, 128), (16, 128, 60, 80, 256),
    (16, 128, 120, 160, 128), (16, 64, 120, 160, 256), (16, 256, 120, 160, 128), (16, 64, 120, 160, 64),
]
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for (N, C, H, W, K) in shapes:
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * (1.0 / (3 * C ** 0.5))
    U = ops.winograd_filter(w)
    got = ops.conv3x3_winograd(x, U, K)
    want = F.conv2d(x, w, None, 1, 1)
    err = (got - want).abs().max().item() / want.abs().max().item()
    t1, t2 = timeit(lambda: ops.conv3x3_winograd(x, U, K)), timeit(lambda: F.conv2d(x, w, None, 1, 1))
    gf = 2 * N * K * H * W * C * 9 / 1e9
    print(f"N{N:3d} {C:3d}->{K:3d} @{H}x{W}: rel err {err:.1e}  ours {t1:7.3f} ms ({gf/t1:6.1f} TF/s eff)  miopen {t2:7.3f} ms ({gf/t2:6.1f})  x{t2/t1:.2f}", flush=True)
