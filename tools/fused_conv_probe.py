#!/usr/bin/env python3
"""Probe: MIOpen's fused conv+bias+ReLU (aten.miopen_convolution_relu) against the
bias-free convolution followed by the in-place bias+ReLU pass."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops

shapes = [  # N, C, H, W, K, kh, kw
    (8, 324, 60, 80, 256, 1, 1), (8, 256, 60, 80, 192, 3, 3), (8, 2, 60, 80, 128, 7, 7), (8, 128, 60, 80, 64, 3, 3),
    (8, 256, 60, 80, 126, 3, 3), (8, 128, 60, 80, 256, 3, 3),
    (8, 36, 120, 160, 64, 1, 1), (8, 64, 120, 160, 64, 3, 3), (8, 2, 120, 160, 64, 7, 7), (8, 128, 120, 160, 126, 3, 3),
    (8, 64, 120, 160, 256, 3, 3),
]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for (N, C, H, W, K, kh, kw) in shapes:
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, kh, kw, device="cuda") * 0.05
    b = torch.randn(K, device="cuda"); ones = torch.ones(K, device="cuda")
    pad = (kh // 2, kw // 2)
    def ours():
        y = F.conv2d(x, w, None, 1, pad)
        return ops.channel_affine_relu_(y, ones, b, True)
    def fused():
        return torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], list(pad), [1, 1], 1)
    def stock():
        return torch.relu_(F.conv2d(x, w, b, 1, pad))
    try:
        err = (ours() - fused()).abs().max().item()
        tf = timeit(fused)
    except Exception as e:
        err, tf = float("nan"), float("nan"); print("fused failed:", str(e)[:100])
    print(f"{C:4d}->{K:4d} {kh}x{kw} @{H}x{W}: ours {timeit(ours):7.1f} us  fused {tf:7.1f} us  stock {timeit(stock):7.1f} us  maxdiff {err:.2e}", flush=True)
