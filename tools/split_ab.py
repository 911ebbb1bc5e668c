#!/usr/bin/env python3
"""Times of the split-bf16 convolutions on the step's large shapes (one line per shape), for A/B runs of two builds."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

out = []
for (N, C, H, W, K) in [(16, 64, 480, 640, 64), (16, 96, 240, 320, 96), (16, 128, 120, 160, 128), (64, 128, 60, 80, 128)]:
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") / (3 * C ** 0.5)
    Us = ops.conv3x3_split_filter(w)
    t = timeit(lambda: ops.conv3x3_winograd(x, Us, K))
    out.append(f"{C}->{K}@{H}: {t:.3f} ms {2 * N * K * H * W * C * 27 / t / 1e9:.0f}")
    del x
for (N, C, H, W, K) in [(64, 384, 60, 80, 256)]:
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, 1, 5, device="cuda") / (5 * C) ** 0.5
    Us = ops.conv5_split_filter(w)
    t = timeit(lambda: ops.conv5_split(x, Us, K, False))
    out.append(f"1x5 {C}->{K}: {t:.3f} ms {2 * N * K * H * W * C * 15 / t / 1e9:.0f}")
print(" | ".join(out), flush=True)
