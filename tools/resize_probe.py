#!/usr/bin/env python3
"""Time of vo_resize_bilinear on the inter-level exchange of RAFT-Stereo's GRU (64 channels, 60x80 -> 120x160)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops
for (N, C, h, w) in [(64, 64, 60, 80), (192, 64, 60, 80), (192, 128, 30, 40)]:
    x = torch.randn(N, C, h, w, device="cuda")
    for _ in range(3): y = ops.resize_bilinear(x, (2 * h, 2 * w))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): y = ops.resize_bilinear(x, (2 * h, 2 * w))
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 20 * 1e3
    print(f"N{N} C{C} {h}x{w} -> {2*h}x{2*w}: {ms:.3f} ms  {(x.numel() + y.numel()) * 4 / ms / 1e6:.0f} GB/s", flush=True)
