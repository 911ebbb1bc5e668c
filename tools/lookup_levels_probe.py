#!/usr/bin/env python3
"""Time of the 2-D correlation lookup with 1..4 pyramid levels (which level costs what)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W, D = 60, 80, 256
f1, f2 = torch.randn(B, H, W, D, device="cuda"), torch.randn(B, H, W, D, device="cuda")
pyr = ops.corr_pyramid_2d(ops.corr_volume_2d(f1, f2, split=True), 4)
g = torch.Generator().manual_seed(1)
ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
coords = (torch.stack([xs, ys])[None] + torch.randn(B, 2, H, W, generator=g) * 4.0).cuda().contiguous()


def timeit(fn, reps=5):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


prev = 0.0
for levels in (1, 2, 3, 4):
    out = torch.empty(B, 81 * levels, H, W, device="cuda")
    t = timeit(lambda: ops.corr_lookup_2d(pyr[:levels], coords, 4, out))
    print(f"{B} pairs, levels 0..{levels - 1}: {t:.3f} ms (+{t - prev:.3f} for level {levels - 1})", flush=True)
    prev = t
