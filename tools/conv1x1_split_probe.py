#!/usr/bin/env python3
"""1x1 convolution: the float32 streaming kernel against the split-bf16 template (time, error)."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops


def timeit(fn, n=6):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for (N, C, H, W, K) in [(192, 336, 60, 80, 256), (192, 256, 60, 80, 576), (192, 256, 120, 160, 144), (192, 128, 120, 160, 256),
                        (192, 48, 120, 160, 64), (192, 256, 60, 80, 18)]:
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, 1, 1, device="cuda") / C ** 0.5
    b = torch.randn(K, device="cuda")
    Wg, A = ops.conv1x1_filter(w), ops.conv1x1_split_filter(w)
    want = torch.relu(F.conv2d(x[:2].double(), w.double(), b.double()))
    scale = want.abs().max().item()
    e0 = (ops.conv1x1(x[:2].contiguous(), Wg, K, None, b, True).double() - want).abs().max().item() / scale
    e1 = (ops.conv1x1_split(x[:2].contiguous(), A, K, None, b, True).double() - want).abs().max().item() / scale
    t0 = timeit(lambda: ops.conv1x1(x, Wg, K, None, b, True))
    t1 = timeit(lambda: ops.conv1x1_split(x, A, K, None, b, True))
    print(f"N{N} {C}->{K} @{H}x{W}: err f32 {e0:.1e} split {e1:.1e} | f32 {t0:7.3f} ms  split {t1:7.3f} ms  x{t0 / t1:.2f}", flush=True)
