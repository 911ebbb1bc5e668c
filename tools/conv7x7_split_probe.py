#!/usr/bin/env python3
"""One- and two-channel 7x7 convolution: the split-bf16 kernel against the float32 streaming kernel, and
the 1x1 stride-2 projection shortcut on the split-bf16 template against the float32 kernel."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops


def timeit(fn, n=8):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for (N, C, H, W, K, st) in [(128, 1, 480, 640, 64, 1), (64, 1, 480, 640, 64, 1), (130, 1, 480, 640, 64, 2), (64, 2, 120, 160, 64, 1),
                            (64, 2, 60, 80, 128, 1)]:
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, 7, 7, device="cuda") / (49 * C) ** 0.5
    Wg, A = ops.conv7x7_filter(w), ops.conv7x7_split_filter(w)
    want = F.conv2d(x[:2].double(), w.double(), None, st, 3)
    scale = want.abs().max().item()
    e0 = (ops.conv7x7(x[:2].contiguous(), Wg, K, stride=st).double() - want).abs().max().item() / scale
    e1 = (ops.conv7x7_split(x[:2].contiguous(), A, K, stride=st).double() - want).abs().max().item() / scale
    t0 = timeit(lambda: ops.conv7x7(x, Wg, K, stride=st))
    t1 = timeit(lambda: ops.conv7x7_split(x, A, K, stride=st))
    stats = torch.zeros(N, K, 2, device="cuda", dtype=torch.float64)
    t2 = timeit(lambda: ops.conv7x7_split(x, A, K, stride=st, stats=stats))
    gb = 4 * (x.numel() + N * K * ((H - 1) // st + 1) * ((W - 1) // st + 1)) / 1e9
    print(f"7x7 N{N:3d} {C}->{K:3d} @{H}x{W} s{st}: err f32 {e0:.1e} split {e1:.1e} | f32 {t0:7.3f} ms  split {t1:7.3f} ms "
          f"({gb / t1:.2f} TB/s)  +stats {t2:7.3f} ms  x{t0 / t1:.2f}", flush=True)
for (N, C, H, W, K) in [(128, 64, 480, 640, 96), (128, 96, 240, 320, 128), (65, 64, 240, 320, 96)]:
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, 1, 1, device="cuda") / C ** 0.5
    Wg, A = ops.conv1x1_filter(w), ops.conv1x1_split_filter(w)
    want = F.conv2d(x[:2].double(), w.double(), None, 2, 0)
    scale = want.abs().max().item()
    e0 = (ops.conv1x1s2(x[:2].contiguous(), Wg, K).double() - want).abs().max().item() / scale
    e1 = (ops.conv1x1s2_split(x[:2].contiguous(), A, K).double() - want).abs().max().item() / scale
    t0 = timeit(lambda: ops.conv1x1s2(x, Wg, K))
    t1 = timeit(lambda: ops.conv1x1s2_split(x, A, K))
    stats = torch.zeros(N, K, 2, device="cuda", dtype=torch.float64)
    t2 = timeit(lambda: ops.conv1x1s2_split(x, A, K, stats=stats))
    print(f"1x1s2 N{N:3d} {C}->{K:3d} @{H}x{W}: err f32 {e0:.1e} split {e1:.1e} | f32 {t0:7.3f} ms  split {t1:7.3f} ms  "
          f"+stats {t2:7.3f} ms  x{t0 / t1:.2f}", flush=True)
