#!/usr/bin/env python3
"""One downsampling pair (3x3 stride-2 convolution + its 1x1 stride-2 shortcut in one pass) at a layer shape of the
networks, a few launches, with its time: the workload of tools/pmc_s2_shortcut.sh.
  s2_shortcut_probe.py N C H W K [pieces]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops
N, C, H, W, K = (int(v) for v in sys.argv[1:6])
pieces = sys.argv[6] if len(sys.argv) > 6 else "f16"
with_stats = (sys.argv[7] if len(sys.argv) > 7 else "stats") == "stats"
g = torch.Generator().manual_seed(0)
x = torch.relu(torch.randn(N, C, H, W, generator=g)).cuda()
A3 = ops.conv3x3_split_filter((torch.randn(K, C, 3, 3, generator=g) / (3 * C ** 0.5)).cuda(), pieces)
A1 = ops.conv1x1_split_filter((torch.randn(K, C, 1, 1, generator=g) / C ** 0.5).cuda(), pieces)
st = lambda: torch.zeros((N, K, 2), device="cuda", dtype=torch.float64)
bias = torch.zeros(K, device="cuda")
run = ((lambda: ops.conv3x3s2_shortcut_split(x, A3, A1, K, stats=st(), stats2=st())) if with_stats else
       (lambda: ops.conv3x3s2_shortcut_split(x, A3, A1, K, None, bias, True, None, None, bias, False, None)))
for _ in range(2):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
Ho, Wo = (H + 1) // 2, (W + 1) // 2
gb = 4 * (x.numel() + 2 * N * K * Ho * Wo) / 1e9
print(f"{pieces} {'plane sums' if with_stats else 'bias + ReLU'} ({N},{C},{H},{W})->{K}: {dt * 1e3:.3f} ms, {gb / dt / 1e3:.2f} TB/s of x + both outputs, "
      f"{2 * 3 * 10 * N * K * C * Ho * Wo / dt / 1e12:.0f} TFLOP/s issued")
