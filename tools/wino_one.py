#!/usr/bin/env python3
"""A few launches of a Winograd kernel at one shape (for rocprofv3 --pmc passes).
Usage: wino_one.py N C H W K [reps] [tile]   tile 2 / 4: the 3x3 kernels; 5: the 1x5 GRU gate kernel;
32: the streaming 3x3 stride-2 kernel; 1: the split-bf16 direct 3x3 kernel"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops
N, C, H, W, K = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
tile = int(sys.argv[7]) if len(sys.argv) > 7 else 4
x = torch.randn(N, C, H, W, device="cuda")
if tile == 5:
    U = ops.winograd1d_filter(torch.randn(K, C, 1, 5, device="cuda") * 0.05)
    for _ in range(reps):
        y = ops.conv5_winograd(x, U, K, False)
elif tile == 32:
    Wg = ops.conv3x3s2_filter(torch.randn(K, C, 3, 3, device="cuda") * 0.05)
    for _ in range(reps):
        y = ops.conv3x3s2(x, Wg, K)
elif tile == 1:
    U = ops.conv3x3_split_filter(torch.randn(K, C, 3, 3, device="cuda") * 0.05)
    for _ in range(reps):
        y = ops.conv3x3_winograd(x, U, K)
else:
    U = ops.winograd_filter(torch.randn(K, C, 3, 3, device="cuda") * 0.05, tile)
    for _ in range(reps):
        y = ops.conv3x3_winograd(x, U, K)
torch.cuda.synchronize()
print("done")
