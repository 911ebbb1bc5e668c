#!/usr/bin/env python3
"""A few launches of the Winograd kernel at one shape (for rocprofv3 --pmc passes).
Usage: wino_one.py N C H W K [reps] [tile]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops
N, C, H, W, K = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05
tile = int(sys.argv[7]) if len(sys.argv) > 7 else 4
U = ops.winograd_filter(w, tile)
for _ in range(reps):
    y = ops.conv3x3_winograd(x, U, K)
torch.cuda.synchronize()
print("done")
