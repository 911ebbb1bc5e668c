#!/usr/bin/env python3
"""Is the split 3x3 kernel's speed a function of the DATA?  The same launches on random operands, on operands with few
significant bits (small integers) and on zeros: the matrix cores' power draw depends on how many operand bits toggle,
and a kernel that runs at the chip's power limit speeds up when they do not (tools/mfma_f16_vs_bf16.hip measures the
bare instruction the same way)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops


def timeit(fn, n=60):
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


for (N, C, H, W, K) in [(16, 64, 480, 640, 64), (64, 256, 60, 80, 256)]:
    gf = 3 * 2 * N * K * H * W * C * 9 / 1e9
    for pieces in ("bf16", "f16"):
        row = []
        for what in ("random", "integers", "zero x", "zero x and w"):
            g = torch.Generator(device="cuda").manual_seed(1)
            if what == "random":
                x = torch.randn(N, C, H, W, device="cuda", generator=g); w = torch.randn(K, C, 3, 3, device="cuda", generator=g) / (3 * C ** 0.5)
            elif what == "integers":
                x = torch.randint(-3, 4, (N, C, H, W), device="cuda", generator=g).float(); w = torch.randint(-2, 3, (K, C, 3, 3), device="cuda", generator=g).float()
            elif what == "zero x":
                x = torch.zeros(N, C, H, W, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda", generator=g) / (3 * C ** 0.5)
            else:
                x = torch.zeros(N, C, H, W, device="cuda"); w = torch.zeros(K, C, 3, 3, device="cuda"); w[0, 0, 0, 0] = 1.0
            U = ops.conv3x3_split_filter(w, pieces)
            y = torch.empty(N, K, H, W, device="cuda")
            t = timeit(lambda: ops.conv3x3_winograd(x, U, K, out=y))
            row.append(f"{what} {t:6.3f} ms ({gf / t:5.0f} TF)")
        print(f"N{N} {C}->{K} @{H}x{W} {pieces:5s}: " + " | ".join(row), flush=True)
