#!/usr/bin/env python3
"""Split-bf16 direct 3x3 kernel against the F(4x4,3x3) Winograd kernel: error (vs float64) and time."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops

shapes = [  # N, C, H, W, K
    (2, 16, 16, 32, 64), (1, 16, 37, 54, 40), (2, 64, 60, 80, 126),
    (16, 64, 480, 640, 64), (16, 96, 240, 320, 96), (16, 128, 120, 160, 128),
    (16, 256, 60, 80, 192), (16, 256, 120, 160, 128), (16, 64, 120, 160, 256), (16, 128, 60, 80, 256),
    (32, 128, 60, 80, 128), (64, 64, 240, 320, 64),
]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for (N, C, H, W, K) in shapes:
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * (1.0 / (3 * C ** 0.5))
    U4 = ops.winograd_filter(w, 4); Us = ops.conv3x3_split_filter(w)
    want = F.conv2d(x.double(), w.double(), None, 1, 1)
    scale = want.abs().max().item()
    ys = ops.conv3x3_winograd(x, Us, K)
    e4 = (ops.conv3x3_winograd(x, U4, K).double() - want).abs().max().item() / scale
    es = (ys.double() - want).abs().max().item() / scale
    t4, ts = timeit(lambda: ops.conv3x3_winograd(x, U4, K)), timeit(lambda: ops.conv3x3_winograd(x, Us, K))
    gf = 2 * N * K * H * W * C * 9 / 1e9
    line = f"N{N:3d} {C:3d}->{K:3d} @{H}x{W}: err wino4 {e4:.1e} split {es:.1e} | wino4 {t4:7.3f} ms ({gf/t4:6.1f})  split {ts:7.3f} ms ({gf/ts:6.1f})  x{t4/ts:.2f}"
    if N >= 16:
        b = torch.randn(K, device="cuda"); sk = torch.randn(N, K, H, W, device="cuda")
        ref = torch.relu(sk + torch.relu(F.conv2d(x, w, b, 1, 1)))
        got = ops.conv3x3_winograd(x, Us, K, None, b, True, sk)
        eb = (got - ref).abs().max().item() / scale
        tb = timeit(lambda: ops.conv3x3_winograd(x, Us, K, None, b, True, sk))
        st = torch.zeros((N, K, 2), device="cuda", dtype=torch.float64)
        yr = ops.conv3x3_winograd(x, Us, K, stats=st)
        e_st = max((st[..., 0] - yr.double().sum((2, 3))).abs().max().item() / (H * W) ** 0.5,
                   (st[..., 1] - yr.double().pow(2).sum((2, 3))).abs().max().item() / (H * W))
        tst = timeit(lambda: ops.conv3x3_winograd(x, Us, K, stats=st))
        line += f" | +bias,relu,skip {tb:7.3f} ms err {eb:.1e} | stats {tst:7.3f} ms err {e_st:.1e}"
    print(line, flush=True)

print("1x5 / 5x1 GRU gate convolutions and 3x3 stride-2 layers: float32-instruction kernel vs split-bf16")
for (N, C, H, W, K) in [(64, 384, 60, 80, 256), (64, 384, 60, 80, 128)]:
    for vertical in (False, True):
        x = torch.randn(N, C, H, W, device="cuda")
        w = torch.randn(*((K, C, 5, 1) if vertical else (K, C, 1, 5)), device="cuda") / (5 * C) ** 0.5
        U1, Us = ops.winograd1d_filter(w), ops.conv5_split_filter(w)
        want = F.conv2d(x.double(), w.double(), None, 1, (2, 0) if vertical else (0, 2)); scale = want.abs().max().item()
        e1 = (ops.conv5_winograd(x, U1, K, vertical).double() - want).abs().max().item() / scale
        es = (ops.conv5_split(x, Us, K, vertical).double() - want).abs().max().item() / scale
        t1, ts = timeit(lambda: ops.conv5_winograd(x, U1, K, vertical)), timeit(lambda: ops.conv5_split(x, Us, K, vertical))
        gf = 2 * N * K * H * W * C * 5 / 1e9
        print(f"N{N:3d} {C:3d}->{K:3d} @{H}x{W} {'5x1' if vertical else '1x5'}: err wino {e1:.1e} split {es:.1e} | wino {t1:7.3f} ms ({gf/t1:6.1f})  split {ts:7.3f} ms ({gf/ts:6.1f})  x{t1/ts:.2f}", flush=True)
for (N, C, H, W, K) in [(64, 64, 480, 640, 96), (64, 96, 240, 320, 128), (64, 128, 120, 160, 128), (64, 64, 240, 320, 96)]:
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") / (3 * C ** 0.5)
    Wg, Us = ops.conv3x3s2_filter(w), ops.conv3x3_split_filter(w)
    want = F.conv2d(x[:2].double(), w.double(), None, 2, 1); scale = want.abs().max().item()
    e1 = (ops.conv3x3s2(x[:2].contiguous(), Wg, K).double() - want).abs().max().item() / scale
    es = (ops.conv3x3s2_split(x[:2].contiguous(), Us, K).double() - want).abs().max().item() / scale
    t1, ts = timeit(lambda: ops.conv3x3s2(x, Wg, K)), timeit(lambda: ops.conv3x3s2_split(x, Us, K))
    gf = 2 * N * K * (H // 2) * (W // 2) * C * 9 / 1e9
    print(f"N{N:3d} {C:3d}->{K:3d} @{H}x{W} s2: err f32 {e1:.1e} split {es:.1e} | f32 {t1:7.3f} ms ({gf/t1:6.1f})  split {ts:7.3f} ms ({gf/ts:6.1f})  x{t1/ts:.2f}", flush=True)
