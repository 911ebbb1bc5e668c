#!/usr/bin/env python3
"""F(4x4,3x3) Winograd kernel against the F(2x2,3x3) one and MIOpen: error and time."""
import ctypes, os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops, _lib

lib = _lib.load()
vp, ci, f32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
lib.vo_winograd4_filter.argtypes = [vp, vp, ci, ci, vp]; lib.vo_winograd4_filter.restype = ci
lib.vo_conv3x3_winograd4.argtypes = [vp, vp, vp, i64, vp, vp, ci, vp, vp, vp, f32, ci, ci, ci, ci, ci, vp]
lib.vo_conv3x3_winograd4.restype = ci

shapes = [  # N, C, H, W, K
    (2, 8, 16, 32, 64), (1, 16, 37, 53, 40), (2, 64, 60, 80, 126),
    (16, 64, 480, 640, 64), (16, 96, 240, 320, 96), (16, 128, 120, 160, 128),
    (16, 256, 60, 80, 192), (16, 256, 120, 160, 128), (16, 64, 120, 160, 256), (16, 128, 60, 80, 256),
    (32, 128, 60, 80, 128),
]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for (N, C, H, W, K) in shapes:
    x = torch.randn(N, C, H, W, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * (1.0 / (3 * C ** 0.5))
    U = ops.winograd_filter(w)
    Kpad = (K + 63) // 64 * 64
    U4 = torch.zeros(Kpad * C * 36, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    rc = lib.vo_winograd4_filter(w.data_ptr(), U4.data_ptr(), K, C, s); assert rc == 0, _lib.last_error()
    y = torch.empty(N, K, H, W, device="cuda")
    def run4():
        rc = lib.vo_conv3x3_winograd4(x.data_ptr(), U4.data_ptr(), y.data_ptr(), 0, None, None, 0, None, None, None, 0.0,
                                      N, C, H, W, K, s)
        assert rc == 0
    run4()
    want = F.conv2d(x.double(), w.double(), None, 1, 1)
    scale = want.abs().max().item()
    e4 = (y.double() - want).abs().max().item() / scale
    e2 = (ops.conv3x3_winograd(x, U, K).double() - want).abs().max().item() / scale
    e0 = (F.conv2d(x, w, None, 1, 1).double() - want).abs().max().item() / scale
    t2, t4 = timeit(lambda: ops.conv3x3_winograd(x, U, K)), timeit(run4)
    gf = 2 * N * K * H * W * C * 9 / 1e9
    if N >= 16:                                                  # the fused epilogues: bias + ReLU, and the residual join on top
        b = torch.randn(K, device="cuda"); sk = torch.randn(N, K, H, W, device="cuda")
        def run_b():
            assert lib.vo_conv3x3_winograd4(x.data_ptr(), U4.data_ptr(), y.data_ptr(), 0, None, b.data_ptr(), 1, None, None, None,
                                            0.0, N, C, H, W, K, s) == 0
        def run_s():
            assert lib.vo_conv3x3_winograd4(x.data_ptr(), U4.data_ptr(), y.data_ptr(), 0, None, b.data_ptr(), 1, sk.data_ptr(), None,
                                            None, 0.0, N, C, H, W, K, s) == 0
        run_s()
        ref = torch.relu(sk + torch.relu(F.conv2d(x, w, b, 1, 1)))
        es = (y - ref).abs().max().item() / scale
        print(f"      bias+relu {timeit(run_b):7.3f} ms   bias+relu+skip {timeit(run_s):7.3f} ms (err {es:.1e})", flush=True)
    print(f"N{N:3d} {C:3d}->{K:3d} @{H}x{W}: err f4 {e4:.1e} f2 {e2:.1e} miopen {e0:.1e} | f2 {t2:7.3f} ms ({gf/t2:6.1f})  f4 {t4:7.3f} ms ({gf/t4:6.1f})  x{t2/t4:.2f}", flush=True)
