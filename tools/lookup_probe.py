"""RECORD of the round-2 measurement (row-major against tiled lookup): needs tools/tiled_volume.hip.txt built back into the
library and its Python wrappers restored; kept for its method, not runnable as it is."""
raise SystemExit(__doc__)
#!/usr/bin/env python3
"""Row-major against tiled 2-D correlation lookup at the bench's shape (for timing and rocprofv3 --pmc passes).
Usage: lookup_probe.py [pairs] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W, D = 60, 80, 256
f1, f2 = torch.randn(B, H, W, D, device="cuda"), torch.randn(B, H, W, D, device="cuda")
plain = ops.corr_pyramid_2d(ops.corr_volume_2d(f1, f2, split=True), 4)
tiled = ops.corr_pyramid_2d_tiled(ops.corr_volume_2d_tiled(f1, f2, split=True), (H, W), 4)
g = torch.Generator().manual_seed(1)
ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
coords = (torch.stack([xs, ys])[None] + torch.randn(B, 2, H, W, generator=g) * 4.0).cuda().contiguous()   # flows of a few pixels
out = torch.empty(B, 324, H, W, device="cuda")


def timeit(fn):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


t0 = timeit(lambda: ops.corr_lookup_2d(plain, coords, 4, out))
t1 = timeit(lambda: ops.corr_lookup_2d_tiled(tiled, (H, W), coords, 4, out))
alg = B * H * W * (2 + 4 * 81 + 4 * 100) * 4 / 1e9
print(f"{B} pairs: row-major {t0:.3f} ms ({alg / t0:.2f} TB/s algorithmic)  tiled {t1:.3f} ms ({alg / t1:.2f} TB/s)  x{t0 / t1:.2f}")
