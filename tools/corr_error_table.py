#!/usr/bin/env python3
"""Largest |error| / largest |entry| of the correlation volumes against float64, per arithmetic."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops
def err(g, w): return (g.double() - w).abs().max().item() / w.abs().max().item()
for (B, H, W, D, sc) in [(2, 60, 80, 256, 1.0), (2, 60, 80, 256, 0.02), (1, 90, 160, 256, 5.0)]:
    g = torch.Generator().manual_seed(1)
    f1 = (torch.randn(B, H, W, D, generator=g) * sc).cuda(); f2 = (torch.randn(B, H, W, D, generator=g) * sc).cuda()
    want = torch.einsum("bid,bjd->bij", f1.view(B, H * W, D).double(), f2.view(B, H * W, D).double()) / D ** 0.5
    r = {k: err(ops.corr_volume_2d(f1, f2, split=s).view(B, H * W, H * W), want) for k, s in (("f32", False), ("bf16x3", True), ("f16x3", "f16"))}
    print(f"2-D ({B},{H},{W},{D}) scale {sc}: " + "  ".join(f"{k} {v:.2e}" for k, v in r.items()), flush=True)
    del want
for (B, H, W, D) in [(2, 120, 160, 256)]:
    g = torch.Generator().manual_seed(2)
    f1 = torch.randn(B, H, W, D, generator=g).cuda(); f2 = torch.randn(B, H, W, D, generator=g).cuda()
    want = torch.einsum("bhid,bhjd->bhij", f1.double(), f2.double()) / D ** 0.5
    r = {k: err(ops.corr_volume_1d(f1, f2, split=s), want) for k, s in (("f32", False), ("bf16x3", True), ("f16x3", "f16"))}
    print(f"1-D ({B},{H},{W},{D}): " + "  ".join(f"{k} {v:.2e}" for k, v in r.items()), flush=True)
