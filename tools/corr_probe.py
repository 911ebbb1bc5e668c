#!/usr/bin/env python3
"""Correlation volume: float32 matrix instruction against three bf16 terms per product (time, error)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops


def timeit(fn, n=5):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for B in (16, 64):
    f1, f2 = torch.randn(B, 60, 80, 256, device="cuda"), torch.randn(B, 60, 80, 256, device="cuda")
    want = (f1[0].view(4800, 256).double() @ f2[0].view(4800, 256).double().T / 16.0)
    e0 = (ops.corr_volume_2d(f1[:1], f2[:1]).view(4800, 4800).double() - want).abs().max().item() / want.abs().max().item()
    e1 = (ops.corr_volume_2d(f1[:1], f2[:1], split=True).view(4800, 4800).double() - want).abs().max().item() / want.abs().max().item()
    t0, t1 = timeit(lambda: ops.corr_volume_2d(f1, f2)), timeit(lambda: ops.corr_volume_2d(f1, f2, split=True))
    gb = B * 4800 * 4800 * 4 / 1e9
    print(f"2-D B{B}: err f32 {e0:.1e} split {e1:.1e} | f32 {t0:.3f} ms  split {t1:.3f} ms ({gb / t1:.2f} TB/s written)  x{t0 / t1:.2f}", flush=True)
    g1, g2 = torch.randn(B, 120, 160, 256, device="cuda"), torch.randn(B, 120, 160, 256, device="cuda")
    t0, t1 = timeit(lambda: ops.corr_volume_1d(g1, g2)), timeit(lambda: ops.corr_volume_1d(g1, g2, split=True))
    gb = B * 120 * 160 * 160 * 4 / 1e9
    print(f"1-D B{B}: f32 {t0:.3f} ms  split {t1:.3f} ms ({gb / t1:.2f} TB/s written)  x{t0 / t1:.2f}", flush=True)
