#!/usr/bin/env python3
"""Correlation volume: float32 matrix instruction, three bf16 terms per product split inside the kernel, and
the same on rows split beforehand (time, error)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops, _lib


def timeit(fn, n=5):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


lib = _lib.load()
knob = getattr(lib, "vo_debug_corr_nt", None)
for B in (16, 64, 192):
    f1, f2 = torch.randn(B, 60, 80, 256, device="cuda"), torch.randn(B, 60, 80, 256, device="cuda")
    want = (f1[0].view(4800, 256).double() @ f2[0].view(4800, 256).double().T / 16.0)
    s1, s2 = ops.split_feature_rows(f1, 32), ops.split_feature_rows(f2, 32)
    err = lambda c: (c.view(4800, 4800).double() - want).abs().max().item() / want.abs().max().item()
    e0, e1, e2 = err(ops.corr_volume_2d(f1[:1], f2[:1])), err(ops.corr_volume_2d(f1[:1], f2[:1], split=True)), err(ops.corr_volume_2d_presplit(s1[:1], s2[:1]))
    same = torch.equal(ops.corr_volume_2d(f1[:1], f2[:1], split=True), ops.corr_volume_2d_presplit(s1[:1], s2[:1]))
    gb = B * 4800 * 4800 * 4 / 1e9
    for nt in ((0, 1) if knob else (0,)):
        if knob: knob(nt)
        t0 = timeit(lambda: ops.corr_volume_2d(f1, f2)) if B <= 64 else float("nan")
        t1, t2 = timeit(lambda: ops.corr_volume_2d(f1, f2, split=True)), timeit(lambda: ops.corr_volume_2d_presplit(s1, s2))
        ts = timeit(lambda: ops.split_feature_rows(f1, 32))
        print(f"2-D B{B} nt{nt}: err f32 {e0:.1e} split {e1:.1e} presplit {e2:.1e} (equal to split: {same}) | f32 {t0:.3f} ms  split {t1:.3f} ms ({gb / t1:.2f} TB/s)  presplit {t2:.3f} ms ({gb / t2:.2f} TB/s written)  split pass {ts:.3f} ms", flush=True)
    if B > 64: continue
    g1, g2 = torch.randn(B, 120, 160, 256, device="cuda"), torch.randn(B, 120, 160, 256, device="cuda")
    r1, r2 = ops.split_feature_rows(g1, 16), ops.split_feature_rows(g2, 16)
    same = torch.equal(ops.corr_volume_1d(g1[:1], g2[:1], split=True), ops.corr_volume_1d_presplit(r1[:1], r2[:1]))
    gb = B * 120 * 160 * 160 * 4 / 1e9
    for nt in ((0, 1) if knob else (0,)):
        if knob: knob(nt)
        t0, t1, t2 = timeit(lambda: ops.corr_volume_1d(g1, g2)), timeit(lambda: ops.corr_volume_1d(g1, g2, split=True)), timeit(lambda: ops.corr_volume_1d_presplit(r1, r2))
        print(f"1-D B{B} nt{nt}: equal {same} | f32 {t0:.3f} ms  split {t1:.3f} ms ({gb / t1:.2f} TB/s)  presplit {t2:.3f} ms ({gb / t2:.2f} TB/s written)", flush=True)
if knob: knob(0)
