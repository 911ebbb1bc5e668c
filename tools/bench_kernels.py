#!/usr/bin/env python3
"""Device time of each hand-written kernel at the benchmark's shapes (events on
the launch stream, median of --reps launches after a warm-up)."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops, synthetic as syn

def timeit(fn, reps):
    for _ in range(3): fn()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    return float(np.median(ts)), float(np.min(ts))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8); ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--height", type=int, default=480); ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    B, H, W = a.batch, a.height, a.width
    dev = "cuda"; g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    h8, w8, h4, w4 = H // 8, W // 8, H // 4, W // 4
    n = h8 * w8
    f1, f2 = rnd(B, h8, w8, 256), rnd(B, h8, w8, 256)
    s1, s2 = rnd(B, h4, w4, 256), rnd(B, h4, w4, 256)
    corr = ops.corr_volume_2d(f1, f2); pyr = ops.corr_pyramid_2d(corr)
    corr1 = ops.corr_volume_1d(s1, s2); pyr1 = ops.corr_pyramid_1d(corr1)
    ys, xs = torch.meshgrid(torch.arange(h8, device=dev), torch.arange(w8, device=dev), indexing="ij")
    c2 = (torch.stack([xs, ys]).float()[None] + 2 * rnd(B, 2, h8, w8)).contiguous()
    ys, xs = torch.meshgrid(torch.arange(h4, device=dev), torch.arange(w4, device=dev), indexing="ij")
    c1 = (torch.stack([xs, ys]).float()[None] - 8 + 2 * rnd(B, 2, h4, w4)).contiguous()
    flow8, mask8 = rnd(B, 2, h8, w8), rnd(B, 576, h8, w8)
    flow4, mask4 = rnd(B, 2, h4, w4), rnd(B, 144, h4, w4)
    out2 = torch.empty(B, 324, h8, w8, device=dev); out1 = torch.empty(B, 36, h4, w4, device=dev)
    lv2 = sum((h8 >> i) * (w8 >> i) for i in range(4)); lv1 = sum(w4 >> i for i in range(4))
    cases = {
        "corr2d": (lambda: ops.corr_volume_2d(f1, f2), (2 * n * 256 + n * n) * 4 * B, 2.0 * n * n * 256 * B),
        "pyr2d": (lambda: ops.corr_pyramid_2d(corr), B * n * lv2 * 4, 0),
        "lookup2d": (lambda: ops.corr_lookup_2d(pyr, c2, 4, out2), B * n * (2 + 324 + 400) * 4, 0),
        "corr1d": (lambda: ops.corr_volume_1d(s1, s2), (2 * h4 * w4 * 256 + h4 * w4 * w4) * 4 * B, 2.0 * h4 * w4 * w4 * 256 * B),
        "pyr1d": (lambda: ops.corr_pyramid_1d(corr1), B * h4 * w4 * lv1 * 4, 0),
        "lookup1d": (lambda: ops.corr_lookup_1d(pyr1, c1, 4, out1), B * h4 * w4 * (2 + 36 + 40) * 4, 0),
        "up8": (lambda: ops.convex_upsample(flow8, mask8, 8, mask_scale=0.25), B * 4 * (578 * n + 2 * H * W), 0),
        "up4": (lambda: ops.convex_upsample(flow4, mask4, 4, out_channels=1, mask_scale=0.25), B * 4 * (146 * h4 * w4 + H * W), 0),
    }
    for k, (fn, nbytes, flops) in cases.items():
        if a.only and k not in a.only.split(","): continue
        med, mn = timeit(fn, a.reps)
        line = f"{k:9s} median {med:8.1f} us  min {mn:8.1f} us  {nbytes / med / 1e3:8.1f} GB/s"
        if flops: line += f"  {flops / med / 1e6:7.1f} TFLOP/s"
        print(line, flush=True)

if __name__ == "__main__":
    main()
