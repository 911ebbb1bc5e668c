#!/usr/bin/env python3
"""Launch only the hand-written kernels at the benchmark's shapes (for rocprofv3
--kernel-trace / --pmc passes, which are slow over the whole network).

  rocprofv3 --kernel-trace --stats -d out -- python3 tools/profile_kernels.py
  rocprofv3 --pmc FETCH_SIZE -d out -- python3 tools/profile_kernels.py --only lookup2d
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops, synthetic as syn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    B, H, W = a.batch, a.height, a.width
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    h8, w8, h4, w4 = H // 8, W // 8, H // 4, W // 4
    want = lambda k: not a.only or k in a.only.split(",")

    f1, f2 = rnd(B, h8, w8, 256), rnd(B, h8, w8, 256)
    s1, s2 = rnd(B, h4, w4, 256), rnd(B, h4, w4, 256)
    corr = ops.corr_volume_2d(f1, f2)
    pyr = ops.corr_pyramid_2d(corr)
    corr1 = ops.corr_volume_1d(s1, s2)
    pyr1 = ops.corr_pyramid_1d(corr1)
    ys, xs = torch.meshgrid(torch.arange(h8, device=dev), torch.arange(w8, device=dev), indexing="ij")
    c2 = (torch.stack([xs, ys]).float()[None] + 2 * rnd(B, 2, h8, w8)).contiguous()
    ys, xs = torch.meshgrid(torch.arange(h4, device=dev), torch.arange(w4, device=dev), indexing="ij")
    c1 = (torch.stack([xs, ys]).float()[None] - 8 + 2 * rnd(B, 2, h4, w4)).contiguous()
    flow8, mask8 = rnd(B, 2, h8, w8), rnd(B, 576, h8, w8)
    flow4, mask4 = rnd(B, 2, h4, w4), rnd(B, 144, h4, w4)
    disp = -60 * torch.rand(B + 1, H, W, device=dev, generator=g)
    from visual_odometry_amd.geometry import UniformKeyPoints
    kp = UniformKeyPoints(np.ones((H, W), bool)).get_keypoints_device(320 if W == 640 else 20000)
    flow_full = rnd(B, 2, H, W)
    Kinv = np.linalg.inv(syn.left_K(W / 640.0))
    if a.only and "encoders" in a.only.split(","):
        # the instance-norm / add+ReLU passes at exactly the shapes of one bench step:
        # RAFT-Flow's feature encoder on B+1 frames, RAFT-Stereo's on 2B images
        from visual_odometry_amd.raft.layers import BasicEncoder, fused_passes
        torch.manual_seed(0)
        enc8 = BasicEncoder(256, "instance", 3).to(dev).eval()
        enc4 = BasicEncoder(256, "instance", 2).to(dev).eval()
        i8, i4 = rnd(B + 1, 3, H, W), rnd(2 * B, 3, H, W)
        with torch.no_grad(), fused_passes():
            for _ in range(a.reps):
                enc8(i8); enc4(i4)
        torch.cuda.synchronize()
        print("done")
        return
    torch.cuda.synchronize()
    for _ in range(a.reps):
        if want("corr2d"): ops.corr_volume_2d(f1, f2)
        if want("pyr2d"): ops.corr_pyramid_2d(corr)
        if want("lookup2d"): ops.corr_lookup_2d(pyr, c2)
        if want("corr1d"): ops.corr_volume_1d(s1, s2)
        if want("pyr1d"): ops.corr_pyramid_1d(corr1)
        if want("lookup1d"): ops.corr_lookup_1d(pyr1, c1)
        if want("up8"): ops.convex_upsample(flow8, mask8, 8, mask_scale=0.25)
        if want("up4"): ops.convex_upsample(flow4, mask4, 4, out_channels=1, mask_scale=0.25)
        if want("geom"):
            depth = ops.disparity_to_depth(disp, 22.14)
            P1, P2, w, n = ops.pair_correspondences(kp, depth[:-1], depth[1:], flow_full, Kinv)
            ops.kabsch_batch(P1, P2, w)
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
