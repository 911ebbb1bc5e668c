#!/usr/bin/env python3
"""Flow / disparity deviation of the Winograd convolution kernels from the all-MIOpen float32 networks
(640x480 synthetic frames, 12 iterations)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic as syn
from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
from visual_odometry_amd.raft import layers
from visual_odometry_amd.raft.models import RAFTStereoConfig
H, W, B = 480, 640, 4
vo = VisualOdometry(VOConfig(iters=12, stereo=RAFTStereoConfig.realtime(), camera_offset=syn.CAMERA_OFFSET), syn.left_K(1.0),
                    syn.focal_length_px(), syn.baseline(), np.ones((H, W), bool), seed=0)
left, right = syn.stereo_sequence(seed=100, frames=B + 1, h=H, w=W)
left, right = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
def run():
    with torch.no_grad():
        return vo.flows(left), vo.disparity_solver.compute_disparity_gray_batch(left[1:], right[1:])
vo.set_conv_arithmetic("f32")
with layers.arithmetic(hand_kernels=False):
    f0, d0 = run()
    f0b, d0b = run()
print(f"MIOpen run to run: flow EPE {(f0b - f0).pow(2).sum(1).sqrt().mean().item():.2e} px, disparity {(d0b - d0).abs().mean().item():.2e} px")
for tile in (2, 4, 0):
    with layers.arithmetic(winograd_tile=tile):
        f, d = run()
    print(f"tile {tile}: flow EPE mean {(f - f0).pow(2).sum(1).sqrt().mean().item():.2e} max {(f - f0).abs().max().item():.2e} px; "
          f"disparity mean {(d - d0).abs().mean().item():.2e} max {(d - d0).abs().max().item():.2e} px")
for kind in ("bf16x3", "f16x3"):
    vo.set_conv_arithmetic(kind)
    f, d = run()
    print(f"{kind}: flow EPE mean {(f - f0).pow(2).sum(1).sqrt().mean().item():.2e} max {(f - f0).abs().max().item():.2e} px; "
          f"disparity mean {(d - d0).abs().mean().item():.2e} max {(d - d0).abs().max().item():.2e} px")
