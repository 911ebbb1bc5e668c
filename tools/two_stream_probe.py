#!/usr/bin/env python3
"""The step with RAFT-Flow on a second stream beside RAFT-Stereo (the two networks of a chunk are independent) against
the step with both on one stream: time and poses."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic as syn
from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
from visual_odometry_amd.raft.models import RAFTStereoConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
H, W = 480, 640
left, right = syn.stereo_sequence(seed=100, frames=B + 1, h=H, w=W)
left, right = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
cfg = VOConfig(iters=12, stereo=RAFTStereoConfig.realtime(), camera_offset=syn.CAMERA_OFFSET)
vo = VisualOdometry(cfg, syn.left_K(1.0), syn.focal_length_px(), syn.baseline(), np.ones((H, W), bool), seed=0)
side = torch.cuda.Stream()


def step_two_streams(cached):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        flow = vo.flows(left)
    depth = torch.cat([cached[None], vo.depth_maps(left[1:], right[1:])], dim=0)
    cur.wait_stream(side)
    T, skipped = vo.poses(depth, flow, 0)
    return T, skipped, depth[-1]


_, _, cached = vo.process_chunk(left, right, None)
for name, fn in (("one stream", lambda: vo.process_chunk(left, right, cached)), ("two streams", lambda: step_two_streams(cached)),
                 ("one stream", lambda: vo.process_chunk(left, right, cached)), ("two streams", lambda: step_two_streams(cached))):
    out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {1e3 * dt:.1f} ms per step, {B / dt:.1f} pairs/s, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, pose checksum {out[0].abs().sum().item():.9f}", flush=True)
