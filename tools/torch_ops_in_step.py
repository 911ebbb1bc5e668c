#!/usr/bin/env python3
"""Which ATen operators (and from where) launch the non-hand-written kernels of one bench step."""
import os, sys
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic as syn
from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
from visual_odometry_amd.raft.models import RAFTStereoConfig
B, H, W = 32, 480, 640
vo = VisualOdometry(VOConfig(iters=12, stereo=RAFTStereoConfig.realtime(), camera_offset=syn.CAMERA_OFFSET), syn.left_K(1.0),
                    syn.focal_length_px(), syn.baseline(), np.ones((H, W), bool), seed=0)
left, right = syn.stereo_sequence(seed=100, frames=B + 1, h=H, w=W)
left, right = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
_, _, cached = vo.process_chunk(left, right, None)
vo.process_chunk(left, right, cached)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    vo.process_chunk(left, right, cached)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and "convolution" not in e.key and e.device_time_total > 50]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:30]:
    print(f"{e.self_device_time_total / 1e3:8.2f} ms self  x{e.count:4d}  {e.key:28s} {str(e.input_shapes)[:110]}")
