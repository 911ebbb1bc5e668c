#!/usr/bin/env python3
"""A few steps of the pipeline with float16 activations in the conv nets (mixed_precision), for a kernel trace
(rocprofv3 --kernel-trace -d DIR -- python3 tools/amp_step.py; then tools/step_breakdown.py DIR)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic as syn
from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
from visual_odometry_amd.raft.models import RAFTStereoConfig
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mixed = (sys.argv[2] if len(sys.argv) > 2 else "f16") == "f16"
H, W = 480, 640
vo = VisualOdometry(VOConfig(iters=12, stereo=RAFTStereoConfig.realtime()), syn.left_K(), syn.focal_length_px(),
                    syn.baseline(), np.ones((H, W), bool))
vo.flow_solver.model.mixed_precision = mixed; vo.disparity_solver.model.mixed_precision = mixed
left, right = syn.stereo_sequence(100, B + 1, H, W)
l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
_, _, cached = vo.process_chunk(l, r, None)
for _ in range(2): vo.process_chunk(l, r, cached)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(4): vo.process_chunk(l, r, cached)
torch.cuda.synchronize(); d = (time.perf_counter() - t) / 4
print(f"{d*1e3:.1f} ms/step {B/d:.1f} pairs/s", flush=True)
