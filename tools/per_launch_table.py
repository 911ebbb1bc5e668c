#!/usr/bin/env python3
"""One step of the default bench workload with the launch recorder on: every launch of the split convolution entry
points with its duration, issued TFLOP/s and algorithmic GB/s, grouped by (entry point, flops, bytes) = layer shape,
sorted by time.  Shows which layer shapes hold the step's time and how far each is from the kernel's best."""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops, synthetic as syn
from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
from visual_odometry_amd.raft.models import RAFTStereoConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
H, W = 480, 640
left, right = syn.stereo_sequence(seed=100, frames=B + 1, h=H, w=W)
left, right = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
cfg = VOConfig(iters=12, stereo=RAFTStereoConfig.realtime(), camera_offset=syn.CAMERA_OFFSET)
vo = VisualOdometry(cfg, syn.left_K(1.0), syn.focal_length_px(), syn.baseline(), np.ones((H, W), bool), seed=0)
_, _, cached = vo.process_chunk(left, right, None)
vo.process_chunk(left, right, cached)
rec = {}
ops.set_recorder(rec)
vo.process_chunk(left, right, cached)
torch.cuda.synchronize()
ops.set_recorder(None)
groups = collections.defaultdict(list)
for name, evs in rec.items():
    for ev in evs:
        groups[(name, ev[3], ev[2])].append(ev[0].elapsed_time(ev[1]))
rows = sorted(((sum(ms), name, fl, by, ms) for (name, fl, by), ms in groups.items()), reverse=True)
total = sum(r[0] for r in rows)
print(f"{total:.1f} ms in recorded launches")
for tot, name, fl, by, ms in rows[:45]:
    n = len(ms)
    tf = fl / (tot / n * 1e-3) / 1e12 if fl else 0.0
    gb = by / (tot / n * 1e-3) / 1e9 if by else 0.0
    print(f"{tot:8.2f} ms {100 * tot / total:5.1f}%  x{n:3d}  {tot / n:7.3f} ms each  {tf:7.1f} TFLOP/s  {gb:7.1f} GB/s  {fl / 1e12:8.3f} TFLOP {by / 1e9:7.2f} GB  {name}")
