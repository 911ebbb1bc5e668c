#!/usr/bin/env python3
"""Does one pipeline step capture into a HIP graph, and what does replaying it cost against eager launches?
Usage: graph_probe.py [batch]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic as syn
from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
from visual_odometry_amd.raft.models import RAFTStereoConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W = 480, 640
cfg = VOConfig(iters=12, stereo=RAFTStereoConfig.realtime(), camera_offset=syn.CAMERA_OFFSET, pose="svd")
vo = VisualOdometry(cfg, syn.left_K(1.0), syn.focal_length_px(), syn.baseline(), np.ones((H, W), bool), seed=0)
left, right = syn.stereo_sequence(seed=100, frames=B + 1, h=H, w=W)
left, right = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
_, _, cached = vo.process_chunk(left, right, None)
for _ in range(2):
    T0, sk0, _ = vo.process_chunk(left, right, cached)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    vo.process_chunk(left, right, cached)
torch.cuda.synchronize()
eager = (time.perf_counter() - t) / 5 * 1e3
print(f"eager: {eager:.2f} ms per step of {B} pairs", flush=True)

g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    vo.process_chunk(left, right, cached)                 # warm-up on the capture stream
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    T1, sk1, _ = vo.process_chunk(left, right, cached)
torch.cuda.synchronize()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("pose difference eager vs replay:", (T1 - T0).abs().max().item(), "skips equal:", bool((sk1 == sk0).all()), flush=True)
t = time.perf_counter()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t) / 5 * 1e3
print(f"graph replay: {graph:.2f} ms per step  ({eager / graph:.3f}x)", flush=True)
