#!/usr/bin/env python3
"""Mixed precision (float16 activations in the conv nets, the published models' switch;
NOT the reported configuration) against the float32 path the reference prescribes:
end-point error of flow and disparity, and step time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic as syn
from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
from visual_odometry_amd.raft.models import RAFTStereoConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = 480, 640
vo = VisualOdometry(VOConfig(iters=12, stereo=RAFTStereoConfig.realtime()), syn.left_K(), syn.focal_length_px(),
                    syn.baseline(), np.ones((H, W), bool))
left, right = syn.stereo_sequence(100, B + 1, H, W)
l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()

def run(mixed):
    vo.flow_solver.model.mixed_precision = mixed; vo.disparity_solver.model.mixed_precision = mixed
    with torch.no_grad():
        for _ in range(2):
            flow = vo.flows(l); disp = vo.disparity_solver.compute_disparity_gray_batch(l[1:], r[1:])
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3):
            flow = vo.flows(l); disp = vo.disparity_solver.compute_disparity_gray_batch(l[1:], r[1:])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    return flow, disp, dt

f32, d32, t32 = run(False)
print(f"f32   {t32*1e3:7.1f} ms/step  {B/t32:6.1f} pairs/s  |flow| mean {f32.abs().mean():.3f} |disp| mean {d32.abs().mean():.3f}", flush=True)
f, d, t = run(True)
epe = (f - f32).pow(2).sum(1).sqrt().mean().item(); de = (d - d32).abs().mean().item()
print(f"f16   {t*1e3:7.1f} ms/step  {B/t:6.1f} pairs/s  flow EPE {epe:.4f} px  disparity err {de:.4f} px", flush=True)
