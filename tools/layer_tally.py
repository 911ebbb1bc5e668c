#!/usr/bin/env python3
"""Device time of one steady-state step of the pipeline, per entry point AND per shape.
Every C-ABI call is bracketed by two events; calls are keyed on the entry point's name and
its integer arguments (shapes, strides, flags), so the table shows which layer shapes the
time goes to.  Usage: layer_tally.py [batch] [stereo realtime|full] [iters] [top]"""
import collections, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import _lib, ops, synthetic as syn
from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
from visual_odometry_amd.raft.models import RAFTStereoConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
stereo = sys.argv[2] if len(sys.argv) > 2 else "realtime"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 12
top = int(sys.argv[4]) if len(sys.argv) > 4 else 60
H, W = 480, 640
cfg = VOConfig(iters=iters, stereo=RAFTStereoConfig.realtime() if stereo == "realtime" else RAFTStereoConfig(),
               camera_offset=syn.CAMERA_OFFSET, pose="svd")
vo = VisualOdometry(cfg, syn.left_K(1.0), syn.focal_length_px(), syn.baseline(), np.ones((H, W), bool), seed=0)
left, right = syn.stereo_sequence(seed=100, frames=B + 1, h=H, w=W)
left, right = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
_, _, cached = vo.process_chunk(left, right, None)
vo.process_chunk(left, right, cached)
torch.cuda.synchronize()

log = []
real_call = _lib.call


def tallied(name, *args):
    sig = _lib.SIGNATURES.get(name, [])
    ints = tuple(int(a) for a, t in zip(args, sig) if t in (ctypes.c_int, ctypes.c_int64) and abs(int(a)) < (1 << 31))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    real_call(name, *args)
    e.record()
    log.append((name, ints, s, e))


_lib.call = tallied
wall0, wall1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
wall0.record()
vo.process_chunk(left, right, cached)
wall1.record()
torch.cuda.synchronize()
_lib.call = real_call
t, c = collections.Counter(), collections.Counter()
for name, ints, s, e in log:
    t[(name, ints)] += s.elapsed_time(e); c[(name, ints)] += 1
tot = sum(t.values())
print(f"batch {B} stereo {stereo} iters {iters}: step {wall0.elapsed_time(wall1):.1f} ms, in C-ABI calls {tot:.1f} ms, "
      f"{len(log)} calls")
byname = collections.Counter()
for (name, ints), v in t.items():
    byname[name] += v
for name, v in byname.most_common():
    print(f"{v:8.2f} ms {100 * v / tot:5.1f}%  {name}")
print()
for (name, ints), v in t.most_common(top):
    print(f"{v:8.2f} ms {100 * v / tot:5.1f}% x{c[(name, ints)]:3d} {v / c[(name, ints)]:7.3f} ms  {name} {ints}")
