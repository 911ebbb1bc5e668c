#!/usr/bin/env python3
"""Per-shape device time of the Winograd convolutions inside one bench step.
Usage: wino_shapes.py [batch]"""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops, synthetic as syn
from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
from visual_odometry_amd.raft.models import RAFTStereoConfig
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, W = 480, 640
vo = VisualOdometry(VOConfig(iters=12, stereo=RAFTStereoConfig.realtime(), camera_offset=syn.CAMERA_OFFSET), syn.left_K(1.0),
                    syn.focal_length_px(), syn.baseline(), np.ones((H, W), bool), seed=0)
left, right = syn.stereo_sequence(seed=100, frames=B + 1, h=H, w=W)
left, right = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
_, _, cached = vo.process_chunk(left, right, None)
vo.process_chunk(left, right, cached)
events = []
orig = ops.conv3x3_winograd
def wrapped(x, transformed, K, *a, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig(x, transformed, K, *a, **kw); e1.record()
    tile = 4 if transformed.numel() == (K + 63) // 64 * 64 * x.shape[1] * 36 else 2
    events.append(((tuple(x.shape), K, tile, kw.get("in_stats") is not None or len(a) > 6 and a[6] is not None), e0, e1))
    return y
ops.conv3x3_winograd = wrapped
import torch.nn.functional as F
mi = []
orig_conv = F.conv2d
def conv_wrapped(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = orig_conv(x, w, b, stride, padding, dilation, groups); e1.record()
    mi.append(((tuple(x.shape), tuple(w.shape), str(stride), str(x.dtype)[6:], tuple(y.shape)), e0, e1))
    return y
F.conv2d = conv_wrapped
torch.cuda.synchronize()
vo.process_chunk(left, right, cached)
torch.cuda.synchronize()
d, c = collections.Counter(), collections.Counter()
for key, e0, e1 in events:
    d[key] += e0.elapsed_time(e1); c[key] += 1
tot = sum(d.values())
print(f"total {tot:.1f} ms in {len(events)} launches")
for key, ms in d.most_common():
    (N, C, h, w), K, tile, nrm = key
    gf = 2 * N * K * h * w * C * 9 / 1e9
    print(f"{ms:8.2f} ms {100 * ms / tot:5.1f}%  x{c[key]:3d}  N{N} {C}->{K} @{h}x{w} F{tile}{' norm-in' if nrm else ''}  {gf * c[key] / ms:6.1f} TF/s eff")

d, c = collections.Counter(), collections.Counter()
for key, e0, e1 in mi:
    d[key] += e0.elapsed_time(e1); c[key] += 1
tot = sum(d.values())
print(f"MIOpen convolutions (incl. layout transposes): total {tot:.1f} ms in {len(mi)} calls")
for key, ms in d.most_common(25):
    xs, ws, st, dt, ys = key
    gf = 2 * ys[0] * ys[1] * ys[2] * ys[3] * ws[1] * ws[2] * ws[3] / 1e9
    print(f"{ms:8.2f} ms {100 * ms / tot:5.1f}%  x{c[key]:3d}  x{xs} w{ws} stride {st}  {gf * c[key] / ms:6.1f} TF/s")
