#!/usr/bin/env python3
"""A large chunk against the same frames in small chunks: poses, skip flags and the peak
device memory of the large step (index arithmetic and buffer sizes at bench chunk sizes).
Usage: chunk_parity.py [pairs_large=192] [pairs_small=16]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic as syn
from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
from visual_odometry_amd.raft.models import RAFTStereoConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
b = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H, W = 480, 640
cfg = VOConfig(iters=12, stereo=RAFTStereoConfig.realtime(), camera_offset=syn.CAMERA_OFFSET)
vo = VisualOdometry(cfg, syn.left_K(1.0), syn.focal_length_px(), syn.baseline(), np.ones((H, W), bool), seed=0)
left, right = syn.stereo_sequence(seed=100, frames=B + 1, h=H, w=W)
left, right = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
torch.cuda.reset_peak_memory_stats()
T_big, sk_big, _ = vo.process_chunk(left, right, None)
torch.cuda.synchronize()
print(f"{B} pairs in one chunk: peak device memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
T_big, sk_big = T_big.cpu(), sk_big.cpu()
del _
torch.cuda.empty_cache()
Ts, sks, cached = [], [], None
for s in range(0, B, b):
    T, sk, cached = vo.process_chunk(left[s:s + b + 1], right[s:s + b + 1], cached)
    Ts.append(T.cpu()); sks.append(sk.cpu())
T_small, sk_small = torch.cat(Ts), torch.cat(sks)
assert torch.equal(sk_big, sk_small), "skip flags differ"
ok = ~sk_big
d = (T_big[ok] - T_small[ok]).abs().max().item()
print(f"chunks of {b}: skip flags equal ({int(sk_big.sum())} skipped), max |T_big - T_small| = {d:.3e}", flush=True)
assert d < 1e-3, d
print("OK")
