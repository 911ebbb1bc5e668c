#!/usr/bin/env python3
"""Experiment: let MIOpen time every applicable solver per convolution shape (PyTorch's
cudnn.benchmark switch) and keep its answers in a find-db under DB_DIR, then time the
steady-state step with the tuned choices.  Compare with the default (immediate-mode
heuristics).  Usage: miopen_find.py DB_DIR [batch]"""
import os, sys, threading, time
db = os.path.abspath(sys.argv[1])
os.makedirs(db, exist_ok=True)
os.environ["MIOPEN_USER_DB_PATH"] = db
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(db, "kcache"))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic as syn
from visual_odometry_amd.pipeline import VisualOdometry, VOConfig
from visual_odometry_amd.raft.models import RAFTStereoConfig

B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
stop = False
def beat():
    t0 = time.time()
    while not stop:
        with open(os.path.join(db, "heartbeat.log"), "a") as f:
            f.write(f"{time.time() - t0:.0f}s\n")
        time.sleep(20)
threading.Thread(target=beat, daemon=True).start()

H, W = 480, 640
torch.backends.cudnn.benchmark = True
vo = VisualOdometry(VOConfig(iters=12, stereo=RAFTStereoConfig.realtime(), camera_offset=syn.CAMERA_OFFSET), syn.left_K(),
                    syn.focal_length_px(), syn.baseline(), np.ones((H, W), bool), seed=0)
left, right = syn.stereo_sequence(100, B + 1, H, W)
l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
t = time.time()
_, _, cached = vo.process_chunk(l, r, None)
torch.cuda.synchronize()
print(f"find pass 1: {time.time() - t:.1f} s", flush=True)
t = time.time()
vo.process_chunk(l, r, cached); torch.cuda.synchronize()
print(f"find pass 2: {time.time() - t:.1f} s", flush=True)
for _ in range(2):
    vo.process_chunk(l, r, cached)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    vo.process_chunk(l, r, cached)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f"tuned: {dt * 1e3:.1f} ms/step  {B / dt:.1f} pairs/s", flush=True)
stop = True
