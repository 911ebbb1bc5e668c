#!/usr/bin/env python3
"""Time every distinct convolution of the two networks at benchmark shapes, with
MIOpen's immediate mode vs its find (benchmark) mode."""
import sys, os, time
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd.raft.models import RAFTFlow, RAFTStereo, RAFTStereoConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bench = len(sys.argv) > 2 and sys.argv[2] == "find"
torch.backends.cudnn.benchmark = bench
dev = "cuda"
records = {}

def hook(name):
    def f(m, inp, out):
        x = inp[0]
        key = (tuple(x.shape), m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding)
        records.setdefault(key, [name, 0])[1] += 1
    return f

torch.manual_seed(0)
flow = RAFTFlow().to(dev).eval(); stereo = RAFTStereo(RAFTStereoConfig.realtime()).to(dev).eval()
for n, m in list(flow.named_modules()) + list(stereo.named_modules()):
    if isinstance(m, nn.Conv2d): m.register_forward_hook(hook(n))
img = torch.rand(B, 3, 480, 640, device=dev) * 255
with torch.no_grad():
    flow(img, img, iters=1); stereo(img, img, iters=1)
torch.cuda.synchronize()
rows = []
for key, (name, cnt) in records.items():
    shp, cin, cout, k, s, p = key
    conv = nn.Conv2d(cin, cout, k, s, p).to(dev)
    x = torch.randn(*shp, device=dev)
    with torch.no_grad():
        for _ in range(2): conv(x)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): y = conv(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    fl = 2.0 * y.numel() * cin * k[0] * k[1]
    rows.append((dt * 1e3, fl / dt / 1e12, name, shp, cin, cout, k, s))
rows.sort(reverse=True)
for r in rows:
    print(f"{r[0]:8.3f} ms {r[1]:7.2f} TF  {r[2]:40s} in={r[3]} {r[4]}->{r[5]} k={r[6]} s={r[7]}")
