#!/usr/bin/env python3
"""Largest error, relative to the largest output, of every way this library can evaluate the networks' heavy
layers: float16 pieces (three terms), bf16 pieces (three terms), the float32 matrix instruction (direct or
Winograd, whichever kernel the "f32" arithmetic uses for the layer) and the framework's own float32 convolution,
each against the float64 convolution of the same data.  Layer shapes: the networks' own (channels x taps)."""
import os, sys
import torch
import torch.nn.functional as Fn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import ops

DEV = "cuda"
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False


def rnd(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).to(DEV)


def err(got, want):
    return (got.double() - want).abs().max().item() / want.abs().max().item()


def line(name, errs):
    print(f"{name:46s} " + "  ".join(f"{k} {v:8.2e}" for k, v in errs.items()), flush=True)


print("layer (N,C,H,W -> K)                           largest |error| / largest |output|")
for (N, C, H, W, K, relu_in) in [(2, 64, 120, 160, 64, True), (2, 64, 240, 320, 64, True), (2, 96, 120, 160, 96, True), (2, 128, 60, 80, 128, True),
                                 (2, 256, 60, 80, 192, True), (2, 128, 60, 80, 256, False), (2, 384, 60, 80, 128, False)]:
    x = rnd(N, C, H, W, seed=1)
    x = torch.relu(x) if relu_in else x
    w = rnd(K, C, 3, 3, seed=2) * (2.0 / (9 * C)) ** 0.5
    want = Fn.conv2d(x.double(), w.double(), None, 1, 1)
    line(f"3x3  ({N},{C},{H},{W})->{K}", {
        "f16x3": err(ops.conv3x3_winograd(x, ops.conv3x3_split_filter(w, "f16"), K), want),
        "bf16x3": err(ops.conv3x3_winograd(x, ops.conv3x3_split_filter(w, "bf16"), K), want),
        "f32 F(4x4)": err(ops.conv3x3_winograd(x, ops.winograd_filter(w, 4), K), want),
        "f32 F(2x2)": err(ops.conv3x3_winograd(x, ops.winograd_filter(w, 2), K), want),
        "framework f32": err(Fn.conv2d(x, w, None, 1, 1), want)})
for (N, C, H, W, K) in [(2, 256, 60, 80, 256), (2, 384, 60, 80, 128)]:
    for vertical in (False, True):
        x = rnd(N, C, H, W, seed=3)
        w = rnd(*((K, C, 5, 1) if vertical else (K, C, 1, 5)), seed=4) * (2.0 / (5 * C)) ** 0.5
        pad = (2, 0) if vertical else (0, 2)
        want = Fn.conv2d(x.double(), w.double(), None, 1, pad)
        line(f"{'5x1' if vertical else '1x5'}  ({N},{C},{H},{W})->{K}", {
            "f16x3": err(ops.conv5_split(x, ops.conv5_split_filter(w, "f16"), K, vertical), want),
            "bf16x3": err(ops.conv5_split(x, ops.conv5_split_filter(w, "bf16"), K, vertical), want),
            "f32 F(4,5)": err(ops.conv5_winograd(x, ops.winograd1d_filter(w), K, vertical), want),
            "framework f32": err(Fn.conv2d(x, w, None, 1, pad), want)})
for (N, C, H, W, K) in [(2, 64, 240, 320, 96), (2, 96, 120, 160, 128)]:
    x = torch.relu(rnd(N, C, H, W, seed=5))
    w = rnd(K, C, 3, 3, seed=6) * (2.0 / (9 * C)) ** 0.5
    want = Fn.conv2d(x.double(), w.double(), None, 2, 1)
    line(f"3x3 stride 2  ({N},{C},{H},{W})->{K}", {
        "f16x3": err(ops.conv3x3s2_split(x, ops.conv3x3_split_filter(w, "f16"), K), want),
        "bf16x3": err(ops.conv3x3s2_split(x, ops.conv3x3_split_filter(w, "bf16"), K), want),
        "f32 direct": err(ops.conv3x3s2(x, ops.conv3x3s2_filter(w), K), want),
        "framework f32": err(Fn.conv2d(x, w, None, 2, 1), want)})
for (N, C, H, W, K) in [(2, 336, 60, 80, 256), (2, 256, 60, 80, 576), (2, 256, 120, 160, 144)]:
    x = torch.relu(rnd(N, C, H, W, seed=7))
    w = rnd(K, C, 1, 1, seed=8) * (2.0 / C) ** 0.5
    want = Fn.conv2d(x.double(), w.double(), None)
    line(f"1x1  ({N},{C},{H},{W})->{K}", {
        "f16x3": err(ops.conv1x1_split(x, ops.conv1x1_split_filter(w, "f16"), K), want),
        "bf16x3": err(ops.conv1x1_split(x, ops.conv1x1_split_filter(w, "bf16"), K), want),
        "f32 direct": err(ops.conv1x1(x, ops.conv1x1_filter(w), K), want),
        "framework f32": err(Fn.conv2d(x, w, None), want)})
for (N, C, H, W, K, s) in [(2, 1, 480, 640, 64, 1), (2, 1, 480, 640, 64, 2), (2, 2, 60, 80, 128, 1), (2, 2, 120, 160, 64, 1)]:
    x = rnd(N, C, H, W, seed=9) * (0.5 if C == 1 else 20.0)
    w = rnd(K, C, 7, 7, seed=10) * (2.0 / (49 * C)) ** 0.5
    want = Fn.conv2d(x.double(), w.double(), None, s, 3)
    line(f"7x7 stride {s}  ({N},{C},{H},{W})->{K}", {
        "f16x3": err(ops.conv7x7_split(x, ops.conv7x7_split_filter(w, "f16"), K, stride=s), want),
        "bf16x3": err(ops.conv7x7_split(x, ops.conv7x7_split_filter(w, "bf16"), K, stride=s), want),
        "f32 direct": err(ops.conv7x7(x, ops.conv7x7_filter(w), K, stride=s), want),
        "framework f32": err(Fn.conv2d(x, w, None, s, 3), want)})
