#!/usr/bin/env python3
"""Rounding error of the F(4x4,3x3) Winograd convolution when the products of the transformed
domain are formed from bf16 high + low parts (three terms: lo*hi + hi*lo + hi*hi, float32
accumulation) or from three bf16 parts per operand (six terms) instead of float32 operands,
against the float64 direct convolution.  Runs on the CPU: a product of two bf16 values is exact in float32, so float32 matmuls of the rounded parts
reproduce what three bf16 matrix instructions with float32 accumulation compute (up to the
order of the additions).  Usage: wino_split_accuracy.py [C ...]"""
import sys
import torch
import torch.nn.functional as F

BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                   [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                  [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def split(t):
    hi = t.to(torch.bfloat16).to(torch.float32)
    lo = (t - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


def winograd(x, w, mode):
    """x (C,H,W) float32 with H, W multiples of 4; w (K,C,3,3) float32; returns (K,H,W) float32."""
    C, H, W = x.shape
    K = w.shape[0]
    bt, g, at = BT.float(), G.float(), AT.float()
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(1, 6, 4).unfold(2, 6, 4)                       # (C, th, tw, 6, 6)
    th, tw = d.shape[1], d.shape[2]
    V = (bt @ d @ bt.T).permute(3, 4, 0, 1, 2).reshape(36, C, th * tw)
    U = (g @ w @ g.T).permute(2, 3, 0, 1).reshape(36, K, C)
    if mode == "f32":
        M = U @ V
    else:
        Uh, Ul = split(U)
        Vh, Vl = split(V)
        if mode == "bf16":
            M = Uh @ Vh
        elif mode == "split":
            M = (Ul @ Vh + Uh @ Vl) + Uh @ Vh
        else:                                                    # three parts per operand, six terms
            Um, Ul = split(U - Uh)
            Vm, Vl = split(V - Vh)
            M = ((Ul @ Vh + Uh @ Vl) + Um @ Vm) + (Um @ Vh + Uh @ Vm) + Uh @ Vh
    M = M.reshape(6, 6, K, th, tw).permute(2, 3, 4, 0, 1)        # (K, th, tw, 6, 6)
    Y = at @ M @ at.T                                            # (K, th, tw, 4, 4)
    return Y.permute(0, 1, 3, 2, 4).reshape(K, H, W)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for C in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
        K, H, W = 64, 48, 64
        x = torch.relu(torch.randn(C, H, W)) * 2.0               # activations after a ReLU
        w = torch.randn(K, C, 3, 3) / (9 * C) ** 0.5
        ref = F.conv2d(x[None].double(), w.double(), None, 1, 1)[0]
        big = ref.abs().max().item()
        out = [f"C = {C:3d}: max error / max |y|:"]
        for mode, label in (("f32", "float32 products"), ("split", "split bf16, 3 terms"),
                            ("split3", "3-part bf16, 6 terms"), ("bf16", "plain bf16")):
            e = (winograd(x, w, mode).double() - ref).abs().max().item() / big
            out.append(f"{label} {e:.1e}")
        print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()
