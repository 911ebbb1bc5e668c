#!/usr/bin/env python3
"""Calibrates rocprofv3's FETCH_SIZE for GATHERED reads: random rows of 32 / 64 / 128 / 256 bytes out of a 2 GiB
table (far beyond the 256 MiB on-die cache), so the bytes that must come from memory are known."""
import sys
import torch
torch.manual_seed(0)
n = 4 * 1024 * 1024
for row_floats in (8, 16, 32, 64):
    rows = (2 << 30) // (4 * row_floats)
    table = torch.randn(rows, row_floats, device="cuda")
    idx = torch.randint(0, rows, (n,), device="cuda")
    out = torch.index_select(table, 0, idx)
    torch.cuda.synchronize()
    print(f"rows of {4 * row_floats} B: gathered {n * 4 * row_floats / 2**20:.0f} MiB + indices {n * 8 / 2**20:.0f} MiB, written {out.numel() * 4 / 2**20:.0f} MiB", flush=True)
    del table, idx, out
