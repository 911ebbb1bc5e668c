#!/usr/bin/env python3
"""Headline and per-entry-point ms per step of a bench JSON line: tools/show_kernels.py file.json [substring | min_ms]."""
import json, sys
d = json.load(open(sys.argv[1]))
sel = sys.argv[2] if len(sys.argv) > 2 else "10"
k = d["kernels"]
try:
    lim = float(sel); rows = {n: v for n, v in k.items() if v["ms_per_step"] >= lim}
except ValueError:
    rows = {n: v for n, v in k.items() if sel in n}
print(round(d["value"], 1), {n: round(v["ms_per_step"], 2) for n, v in rows.items()})
