#!/usr/bin/env python3
"""Per-kernel time of ONE steady-state bench step from a rocprofv3 kernel trace
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ...)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "corr_abt_kernel<160" in r["Kernel_Name"]]
t0, t1 = int(rows[idx[-2]]["Start_Timestamp"]), int(rows[idx[-1]]["Start_Timestamp"])
seg = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
d, c = collections.Counter(), collections.Counter()
for r in seg:
    n = r["Kernel_Name"][:90]
    d[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); c[n] += 1
tot = sum(d.values())
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, busy {tot / 1e6:.2f} ms, {len(seg)} kernels")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for n, v in d.most_common(top):
    print(f"{v / 1e6:8.2f} ms {100 * v / tot:5.1f}% x{c[n]:4d}  {n}")
