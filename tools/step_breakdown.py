#!/usr/bin/env python3
"""Per-kernel time of ONE steady-state bench step from a rocprofv3 kernel trace
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ...), and the same time grouped by
the C-ABI entry point the kernels belong to (an entry point may launch two kernels: full 64-channel blocks
and a last block of at most 32 channels)."""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "corr_abt_kernel<160" in r["Kernel_Name"]]
t0, t1 = int(rows[idx[-2]]["Start_Timestamp"]), int(rows[idx[-1]]["Start_Timestamp"])
seg = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
d, c = collections.Counter(), collections.Counter()
for r in seg:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:90]
    d[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); c[n] += 1
tot = sum(d.values())
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, busy {tot / 1e6:.2f} ms, {len(seg)} kernels")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for n, v in d.most_common(top):
    print(f"{v / 1e6:8.2f} ms {100 * v / tot:5.1f}% x{c[n]:4d}  {n}")

groups = [("vo_conv3x3_bf16x3", r"conv_bf16x3_kernel<.*Geo<3, 3, 1, 4, (32|16)>, (true|false), [12], false, false>"),
          ("vo_gru_conv_zr/q_bf16x3", r"conv_bf16x3_kernel<.*Geo<.*>, false, 2, true, false>"),
          ("vo_conv5_bf16x3", r"conv_bf16x3_kernel<.*Geo<(1, 5|5, 1), 1, 4, (32|16)>, (true|false), [12], false, false>"),
          ("vo_conv3x3s2[_shortcut]_bf16x3", r"conv_bf16x3_kernel<.*Geo<3, 3, 2"), ("vo_conv1x1s2_bf16x3", r"conv_bf16x3_kernel<.*Geo<1, 1, 2"),
          ("vo_conv1x1[_pixel_major]_bf16x3", r"conv_bf16x3_kernel<.*Geo<1, 1, 1|conv1x1_stream_kernel"), ("vo_conv7x7_bf16x3", r"conv7_split_kernel"),
          ("vo_conv1x1 / _pixel_major / taps", r"conv1x1_kernel"), ("vo_instance_norm_apply", r"instnorm_apply_kernel"),
          ("vo_corr_volume_2d_bf16x3", r"corr_abt_kernel<128"), ("vo_corr_volume_2d_presplit", r"corr_presplit_big_kernel"), ("vo_corr_volume_1d_bf16x3", r"corr_abt_kernel<160"),
          ("vo_corr_pyramid_2d", r"pyramid2d_kernel"), ("vo_corr_lookup_2d", r"lookup2d_kernel"), ("vo_corr_lookup_1d", r"lookup1d_kernel")]
print("\nby entry point (kernel launches of one step; a call with a last block of <= 32 channels launches two kernels):")
for name, pat in groups:
    sel = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg if re.search(pat, r["Kernel_Name"])]
    if sel:
        print(f"{sum(sel) / 1e6:8.2f} ms {100 * sum(sel) / tot:5.1f}%  {len(sel):4d} kernel launches  {name}")
