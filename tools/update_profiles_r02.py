#!/usr/bin/env python3
"""Fold what tools/collect_profiles_r02.sh left under gpurun_out/ into profiles/: the PMC traffic table
and the matrix-core-busy table bench.py reads (pmc_traffic.json, mfma_util.json; copies with the round tag),
the steady-state step breakdowns and the kernel stats of the traced bench run."""
import json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = lambda n: os.path.join(R, "gpurun_out", n)
P = lambda n: os.path.join(R, "profiles", n)
TAG = "r02"
fetch, write, mfma = (json.load(open(G(f"{TAG}_pmc_{c}.json"))) for c in ("FETCH_SIZE", "WRITE_SIZE", "mfma"))
names = {"conv3x3_bf16x3": "Geo<3, 3, 1", "conv1x5_bf16x3": "Geo<1, 5, 1", "conv5x1_bf16x3": "Geo<5, 1, 1",
         "conv3x3s2_bf16x3": "Geo<3, 3, 2", "conv1x1s2_bf16x3": "Geo<1, 1, 2", "conv7x7_bf16x3": "conv7_split_kernel",
         "corr2d_bf16x3": "corr_abt_kernel<128", "corr1d_bf16x3": "corr_abt_kernel<160", "pyr2d": "pyramid2d", "pyr1d": "pyramid1d",
         "lookup2d": "lookup2d", "lookup1d": "lookup1d", "instnorm_apply": "instnorm_apply_kernel", "conv1x1": "conv1x1_kernel"}
traffic = {"batch_pairs": 192, "height": 480, "width": 640,
           "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 0` "
                   "(tools/collect_profiles_r02.sh): the first chunk (193 stereo frames) and one steady step, mean per launch "
                   "over every launch of the kernel (all layer shapes). Values in KiB as reported. On gfx950 FETCH_SIZE counts "
                   "64 B per 128-B fabric read request, so wide streaming reads show half their bytes: read bytes ~= 2 x FETCH_SIZE; "
                   "WRITE_SIZE is exact.", "kernels": {}}
util = {"batch_pairs": 192, "height": 480, "width": 640,
        "note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), means per launch over the same run "
                "(all layer shapes of a kernel together)", "kernels": {}}
for key, sub in names.items():
    f, w = fetch.get(sub + "|FETCH_SIZE"), write.get(sub + "|WRITE_SIZE")
    if f and w:
        traffic["kernels"][key] = {"FETCH_SIZE_KiB_per_launch": f["mean_per_launch"], "launches_fetch": f["launches"],
                                   "WRITE_SIZE_KiB_per_launch": w["mean_per_launch"], "launches_write": w["launches"]}
    c = {k.split("|")[1]: v["mean_per_launch"] for k, v in mfma.items() if k.startswith(sub + "|")}
    if c.get("GRBM_GUI_ACTIVE"):
        c["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
        util["kernels"][key] = c
for name, obj in (("pmc_traffic.json", traffic), ("mfma_util.json", util)):
    json.dump(obj, open(P(name), "w"), indent=1)
    json.dump(obj, open(P(f"{TAG}_{name}"), "w"), indent=1)
for n in ("steady_state_step_breakdown.txt", "full_args_step_breakdown.txt", "bench_kernel_stats_top45.csv"):
    shutil.copy(G(f"{TAG}_{n}"), P(f"{TAG}_{n}"))
for k, v in traffic["kernels"].items():
    print(f"{k:20s} 2xFETCH+WRITE = {(2 * v['FETCH_SIZE_KiB_per_launch'] + v['WRITE_SIZE_KiB_per_launch']) * 1024 / 1e9:8.3f} GB/launch"
          f"   mfma busy {util['kernels'].get(k, {}).get('mfma_busy_frac')}")
