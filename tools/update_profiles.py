#!/usr/bin/env python3
"""Fold the reductions tools/collect_profiles.sh left under gpurun_out/ into profiles/
(PMC traffic table read by bench.py, MFMA-utilisation table, step breakdown, kernel stats).
Usage: update_profiles.py ROUND_TAG [batch]"""
import json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
G = lambda n: os.path.join(R, "gpurun_out", n)
P = lambda n: os.path.join(R, "profiles", n)

def counter(file, key):
    d = json.load(open(G(file)))
    return d[key]

traffic = {"batch_pairs": batch, "height": 480, "width": 640,
           "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/collect_profiles.sh); values in "
                   "KiB per launch as reported, reduced with tools/pmc_summary.py. On gfx950 FETCH_SIZE counts 64 B per "
                   "128-B fabric read request, so wide streaming reads show half their bytes: read bytes ~= 2 x FETCH_SIZE; "
                   "WRITE_SIZE is exact.", "kernels": {}}
names = {"corr2d": "corr_abt_kernel<128", "corr1d": "corr_abt_kernel<160", "pyr2d": "pyramid2d", "pyr1d": "pyramid1d",
         "lookup2d": "lookup2d", "lookup1d": "lookup1d"}
for k, n in names.items():
    f, w = counter("pmc_FETCH_SIZE_corr.json", n + "|FETCH_SIZE"), counter("pmc_WRITE_SIZE_corr.json", n + "|WRITE_SIZE")
    traffic["kernels"][k] = {"FETCH_SIZE_KiB_per_launch": f["mean_per_launch"], "launches_fetch": f["launches"],
                             "WRITE_SIZE_KiB_per_launch": w["mean_per_launch"], "launches_write": w["launches"]}
f = counter("pmc_FETCH_SIZE_encoders.json", "instnorm_kernel|FETCH_SIZE")
w = counter("pmc_WRITE_SIZE_encoders.json", "instnorm_kernel|WRITE_SIZE")
traffic["kernels"]["instnorm"] = {"FETCH_SIZE_KiB_per_launch": f["mean_per_launch"], "launches_fetch": f["launches"],
                                  "WRITE_SIZE_KiB_per_launch": w["mean_per_launch"], "launches_write": w["launches"],
                                  "note": "mean over the launches of the feature encoders of both networks"}
for key, kern in (("winograd4", "wino4_kernel"), ("winograd1d", "wino1d_kernel")):
    f = counter("pmc_wino_FETCH_SIZE.json", kern + "|FETCH_SIZE")
    w = counter("pmc_wino_WRITE_SIZE.json", kern + "|WRITE_SIZE")
    traffic["kernels"][key] = {"FETCH_SIZE_KiB_per_launch": f["mean_per_launch"], "launches_fetch": f["launches"],
                               "WRITE_SIZE_KiB_per_launch": w["mean_per_launch"], "launches_write": w["launches"],
                               "note": "mean over every launch of a short bench run (all layer shapes of both networks)"}
json.dump(traffic, open(P(f"{tag}_pmc_traffic.json"), "w"), indent=1)
json.dump(traffic, open(P("pmc_traffic.json"), "w"), indent=1)

util_path = P(f"{tag}_mfma_util.json")
util = json.load(open(util_path)) if os.path.exists(util_path) else {"kernels": {}}
for shape, kern in (("16_64_480_640_64", "wino4_kernel"), ("16_256_120_160_128", "wino4_kernel"),
                    ("32_384_60_80_256_3_5", "wino1d_kernel")):
    d = json.load(open(G(f"pmc_w1_{shape}.json")))
    c = {k.split("|")[1]: v["mean_per_launch"] for k, v in d.items() if k.startswith(kern)}
    c["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
    util["kernels"][f"{kern} N,C,H,W,K = {','.join(shape.split('_')[:5])}"] = c
util["note_winograd"] = ("same counters over tools/wino_one.py; mfma_util = MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE/8 XCDs); "
                         "wino_kernel rows are the F(2x2,3x3) kernel, wino4_kernel F(4x4,3x3), wino1d_kernel F(4,5)")
json.dump(util, open(util_path, "w"), indent=1)

shutil.copy(G("step_breakdown.txt"), P(f"{tag}_steady_state_step_breakdown.txt"))
shutil.copy(G("bench_kernel_stats_top45.csv"), P(f"{tag}_bench_kernel_stats_top45.csv"))
print(json.dumps({k: v["mfma_util"] for k, v in util["kernels"].items() if "wino" in k}, indent=1))
t = traffic["kernels"]["winograd4"]
print("winograd bytes/launch (2xFETCH + WRITE):", (2 * t["FETCH_SIZE_KiB_per_launch"] + t["WRITE_SIZE_KiB_per_launch"]) * 1024 / 1e6, "MB")
