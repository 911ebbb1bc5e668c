#!/usr/bin/env python3
"""Fold what tools/collect_profiles_r03.sh left under gpurun_out/ into profiles/: the PMC traffic table
and the matrix-core-busy table bench.py reads (pmc_traffic.json, mfma_util.json; copies with the round tag),
the steady-state step breakdowns and the kernel stats of the traced bench run."""
import json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = lambda n: os.path.join(R, "gpurun_out", n)
P = lambda n: os.path.join(R, "profiles", n)
TAG = "r03"
fetch, write, mfma = (json.load(open(G(f"{TAG}_pmc_{c}.json"))) for c in ("FETCH_SIZE", "WRITE_SIZE", "mfma"))
import re
# entry point -> pattern over the kernels' templated names (GRU stores are the instantiations whose last argument is true)
names = {"conv3x3_bf16x3": r"Geo<3, 3, 1, 4, (32|16)>, (true|false), [12], false, false>", "conv5_bf16x3": r"Geo<(1, 5|5, 1), 1, 4, (32|16)>, (true|false), [12], false, false>",
         "gru_conv_bf16x3": r"Geo<.*>, false, 2, true, false>", "conv3x3s2_bf16x3": r"Geo<3, 3, 2", "conv1x1s2_bf16x3": r"Geo<1, 1, 2",
         "conv1x1_bf16x3": r"Geo<1, 1, 1|conv1x1_stream_kernel", "conv7x7_bf16x3": r"conv7_split_kernel", "corr2d": r"corr_abt_kernel<128, 128, 32, 2, 2, false", "corr2d_bf16x3": r"corr_abt_kernel<128, 128, 32, 2, 2, true",
         "corr1d": r"corr_abt_kernel<160, 160, 16, 1, 5, false", "corr1d_bf16x3": r"corr_abt_kernel<160, 160, 16, 1, 5, true", "corr2d_presplit": r"corr_presplit_big_kernel", "rows_amax": r"rows_amax_kernel", "split_rows": r"split_rows_kernel", "pyr2d": r"pyramid2d", "pyr1d": r"pyramid1d", "lookup2d": r"lookup2d", "lookup1d": r"lookup1d",
         "instnorm_apply": r"instnorm_apply_kernel", "conv1x1": r"conv1x1_kernel"}


def grouped(table, pattern, counter):
    """Launch-weighted mean of `counter` over the kernels whose name matches `pattern`."""
    tot, n = 0.0, 0
    for key, v in table.items():
        name, c = key.rsplit("|", 1)
        if c == counter and re.search(pattern, name):
            tot += v["mean_per_launch"] * v["launches"]; n += v["launches"]
    return (tot / n, n) if n else (None, 0)


traffic = {"batch_pairs": 192, "height": 480, "width": 640, "conv_arithmetic": "f16x3",
           "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 0` "
                   "(tools/collect_profiles_r03.sh): the first chunk (193 stereo frames) and one steady step, launch-weighted mean "
                   "over every launch of the entry point's kernels (all layer shapes). Values in KiB as reported. On gfx950 FETCH_SIZE "
                   "counts 64 B per 128-B fabric read request of a wide streaming read, so such reads show half their bytes: read bytes "
                   "~= 2 x FETCH_SIZE; gathered reads (the lookups) are counted per touched 128-byte line without that halving "
                   "(profiles/r02_lookup_layouts.txt); WRITE_SIZE is exact.", "kernels": {}}
util = {"batch_pairs": 192, "height": 480, "width": 640, "conv_arithmetic": "f16x3",
        "note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), launch-weighted means over the same run "
                "(all layer shapes of an entry point together)", "kernels": {}}
for key, pat in names.items():
    f, nf = grouped(fetch, pat, "FETCH_SIZE")
    w, nw = grouped(write, pat, "WRITE_SIZE")
    if nf and nw:
        traffic["kernels"][key] = {"FETCH_SIZE_KiB_per_launch": f, "launches_fetch": nf, "WRITE_SIZE_KiB_per_launch": w, "launches_write": nw}
    c = {}
    for counter in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"):
        v, n = grouped(mfma, pat, counter)
        if n:
            c[counter] = v; c["launches"] = n
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
