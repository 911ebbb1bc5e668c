#!/usr/bin/env python3
"""Mean counter value per launch and kernel from a rocprofv3 --pmc output directory
(…/*_counter_collection.csv).  Usage: pmc_summary.py DIR [name-substring ...]"""
import collections, csv, glob, json, sys
csv.field_size_limit(1 << 30)
f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
want = sys.argv[2:]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if want and not any(w in name for w in want):
        continue
    # without a filter: the kernel's name up to its parameter list (template arguments kept), last 90 characters
    clean = name.replace("(anonymous namespace)::", "")
    short = next(w for w in want if w in name) if want else (clean[:clean.rfind(">(") + 1] if ">(" in clean else clean.split("(")[0])[-90:]
    a = acc[(short, r["Counter_Name"])]
    a[0] += float(r["Counter_Value"]); a[1] += 1
print(json.dumps({f"{k[0]}|{k[1]}": {"mean_per_launch": v[0] / v[1], "launches": v[1]} for k, v in acc.items()}, indent=1))
