#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by tools/profile.sh into one small JSON (kept under profiles/)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir = sys.argv[1]
res = {"kernels": {}, "pmc": {}}
stats = glob.glob(os.path.join(out_dir, "trace_kernel_stats.csv"))
if stats:
    for r in csv.DictReader(open(stats[0])):
        res["kernels"][r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                   "total_us": float(r["TotalDurationNs"]) / 1e3, "pct": float(r["Percentage"])}
for tag in ("fetch", "write", "sq"):
    f = glob.glob(os.path.join(out_dir, tag + "_counter_collection.csv"))
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, vals in cs.items():
            res["pmc"].setdefault(k, {})[c] = {"mean_per_dispatch": sum(vals) / len(vals), "dispatches": len(vals)}
path = os.path.join(out_dir, "summary.json")
json.dump(res, open(path, "w"), indent=1, sort_keys=True)
lk = [k for k in res["pmc"] if "k_fb_klt" in k]
for k in lk:
    print(k, json.dumps(res["pmc"][k]))
print("wrote", path)
